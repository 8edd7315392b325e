"""Direct AQL dispatch (csrc/bsr_aql.h) lays a kernel's arguments out itself: the explicit ones packed with their natural
alignment, then code object v5's implicit block at the next multiple of eight -- block counts at +0, group sizes at +12,
remainders at +18, global offsets at +40, grid dimensions at +64, dynamic LDS size at +120.  This test reads the metadata
of every kernel in the built library's own gfx950 code objects (the ones the loader hands to ROCr) and checks that
assumption kernel by kernel, and that the loader's way of finding the code objects in the fat binary finds them all.
No GPU needed: llvm-readelf from the ROCm image."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mcmc-symreg_amd", "bsr", "libbsr_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

IMPLICIT = {"hidden_block_count_x": 0, "hidden_block_count_y": 4, "hidden_block_count_z": 8,
            "hidden_group_size_x": 12, "hidden_group_size_y": 14, "hidden_group_size_z": 16,
            "hidden_remainder_x": 18, "hidden_remainder_y": 20, "hidden_remainder_z": 22,
            "hidden_global_offset_x": 40, "hidden_global_offset_y": 48, "hidden_global_offset_z": 56,
            "hidden_grid_dims": 64, "hidden_dynamic_lds_size": 120}
# what aql_append fills in (everything else in the block stays zero)
FILLED = {"hidden_block_count_x", "hidden_block_count_y", "hidden_block_count_z", "hidden_group_size_x", "hidden_group_size_y",
          "hidden_group_size_z", "hidden_grid_dims", "hidden_dynamic_lds_size"}
ZERO_IS_RIGHT = {"hidden_remainder_x", "hidden_remainder_y", "hidden_remainder_z", "hidden_global_offset_x",
                 "hidden_global_offset_y", "hidden_global_offset_z"}   # grids are whole workgroups, no global offset


def code_objects(data):
    """(offset, size) of every gfx950 code object: the walk csrc/bsr_aql.hip find_code_objects does."""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    for m in re.finditer(re.escape(magic), data):
        at = m.start()
        n = struct.unpack_from("<Q", data, at + 24)[0]
        if n == 0 or n > 64:
            continue
        o = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            target = data[o:o + tl]
            o += tl
            if size and b"amdgcn" in target and b"gfx950" in target and data[at + off:at + off + 4] == b"\x7fELF":
                out.append((at + off, size))
    return out


@pytest.mark.skipif(not os.path.exists(READELF), reason="no llvm-readelf")
def test_implicit_arguments_sit_where_the_dispatcher_puts_them():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    data = open(LIB, "rb").read()
    cos = code_objects(data)
    assert len(cos) >= 5, cos       # one per translation unit with kernels
    n_kernels = n_hidden = 0
    names = set()
    with tempfile.TemporaryDirectory() as tmp:
        for i, (off, size) in enumerate(cos):
            path = os.path.join(tmp, "co%d.hsaco" % i)
            open(path, "wb").write(data[off:off + size])
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
            for k in re.split(r"\n  - \.", notes):
                if "kernarg_segment_size" not in k:
                    continue
                n_kernels += 1
                name = re.search(r"\n    \.name:\s+(\S+)", k).group(1)
                names.add(name)
                seg = int(re.search(r"kernarg_segment_size:\s+(\d+)", k).group(1))
                assert seg <= 2048, (name, seg)      # BSR_AQL_KERNARG_BYTES
                explicit_end, hidden = 0, {}
                for o, s, kind in re.findall(r"\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)\n\s+\.value_kind:\s+(\w+)", k):
                    if kind.startswith("hidden"):
                        hidden[kind] = int(o)
                    else:
                        explicit_end = max(explicit_end, int(o) + int(s))
                if not hidden:
                    continue
                n_hidden += 1
                base = (explicit_end + 7) // 8 * 8
                assert base + 256 <= 2048, (name, base)
                for kind, o in hidden.items():
                    assert kind in IMPLICIT, (name, kind)           # nothing the dispatcher does not know about
                    assert o - base == IMPLICIT[kind], (name, kind, o, base)
                    assert kind in FILLED or kind in ZERO_IS_RIGHT, (name, kind)
    assert n_kernels > 100 and n_hidden > 50
    # the kernels of a scoring batch are all there under the names the HIP runtime reports for their host stubs
    for frag in ("k_tile1aILi3E", "k_solve", "k_finalize", "k_events", "k_rowsIdLi3E", "k_streamILi3E"):
        assert any(frag in n for n in names), frag


def test_the_library_exports_the_dispatch_report():
    import ctypes
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    L = ctypes.CDLL(LIB)
    assert hasattr(L, "bsr_dispatch_info")
