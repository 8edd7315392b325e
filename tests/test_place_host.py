"""csrc/bsr_place.h on the host (CPU): a rank's threads go to an L3 domain of ITS GPU's NUMA node
(/sys/bus/pci/devices/<bdf>/numa_node), not to whatever LOCAL_RANK arithmetic over the CPU numbering lands on
(unverifiable on a 1-GPU box -- so it is verified on a faked sysfs tree: two sockets, four L3 domains of
eight cores with their SMT siblings each per socket, GPUs whose LOCAL_RANK order does NOT follow the socket order)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("place") / "libplace.so")
    src = os.path.join(ROOT, "tests", "native", "place_shim.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", out, src], check=True)
    L = C.CDLL(out)
    L.place_pick.restype = C.c_int
    L.place_pick.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                             C.POINTER(C.c_int)]
    return L


def fake_sysfs(root, gpu_numa):
    """Two NUMA nodes of 32 cores (+32 SMT siblings numbered 64..127), L3 domains of 8 cores + their siblings."""
    def l3(cpu):
        core = cpu % 64
        base = core - core % 8
        return "%d-%d,%d-%d" % (base, base + 7, base + 64, base + 71)
    for cpu in range(128):
        d = root / ("sys/devices/system/cpu/cpu%d/cache/index3" % cpu)
        d.mkdir(parents=True)
        (d / "shared_cpu_list").write_text(l3(cpu) + "\n")
    for node, lst in ((0, "0-31,64-95"), (1, "32-63,96-127")):
        d = root / ("sys/devices/system/node/node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(lst + "\n")
    for bdf, numa in gpu_numa.items():
        d = root / "sys/bus/pci/devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % numa)


def pick(L, root, bdf, allowed, lr, lw, cur):
    out = (C.c_int * 256)()
    numa = C.c_int(-9)
    n = L.place_pick(str(root).encode(), bdf.encode(), allowed.encode(), lr, lw, cur, out, 256, C.byref(numa))
    return [out[i] for i in range(max(n, 0))], numa.value


def test_a_ranks_threads_sit_on_its_gpus_numa_node(shim, tmp_path):
    # LOCAL_RANK 0..3 on socket 1, 4..7 on socket 0: the opposite of what dealing domains by rank over CPU order assumes
    gpus = {"0000:%02x:00.0" % (0x10 + r): (1 if r < 4 else 0) for r in range(8)}
    fake_sysfs(tmp_path, gpus)
    all_cpus = "0-127"
    seen = {}
    for r in range(8):
        cpus, numa = pick(shim, tmp_path, "0000:%02X:00.0" % (0x10 + r), all_cpus, r, 8, 5)   # (upper case as HIP prints it)
        assert numa == gpus["0000:%02x:00.0" % (0x10 + r)]
        node_cpus = set(range(32, 64)) | set(range(96, 128)) if numa == 1 else set(range(0, 32)) | set(range(64, 96))
        assert len(cpus) == 16 and set(cpus) <= node_cpus, (r, cpus)
        seen.setdefault(numa, []).append(tuple(cpus))
    # the four ranks of a socket take its four L3 domains, one each
    assert all(len(set(v)) == 4 for v in seen.values())


def test_single_rank_unknown_node_and_narrow_affinity(shim, tmp_path):
    fake_sysfs(tmp_path, {"0000:05:00.0": 1, "0000:06:00.0": -1})
    # one rank: the domain it is running in when that one is on the GPU's node ...
    cpus, numa = pick(shim, tmp_path, "0000:05:00.0", "0-127", -1, 1, 45)
    assert numa == 1 and 45 in cpus and len(cpus) == 16
    # ... else the node's first domain
    cpus, _ = pick(shim, tmp_path, "0000:05:00.0", "0-127", -1, 1, 3)
    assert cpus[0] == 32 and len(cpus) == 16
    # the platform does not say (numa_node = -1, or no such device): any allowed domain, the current one first
    cpus, numa = pick(shim, tmp_path, "0000:06:00.0", "0-127", -1, 1, 3)
    assert numa == -1 and 3 in cpus
    cpus, numa = pick(shim, tmp_path, "0000:77:00.0", "0-127", -1, 1, 70)
    assert numa == -1 and 70 in cpus
    # a cgroup that allows nothing on the GPU's node: the allowed CPUs, not an empty set
    cpus, numa = pick(shim, tmp_path, "0000:05:00.0", "0-15", -1, 1, 2)
    assert numa == 1 and set(cpus) <= set(range(16)) and len(cpus) == 8
    # two CPUs are not worth a placement
    cpus, _ = pick(shim, tmp_path, "0000:05:00.0", "0-1", -1, 1, 0)
    assert cpus == []


def test_ranks_whose_gpus_node_is_unknown_spread_over_all_domains(shim, tmp_path):
    """numa_node = -1 (VMs, containers): eight ranks take eight DIFFERENT L3 domains spread over both sockets -- dealt by
    rank modulo the number of domains they would sit on the first domains of socket 0 only."""
    fake_sysfs(tmp_path, {"0000:%02x:00.0" % (0x10 + r): -1 for r in range(8)})
    doms = []
    for r in range(8):
        cpus, numa = pick(shim, tmp_path, "0000:%02x:00.0" % (0x10 + r), "0-127", r, 8, 5)
        assert numa == -1 and len(cpus) == 16
        doms.append(tuple(cpus))
    assert len(set(doms)) == 8
    firsts = sorted(d[0] for d in doms)
    assert any(32 <= f < 64 for f in firsts) and any(f < 32 for f in firsts)      # both sockets' first halves in use
    # four ranks: every other domain
    four = [tuple(pick(shim, tmp_path, "0000:%02x:00.0" % (0x10 + r), "0-127", r, 4, 5)[0]) for r in range(4)]
    assert len(set(four)) == 4
