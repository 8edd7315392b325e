"""Sharding of independent chains over the GPUs of a node and the one exchange of the path: an all-gather of the
chains' accepted trees.

A "chain" is one pass of the reference's restart loop (codes/bsr_class.py:99): fresh sigma, fresh K trees, no data
flowing between chains except the final append to ROOTS/BETAS/trainERRS (codes/bsr_class.py:270-273).  Chain c runs
on rank c % world with its own RNG stream (np.random.seed(seed_base + c)), so results do not depend on the world
size.  The gather moves the chains' records -- K tapes at full length, Beta, counters, the whole RMSE history --
in two collectives (sizes, then payloads): a few KB per rank for the trees the sampler visits, latency-bound.
"""
import os

import numpy as np

from .tape import NODE_DTYPE, flatten, unflatten

MAX_K = 8
HEADER_I32 = 16 + MAX_K       # chain id, K, n_props, n_accept, n_errs, 0, rank rejects, discarded, 0, record bytes, ..., tape lengths
HEADER_F64 = 4 + (MAX_K + 1)  # sigma, last rmse, best rmse, spare, Beta[K+1]
HEADER_BYTES = HEADER_I32 * 4 + HEADER_F64 * 8
assert HEADER_BYTES % 8 == 0


def shard(n_chains, world, rank):
    """Chain ids owned by `rank` (round-robin)."""
    return [c for c in range(n_chains) if c % world == rank]


def pack_record(chain_id, roots, beta, sigma, errs, n_props, n_accept, n_rank_rejects=0, n_discarded=0,
                tapes_in=None):
    """One chain's outcome as bytes: a fixed header, then the WHOLE per-accept RMSE history (train_err_,
    codes/bsr_class.py:233, 270) and the K trees' postfix tapes at their full length -- what ROOTS.append(Roots) keeps
    (codes/bsr_class.py:270-273), whatever its size (roots: Node trees, or tapes_in: their tapes)."""
    K = len(roots) if tapes_in is None else len(tapes_in)
    if K > MAX_K:
        raise ValueError("a record holds at most %d trees" % MAX_K)
    hi = np.zeros(HEADER_I32, dtype=np.int32)
    hf = np.zeros(HEADER_F64, dtype=np.float64)
    he = np.ascontiguousarray(errs, dtype=np.float64).reshape(-1)
    tapes = [np.ascontiguousarray(flatten(roots[k]) if tapes_in is None else tapes_in[k], dtype=NODE_DTYPE).reshape(-1)
             for k in range(K)]
    hi[0], hi[1], hi[2], hi[3], hi[4] = chain_id, K, n_props, n_accept, he.size
    hi[6], hi[7] = n_rank_rejects, n_discarded
    for k in range(K):
        hi[16 + k] = len(tapes[k])
    hf[0] = sigma
    hf[1] = he[-1] if he.size else np.nan
    hf[2] = he.min() if he.size else np.nan
    b = np.asarray(beta, dtype=np.float64).reshape(-1)
    hf[4:4 + len(b)] = b
    total = HEADER_BYTES + he.size * 8 + sum(len(t) for t in tapes) * NODE_DTYPE.itemsize
    if total >= 2 ** 31:
        raise ValueError("chain record of %d bytes" % total)
    hi[9] = total
    out = np.concatenate([hi.view(np.uint8), hf.view(np.uint8), he.view(np.uint8)] + [t.view(np.uint8) for t in tapes])
    assert out.size == total and total % 8 == 0
    return out


def pack_chain_record(chain):
    """Record of a live bsr.chain.Chain (its current, i.e. last accepted, trees)."""
    return pack_record(chain.index, chain.roots, chain.Beta, chain.sigma, chain.errs, chain.n_props, chain.n_accept)


def record_bytes(buf):
    """Length of the record that starts at buf[0] (from its header)."""
    return int(np.ascontiguousarray(buf[:HEADER_I32 * 4], dtype=np.uint8).view(np.int32)[9])


def unpack_record(buf):
    """The record that starts at buf[0].  Every length in the header is checked against the buffer (ValueError on a
    short or corrupt record, as split_records raises: no assert that `python -O` would drop, no silently truncated
    slice)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    if buf.size < HEADER_BYTES:
        raise ValueError("chain record of %d bytes is shorter than its header (%d)" % (buf.size, HEADER_BYTES))
    ni = HEADER_I32 * 4
    hi = buf[:ni].view(np.int32)
    hf = buf[ni:HEADER_BYTES].view(np.float64)
    K, n_errs, total = int(hi[1]), int(hi[4]), int(hi[9])
    if not 0 <= K <= MAX_K:
        raise ValueError("corrupt chain record: K = %d (0..%d)" % (K, MAX_K))
    if n_errs < 0 or any(int(hi[16 + k]) < 0 for k in range(K)):
        raise ValueError("corrupt chain record: negative history or tape length")
    want = HEADER_BYTES + n_errs * 8 + sum(int(hi[16 + k]) for k in range(K)) * NODE_DTYPE.itemsize
    if want != total or total > buf.size:
        raise ValueError("corrupt chain record: header says %d bytes, its parts add up to %d, buffer holds %d"
                         % (total, want, buf.size))
    at = HEADER_BYTES
    he = buf[at:at + n_errs * 8].view(np.float64)
    at += n_errs * 8
    roots, lens, tapes = [], [], []
    for k in range(K):
        n = int(hi[16 + k])
        t = buf[at:at + n * NODE_DTYPE.itemsize].view(NODE_DTYPE).copy()
        at += n * NODE_DTYPE.itemsize
        lens.append(n)
        tapes.append(t)
        roots.append(unflatten(t) if n > 0 else None)
    return {"chain": int(hi[0]), "K": K, "n_props": int(hi[2]), "n_accept": int(hi[3]), "n_errs": n_errs,
            "n_rank_rejects": int(hi[6]), "n_discarded": int(hi[7]),
            "errs": [float(v) for v in he], "errs_truncated": False,      # (both kept for callers of the fixed-size
            "truncated": 0, "tape_len": lens, "sigma": float(hf[0]),      #  records of rounds 1-4: nothing is cut now)
            "last_rmse": float(hf[1]), "best_rmse": float(hf[2]), "beta": hf[4:4 + K + 1].copy().reshape(-1, 1),
            "roots": roots, "tapes": tapes}


def split_records(payload):
    """The records packed back to back in `payload` (uint8), in order."""
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out, at = [], 0
    while at + HEADER_BYTES <= payload.size:
        n = record_bytes(payload[at:])
        if n < HEADER_BYTES or at + n > payload.size:
            raise ValueError("corrupt chain record at byte %d" % at)
        out.append(payload[at:at + n])
        at += n
    if at != payload.size:
        raise ValueError("%d stray bytes behind the last chain record" % (payload.size - at))
    return out


def save_records(path, recs):
    """Records of any size in one .npz: the bytes back to back and where each one starts."""
    sizes = np.array([r.size for r in recs], dtype=np.int64)
    flat = np.concatenate(recs) if len(recs) else np.zeros(0, dtype=np.uint8)
    with open(path, "wb") as f:
        np.savez(f, bytes=flat, sizes=sizes)


def load_records(path):
    with np.load(path) as z:
        flat, sizes = z["bytes"], z["sizes"]
    out, at = [], 0
    for n in sizes:
        out.append(flat[at:at + int(n)])
        at += int(n)
    return out


class RcclGather:
    """All-gather through the C ABI's RCCL communicator (bsr_comm_*), device buffers over xGMI."""

    def __init__(self, ctx, world, rank, uid):
        self.ctx = ctx
        self._world = world
        ctx.comm_init(world, rank, uid)

    def world(self):
        return self._world

    def allgather(self, send):
        return self.ctx.comm_allgather(send)


class FileGather:
    """All-gather through the rendezvous directory (bsr.launch.Rendezvous).  Test scaffolding for running several
    ranks on ONE device, where RCCL refuses to build a communicator; selected only by BSR_SHARE_DEVICE=1."""

    def __init__(self, rdv):
        self.rdv = rdv

    def world(self):
        return self.rdv.world

    def allgather(self, send):
        send = np.ascontiguousarray(send, dtype=np.uint8)
        return np.stack([np.frombuffer(b, dtype=np.uint8) for b in self.rdv.allgather(send.tobytes())])


class SoloGather:
    """World of one: the gather is the identity."""

    def world(self):
        return 1

    def allgather(self, send):
        return np.ascontiguousarray(send, dtype=np.uint8).reshape(1, -1)


def local_device(local_rank, n_visible=None):
    """HIP device index of a local rank.  One process per GPU sees either all GPUs of the node (device = LOCAL_RANK) or --
    a launcher that narrows HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank -- just its own, renumbered from 0:
    LOCAL_RANK modulo the number of visible devices serves both (and refuses a process that sees none)."""
    if n_visible is None:
        import ctypes as C
        from . import _lib
        n = C.c_int(0)
        _lib.check(_lib.lib().bsr_device_count(C.byref(n)), None)
        n_visible = n.value
    if n_visible <= 0:
        raise RuntimeError("no HIP device visible to local rank %d (HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r)"
                           % (local_rank, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
    return int(local_rank) % int(n_visible)


_N_CONNECTS = 0


def connect(ctx, rank, world, rdv=None):
    """The node's gather for this rank: RCCL through the C ABI (unique id published by rank 0 through the rendezvous
    directory), or the identity for a world of one.  Returns (gather, rendezvous)."""
    if world <= 1:
        return SoloGather(), None
    from .launch import Rendezvous
    rdv = rdv or Rendezvous(rank, world)
    if os.environ.get("BSR_SHARE_DEVICE") == "1":
        g = FileGather(rdv)
        g.allgather(np.zeros(8, dtype=np.uint8))
        rdv.mark_up()
        return g, rdv
    global _N_CONNECTS
    name = "uid" if _N_CONNECTS == 0 else "uid%d" % _N_CONNECTS      # every rank connects in the same order
    _N_CONNECTS += 1
    uid = rdv.broadcast(name, lambda: ctx.comm_unique_id().tobytes(), nbytes=128)
    g = RcclGather(ctx, world, rank, np.frombuffer(uid, dtype=np.uint8))
    g.allgather(np.zeros(8, dtype=np.uint8))      # first collective: everyone holds the id now
    rdv.mark_up()
    rdv.close()
    return g, rdv


def barrier(gather):
    gather.allgather(np.zeros(8, dtype=np.uint8))


def allreduce_max(gather, value):
    got = gather.allgather(np.array([value], dtype=np.float64).view(np.uint8))
    return float(np.max(got.reshape(-1).view(np.float64)))


def gather_raw(gather, local_records, chains_per_rank=None):
    """Every rank's records, ordered by chain id: a list of uint8 arrays (decode with unpack_record).

    Two collectives, because a record is as long as its trees and its RMSE history are (a depth-12 tree has up to
    8 191 nodes, BASELINE configs[4]): first every rank's payload size, then the payloads padded to the largest of
    them -- what travels is proportional to what the chains hold, and nothing is cut (codes/bsr_class.py:270-276 keeps
    whatever ROOTS.append(Roots) holds).  `chains_per_rank` is accepted for the callers of the fixed-size gather of
    rounds 1-4 and not needed."""
    recs = [np.ascontiguousarray(r, dtype=np.uint8) for r in local_records]
    mine = np.concatenate(recs) if recs else np.zeros(0, dtype=np.uint8)
    sizes = gather.allgather(np.array([mine.size], dtype=np.int64).view(np.uint8)).reshape(-1).view(np.int64)
    widest = int(sizes.max()) if sizes.size else 0
    if widest == 0:
        return []
    send = np.zeros((widest + 7) // 8 * 8, dtype=np.uint8)
    send[:mine.size] = mine
    got = gather.allgather(send)
    out = []
    for r in range(got.shape[0]):
        for rec in split_records(got[r, :int(sizes[r])]):
            cid = int(rec[:4].view(np.int32)[0])
            if cid >= 0:
                out.append((cid, rec.copy()))
    out.sort(key=lambda t: t[0])
    return [rec for _, rec in out]


def gather_chains(gather, local_records, chains_per_rank=None):
    """gather_raw, unpacked: list of dicts ordered by chain id."""
    return [unpack_record(r) for r in gather_raw(gather, local_records, chains_per_rank)]
