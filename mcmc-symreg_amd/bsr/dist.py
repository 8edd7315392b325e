"""Sharding of independent chains over the GPUs of a node and the one exchange of the path: an all-gather of the
chains' accepted trees.

A "chain" is one pass of the reference's restart loop (codes/bsr_class.py:99): fresh sigma, fresh K trees, no data
flowing between chains except the final append to ROOTS/BETAS/trainERRS (codes/bsr_class.py:270-273).  Chain c runs
on rank c % world with its own RNG stream (np.random.seed(seed_base + c)), so results do not depend on the world
size.  The gather moves fixed-size records (K padded tapes + Beta + counters), tens of KB per rank: latency-bound.
"""
import os

import numpy as np

from .tape import NODE_DTYPE, flatten, unflatten

MAX_K = 8
RECORD_NODES = 255            # nodes kept per tree in a record (longer tapes are flagged, not sent)
HEADER_I32 = 16 + MAX_K       # chain id, K, n_props, n_accept, n_errs, truncated, ... , tape lengths
HEADER_F64 = 4 + (MAX_K + 1)  # sigma, last rmse, best rmse, spare, Beta[K+1]
ERRS_CAP = 1024               # per-accept RMSE history kept in a record (train_err_, codes/bsr_class.py:233, 270)
RECORD_BYTES = HEADER_I32 * 4 + HEADER_F64 * 8 + ERRS_CAP * 8 + MAX_K * RECORD_NODES * NODE_DTYPE.itemsize
assert RECORD_BYTES % 8 == 0


def shard(n_chains, world, rank):
    """Chain ids owned by `rank` (round-robin)."""
    return [c for c in range(n_chains) if c % world == rank]


def pack_record(chain_id, roots, beta, sigma, errs, n_props, n_accept, n_rank_rejects=0, n_discarded=0,
                tapes_in=None):
    """One chain's outcome as RECORD_BYTES bytes (roots: Node trees, or tapes_in: their postfix tapes)."""
    K = len(roots) if tapes_in is None else len(tapes_in)
    hi = np.zeros(HEADER_I32, dtype=np.int32)
    hf = np.zeros(HEADER_F64, dtype=np.float64)
    he = np.zeros(ERRS_CAP, dtype=np.float64)
    tapes = np.zeros((MAX_K, RECORD_NODES), dtype=NODE_DTYPE)
    hi[0], hi[1], hi[2], hi[3], hi[4] = chain_id, K, n_props, n_accept, len(errs)
    hi[6], hi[7] = n_rank_rejects, n_discarded
    hi[8] = 1 if len(errs) > ERRS_CAP else 0     # the RMSE history below is the LAST ERRS_CAP entries of a longer one
    he[:min(len(errs), ERRS_CAP)] = np.asarray(errs, dtype=np.float64)[-ERRS_CAP:] if len(errs) else []
    for k in range(K):
        t = flatten(roots[k]) if tapes_in is None else tapes_in[k]
        if len(t) > RECORD_NODES:
            hi[5] |= (1 << k)
            hi[16 + k] = -len(t)
        else:
            hi[16 + k] = len(t)
            tapes[k, :len(t)] = t
    hf[0] = sigma
    hf[1] = errs[-1] if len(errs) else np.nan
    hf[2] = min(errs) if len(errs) else np.nan
    b = np.asarray(beta, dtype=np.float64).reshape(-1)
    hf[4:4 + len(b)] = b
    out = np.concatenate([hi.view(np.uint8), hf.view(np.uint8), he.view(np.uint8), tapes.reshape(-1).view(np.uint8)])
    assert out.size == RECORD_BYTES
    return out


def pack_chain_record(chain):
    """Record of a live bsr.chain.Chain (its current, i.e. last accepted, trees)."""
    return pack_record(chain.index, chain.roots, chain.Beta, chain.sigma, chain.errs, chain.n_props, chain.n_accept)


def unpack_record(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    assert buf.size == RECORD_BYTES
    ni = HEADER_I32 * 4
    nf = HEADER_F64 * 8
    ne = ERRS_CAP * 8
    hi = buf[:ni].view(np.int32)
    hf = buf[ni:ni + nf].view(np.float64)
    he = buf[ni + nf:ni + nf + ne].view(np.float64)
    tapes = buf[ni + nf + ne:].view(NODE_DTYPE).reshape(MAX_K, RECORD_NODES)
    K = int(hi[1])
    roots, lens = [], []
    for k in range(K):
        n = int(hi[16 + k])
        lens.append(n)
        roots.append(unflatten(tapes[k, :n]) if n > 0 else None)
    return {"chain": int(hi[0]), "K": K, "n_props": int(hi[2]), "n_accept": int(hi[3]), "n_errs": int(hi[4]),
            "n_rank_rejects": int(hi[6]), "n_discarded": int(hi[7]),
            "errs": [float(v) for v in he[:min(int(hi[4]), ERRS_CAP)]], "errs_truncated": bool(hi[8]),
            "truncated": int(hi[5]), "tape_len": lens, "sigma": float(hf[0]), "last_rmse": float(hf[1]),
            "best_rmse": float(hf[2]), "beta": hf[4:4 + K + 1].copy().reshape(-1, 1), "roots": roots,
            "tapes": [tapes[k, :max(0, lens[k])].copy() for k in range(K)]}


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


def gather_raw(gather, local_records, chains_per_rank):
    """local_records: list of packed records (padded to chains_per_rank with empty records).
    Returns every rank's records as a uint8 array (n_chains, RECORD_BYTES) ordered by chain id."""
    recs = list(local_records)
    empty = np.zeros(RECORD_BYTES, dtype=np.uint8)
    empty[:4] = np.array([-1], dtype=np.int32).view(np.uint8)
    while len(recs) < chains_per_rank:
        recs.append(empty)
    got = gather.allgather(np.concatenate(recs))
    out = []
    for r in range(got.shape[0]):
        for i in range(chains_per_rank):
            rec = got[r, i * RECORD_BYTES:(i + 1) * RECORD_BYTES]
            cid = int(rec[:4].view(np.int32)[0])
            if cid >= 0:
                out.append((cid, rec))
    out.sort(key=lambda t: t[0])
    return np.stack([rec for _, rec in out]) if out else np.zeros((0, RECORD_BYTES), dtype=np.uint8)


def gather_chains(gather, local_records, chains_per_rank):
    """gather_raw, unpacked: list of dicts ordered by chain id."""
    raw = gather_raw(gather, local_records, chains_per_rank)
    return [unpack_record(raw[i]) for i in range(raw.shape[0])]


