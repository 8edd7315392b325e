"""DeviceContext: numpy-facing wrapper of one bsr_ctx (training data + chain caches on one MI355X)."""
import ctypes as C

import numpy as np

from . import _lib
from .tape import NODE_DTYPE, pack


class DeviceContext:
    """Owns one bsr_ctx.  X: (N,d) array-like, y: (N,) or None.  Not thread-safe (one per device per thread)."""

    def __init__(self, X, y=None, K=0, n_chains=0, max_batch=64, device=0, dtype="f64", typical_chains=0, typical_batch=0):
        """typical_chains / typical_batch: what a batch of this caller looks like as a rule (distinct chains, proposals),
        where that is less than n_chains / max_batch -- the row pass's geometry is then chosen for those batches
        (bsr_ctx_create_tuned; wider batches still score, to the same bytes)."""
        L = _lib.lib()
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        if X.ndim != 2:
            raise ValueError("X must be 2-D")
        self.N, self.d = X.shape
        self.K, self.n_chains, self.max_batch = int(K), int(n_chains), int(max_batch)
        self.device = device
        yv = None if y is None else np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        if yv is not None and yv.shape[0] != self.N:
            raise ValueError("y has %d rows, X has %d" % (yv.shape[0], self.N))
        self._h = C.c_void_p()
        dt = _lib.DTYPE_F64 if dtype in ("f64", "float64") else _lib.DTYPE_F32
        if typical_chains or typical_batch:
            rc = L.bsr_ctx_create_tuned(C.byref(self._h), device, self.N, self.d, _lib.ptr(X),
                                        None if yv is None else _lib.ptr(yv), self.K, self.n_chains, self.max_batch, dt,
                                        int(typical_chains), int(typical_batch))
        else:
            rc = L.bsr_ctx_create(C.byref(self._h), device, self.N, self.d, _lib.ptr(X),
                                  None if yv is None else _lib.ptr(yv), self.K, self.n_chains, self.max_batch, dt)
        _lib.check(rc, None)
        self._L = L
        self._mh_pending = {}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.bsr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- allcal
    def eval_tapes(self, tapes, want_cols=True):
        """tapes: list of NODE_DTYPE arrays -> (cols (n,N) or None, maxabs (n,), flags (n,))."""
        n = len(tapes)
        rows, off = pack(tapes)
        cols = np.empty((n, self.N), dtype=np.float64) if want_cols else None
        maxabs = np.empty(n, dtype=np.float64)
        flags = np.empty(n, dtype=np.uint32)
        rc = self._L.bsr_eval_tapes(self._h, _lib.ptr(rows), _lib.ptr(off), n,
                                    None if cols is None else _lib.ptr(cols), _lib.ptr(maxabs), _lib.ptr(flags))
        _lib.check(rc, self._h)
        return cols, maxabs, flags

    # ---- chain state
    def set_current(self, chain, k, tape):
        tape = np.ascontiguousarray(tape, dtype=NODE_DTYPE)
        _lib.check(self._L.bsr_set_current(self._h, chain, k, _lib.ptr(tape), len(tape)), self._h)

    def commit(self, chain, k, slot):
        _lib.check(self._L.bsr_commit(self._h, chain, k, slot), self._h)

    def refresh(self, chain):
        info = _lib.ChainInfo()
        _lib.check(self._L.bsr_refresh(self._h, chain, C.byref(info)), self._h)
        K = self.K
        return {"sse_old": info.sse_old, "scale_old": info.scale_old, "maxabs": list(info.maxabs)[:K],
                "beta_old": list(info.beta_old)[:K], "colflags": list(info.colflags)[:K]}

    def score_batch(self, tapes, chains, ks, sigmas):
        B = len(tapes)
        rows, off = pack(tapes)
        chains = np.ascontiguousarray(chains, dtype=np.int32)
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        sig = np.ascontiguousarray(sigmas, dtype=np.float64)
        out = np.zeros(B, dtype=_lib.SCORE_DTYPE)
        rc = self._L.bsr_score_batch(self._h, _lib.ptr(rows), _lib.ptr(off), _lib.ptr(chains), _lib.ptr(ks),
                                     _lib.ptr(sig), B, _lib.ptr(out))
        _lib.check(rc, self._h)
        return out

    def score_packed(self, rows, off, chains, ks, sig, out):
        """Same as score_batch with pre-packed, reusable arrays (bench / chain engine inner loop)."""
        rc = self._L.bsr_score_batch(self._h, _lib.ptr(rows), _lib.ptr(off), _lib.ptr(chains), _lib.ptr(ks),
                                     _lib.ptr(sig), len(chains), _lib.ptr(out))
        _lib.check(rc, self._h)
        return out

    def score_submit(self, rows, off, chains, ks, sig):
        """Asynchronous scoring: enqueue a pre-packed batch, return its ticket (up to BSR_MAX_INFLIGHT = 8 batches may be in flight)."""
        t = C.c_int32(-1)
        rc = self._L.bsr_score_submit(self._h, _lib.ptr(rows), _lib.ptr(off), _lib.ptr(chains), _lib.ptr(ks),
                                      _lib.ptr(sig), len(chains), C.byref(t))
        _lib.check(rc, self._h)
        return t.value

    def prepare(self, rows, off, chains, ks, sig):
        """Pins down the five input arrays of a batch once (addresses resolved, references kept) for callers that submit
        the same pre-packed batch many times or want the submission itself as cheap as possible."""
        arrs = (np.ascontiguousarray(rows), np.ascontiguousarray(off, dtype=np.int32),
                np.ascontiguousarray(chains, dtype=np.int32), np.ascontiguousarray(ks, dtype=np.int32),
                np.ascontiguousarray(sig, dtype=np.float64))
        return tuple(a.ctypes.data for a in arrs) + (len(arrs[2]), arrs)

    def score_submit_prepared(self, prep):
        t = C.c_int32(-1)
        rc = self._L.bsr_score_submit(self._h, prep[0], prep[1], prep[2], prep[3], prep[4], prep[5], C.byref(t))
        if rc:
            _lib.check(rc, self._h)
        return t.value

    def score_wait(self, ticket, out):
        _lib.check(self._L.bsr_score_wait(self._h, ticket, _lib.ptr(out)), self._h)
        return out

    def score_wait_ptr(self, ticket, out_ptr):
        """score_wait into a result array whose address the caller resolved once (`out.ctypes.data`)."""
        rc = self._L.bsr_score_wait(self._h, ticket, out_ptr)
        if rc:
            _lib.check(rc, self._h)

    def score_submit_mh(self, rows, off, chains, ks, sig, terms8, flags, span_off):
        """Scoring plus the device-side MH step (codes/funcs.py:1226-1306 on the device; include/bsr_hip.h):
        terms8 (B, 8) float64, flags (B,) int32, span_off (n_spans + 1,) int32 -> ticket."""
        terms8 = np.ascontiguousarray(terms8, dtype=np.float64)
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        span_off = np.ascontiguousarray(span_off, dtype=np.int32)
        B = len(chains)
        if terms8.shape != (B, 8) or flags.shape != (B,) or span_off.ndim != 1 or len(span_off) < 2 or \
                span_off[0] != 0 or span_off[-1] != B:
            raise ValueError("score_submit_mh: terms8 must be (B, 8), flags (B,), span_off from 0 to B = %d" % B)
        t = C.c_int32(-1)
        rc = self._L.bsr_score_submit_mh(self._h, _lib.ptr(rows), _lib.ptr(off), _lib.ptr(chains), _lib.ptr(ks),
                                         _lib.ptr(sig), len(chains), _lib.ptr(terms8), _lib.ptr(flags),
                                         _lib.ptr(span_off), len(span_off) - 1, C.byref(t))
        _lib.check(rc, self._h)
        # up to BSR_MAX_INFLIGHT MH batches may be in flight: what a ticket returns is sized by ITS submission
        self._mh_pending[t.value] = (len(span_off) - 1, B)
        return t.value

    def score_wait_mh(self, ticket, out=None):
        if ticket not in self._mh_pending:
            raise _lib.BsrError(-6, "score_wait_mh: no MH batch under ticket %r" % (ticket,))
        n_spans, B = self._mh_pending.pop(ticket)
        if out is not None and (out.dtype != _lib.SCORE_DTYPE or out.shape[0] < B or not out.flags["C_CONTIGUOUS"]):
            raise ValueError("score_wait_mh: `out` must be a contiguous SCORE_DTYPE array of at least %d records" % B)
        ev = np.zeros(n_spans, dtype=_lib.EVENT_DTYPE)
        rc = self._L.bsr_score_wait_mh(self._h, ticket, None if out is None else _lib.ptr(out), _lib.ptr(ev))
        _lib.check(rc, self._h)
        return ev

    def fit_beta(self, chain):
        beta = np.empty(self.K + 1, dtype=np.float64)
        rmse = C.c_double(0.0)
        _lib.check(self._L.bsr_fit_beta(self._h, chain, _lib.ptr(beta), C.byref(rmse)), self._h)
        return beta.reshape(-1, 1), rmse.value

    def get_current(self, chain):
        out = np.empty((self.K, self.N), dtype=np.float64)
        _lib.check(self._L.bsr_get_current(self._h, chain, _lib.ptr(out)), self._h)
        return out

    # ---- profiling
    def set_profiling(self, level=1):
        """0 off, 1 events around the row pass only, 2 events around every kernel (diagnostic)."""
        _lib.check(self._L.bsr_set_profiling(self._h, int(level)), self._h)

    def info(self):
        """What the context decided for this process and machine (bsr_ctx_info)."""
        v = np.zeros(8, dtype=np.int32)
        _lib.check(self._L.bsr_ctx_info(self._h, _lib.ptr(v)), self._h)
        pl = np.zeros(4, dtype=np.int32)
        _lib.check(self._L.bsr_place_info(_lib.ptr(pl)), self._h)
        return {"gpu_numa_node": int(pl[2]),"submit_threads": int(v[0]), "lib_cpus": int(v[1]), "caller_pinned": bool(v[2]),
                "cpu_budget": v[3] / 100.0, "tape_groups": int(v[4]), "row_slices": int(v[5]),
                "blocks_per_slice": int(v[6]), "slices_whole": int(v[7]) in (1, 3), "streaming": int(v[7]) == 2,
                "row_pass": {1: "k_tile1", 2: "k_stream", 3: "k_tile1a"}.get(int(v[7]), "k_tile/k_rows")}

    def dispatch_info(self):
        """How scoring batches reach the GPU (bsr_dispatch_info): {"direct": the context writes AQL packets into its own
        queues, "queues", "batches_direct", "batches_streamed"}."""
        v = np.zeros(8, dtype=np.int64)
        _lib.check(self._L.bsr_dispatch_info(self._h, _lib.ptr(v)), self._h)
        return {"direct": bool(v[0]), "queues": int(v[1]), "batches_direct": int(v[2]), "batches_streamed": int(v[3])}

    def batch_stats(self, ticket=0):
        """How the last waited batch of `ticket` was scored (bsr_batch_stats): tapes, of which the assembly interpreter's,
        of which chains; stream entries after the fusions."""
        v = np.zeros(4, dtype=np.int32)
        _lib.check(self._L.bsr_batch_stats(self._h, int(ticket), _lib.ptr(v)), self._h)
        return {"tapes": int(v[0]), "asm_program_tapes": int(v[1]), "chain_tapes": int(v[2]), "stream_entries": int(v[3])}

    def last_timing(self):
        us = np.zeros(5, dtype=np.float64)
        _lib.check(self._L.bsr_last_timing(self._h, _lib.ptr(us)), self._h)
        return us

    # ---- RCCL
    @staticmethod
    def comm_unique_id():
        buf = np.zeros(_lib.COMM_ID_BYTES, dtype=np.uint8)
        _lib.check(_lib.lib().bsr_comm_unique_id(_lib.ptr(buf)), None)
        return buf

    def comm_init(self, nranks, rank, uid):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        _lib.check(self._L.bsr_comm_init(self._h, nranks, rank, _lib.ptr(uid)), self._h)
        self._nranks = nranks

    def comm_allgather(self, send_bytes):
        send = np.ascontiguousarray(send_bytes, dtype=np.uint8)
        recv = np.empty(send.size * self._nranks, dtype=np.uint8)
        _lib.check(self._L.bsr_comm_allgather(self._h, _lib.ptr(send), _lib.ptr(recv), send.size), self._h)
        return recv.reshape(self._nranks, send.size)


def yloglike_device(y, outputs, sigma, skipna=True, device=0):
    """ylogLike(y, outputs, sigma) computed on the GPU (codes/funcs.py:1147-1174) -> dict."""
    L = _lib.lib()
    O = np.ascontiguousarray(np.asarray(outputs, dtype=np.float64))
    yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    N, K = O.shape
    ll, sse, scale = C.c_double(), C.c_double(), C.c_double()
    beta = np.zeros(_lib.MAX_K, dtype=np.float64)
    rank = C.c_int32()
    rc = L.bsr_yloglike_host(device, N, K, _lib.ptr(O), _lib.ptr(yv), float(sigma), 1 if skipna else 0,
                             C.byref(ll), C.byref(sse), C.byref(scale), _lib.ptr(beta), C.byref(rank))
    _lib.check(rc, None)
    return {"loglik": ll.value, "sse": sse.value, "scale": scale.value, "beta": beta[:K].copy(), "rank": rank.value}
