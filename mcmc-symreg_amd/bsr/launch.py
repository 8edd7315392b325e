"""One process per GPU without a launcher framework: spawn, rendezvous and the host-side exchange of the 128-byte
RCCL unique id.

Two ways a rank comes to life:
  * `spawn(n, argv)`: the caller (which must not have touched the GPU itself) starts n fresh Python processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / BSR_RDV_DIR in their environment, relays their
    output and returns their exit codes.  `bench.py --gpus N` and `BSR(devices=[...]).fit` use it.
  * an external launcher (e.g. `python -m torch.distributed.run`) that sets the same RANK / WORLD_SIZE variables.

Either way the ranks meet through `Rendezvous`: a directory on the node's file system (one node is the whole scope
of the multi-GPU path) in which rank 0 publishes small blobs by atomic rename and the others poll.  It carries the
unique id that `bsr_comm_init` needs and nothing else on the product path; every later exchange (barrier, timing
reduction, the gather of accepted trees) is an RCCL all-gather through the C ABI (`bsr.dist.RcclGather`).
"""
import os
import socket
import subprocess
import sys
import tempfile
import time


def rank_env():
    """(rank, world, local_rank) of this process; (0, 1, 0) when no launcher variables are set."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class ExitCodes(list):
    """The ranks' exit codes, plus why spawn() ended them where it did: `timed_out` is None, "job" (the whole job
    outlived `timeout`) or "init" (the ranks did not all get through rendezvous and communicator set-up in
    `init_timeout`); `reason` says so in words."""
    timed_out = None
    reason = ""


def spawn(nprocs, argv, env_extra=None, timeout=None, relay_rank0_stdout=True, init_timeout=None):
    """Starts `nprocs` children running `sys.executable argv...`, one per GPU (LOCAL_RANK = rank).  Rank 0's stdout
    is returned (and echoed when relay_rank0_stdout); every child's stderr goes to this process's stderr.
    Returns (exit codes, rank-0 stdout).  The caller must not have initialised HIP: children are fresh processes.

    Every child is watched: when one exits non-zero (a bad device index, an import error, a crash) the others would
    sit in ncclCommInitRank or an all-gather forever, so they are killed after a short grace period and their codes
    returned (-9 for the killed ones).

    `timeout` (seconds) bounds the WHOLE job; None means no limit -- a long fit is not a hang -- unless
    BSR_SPAWN_TIMEOUT is set.  What can hang without anyone dying is the start: a rank waiting for a unique id that
    never comes, ncclCommInitRank on a stale one.  `init_timeout` (seconds; None: not watched) bounds that phase alone:
    every rank leaves a marker in the rendezvous directory once its first collective has returned (bsr.dist.connect),
    and ranks that have not all done so in time are ended.  Either way the returned codes say what happened
    (`ExitCodes.timed_out`, `.reason`) and a line goes to stderr: a timeout does not look like a crash."""
    import threading
    if timeout is None and os.environ.get("BSR_SPAWN_TIMEOUT"):
        timeout = float(os.environ["BSR_SPAWN_TIMEOUT"])
    rdv = tempfile.mkdtemp(prefix="bsr_rdv_")
    port = free_port()
    nonce = "%d-%d-%d" % (os.getpid(), port, int(time.time() * 1e6))
    procs = []
    for r in range(nprocs):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nprocs), "LOCAL_WORLD_SIZE": str(nprocs), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "BSR_RDV_DIR": rdv, "BSR_RDV_NONCE": nonce, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []

    def drain():   # rank 0's stdout must be read while it runs (a full pipe would block it)
        for piece in iter(lambda: procs[0].stdout.read(65536), b""):
            chunks.append(piece)
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    t_start = time.time()
    deadline = None if timeout is None else t_start + timeout
    init_deadline = None if init_timeout is None else t_start + init_timeout
    failed_at = None
    timed_out, reason = None, ""
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            now = time.time()
            if failed_at is None and any(c not in (None, 0) for c in codes):
                failed_at = now          # a rank died: the rest get a moment to fail on their own, then they are ended
            if init_deadline is not None:
                if all(os.path.exists(os.path.join(rdv, "up_%d" % r)) for r in range(nprocs)):
                    init_deadline = None     # everyone is through its first collective: the job runs as long as it runs
                elif now > init_deadline:
                    timed_out = "init"
                    reason = ("spawn: not every rank got through rendezvous and communicator set-up within %.0f s "
                              "(init_timeout); ending %d ranks" % (init_timeout, nprocs))
            if deadline is not None and now > deadline and timed_out is None:
                timed_out = "job"
                reason = "spawn: the job outlived its timeout of %.0f s (timeout= / BSR_SPAWN_TIMEOUT); ending %d ranks" % (
                    timeout, nprocs)
            if timed_out or (failed_at is not None and now - failed_at > 5.0):
                for p in procs:
                    if p.poll() is None:
                        p.kill()         # our own children, by pid
                break
            time.sleep(0.02)
        codes = ExitCodes(p.wait() for p in procs)
        codes.timed_out, codes.reason = timed_out, reason
        if reason:
            sys.stderr.write("bsr.launch." + reason + "\n")
            sys.stderr.flush()
        reader.join(timeout=5.0)
    finally:
        for name in os.listdir(rdv):
            try:
                os.unlink(os.path.join(rdv, name))
            except OSError:
                pass
        try:
            os.rmdir(rdv)
        except OSError:
            pass
    text = b"".join(chunks).decode(errors="replace")
    if relay_rank0_stdout and text:
        sys.stdout.write(text)
        sys.stdout.flush()
    return codes, text


class Rendezvous:
    """File-system meeting point of the ranks of one node."""

    def __init__(self, rank, world, directory=None, timeout=300.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        d = directory or os.environ.get("BSR_RDV_DIR")
        self.own_dir = False
        if not d:
            # external launcher: every rank is a child of the same agent process, and the port is unique per job
            d = os.path.join(tempfile.gettempdir(), "bsr_rdv_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
            self.own_dir = True
        self.dir = d
        self.t_start = time.time()
        # what tells this job's blobs from a crashed job's under the same directory: the launcher's nonce, or (external
        # launcher: port, agent pid and -- what changes when an elastic agent restarts its workers under the same pid and
        # port -- the run id and restart count) -- every blob starts with it
        self.nonce = (os.environ.get("BSR_RDV_NONCE") or "%s-%d-%s-%s" % (
            os.environ.get("MASTER_PORT", "0"), os.getppid(), os.environ.get("TORCHELASTIC_RUN_ID", ""),
            os.environ.get("TORCHELASTIC_RESTART_COUNT", ""))).encode()
        os.makedirs(d, exist_ok=True)
        self.seq = 0
        self.published = []

    def _path(self, name):
        return os.path.join(self.dir, name)

    def publish(self, name, blob):
        tmp = self._path(".%s.%d.tmp" % (name, self.rank))
        with open(tmp, "wb") as f:
            f.write(len(self.nonce).to_bytes(2, "little") + self.nonce + bytes(blob))
        os.replace(tmp, self._path(name))
        self.published.append(name)

    def fetch(self, name, nbytes=None):
        t_end = time.time() + self.timeout
        p = self._path(name)
        while True:
            try:
                with open(p, "rb") as f:
                    raw = f.read()
                n = int.from_bytes(raw[:2], "little")
                # a blob of another job (same directory reused after a crash) carries another nonce: not ours
                if len(raw) >= 2 + n and raw[2:2 + n] == self.nonce:
                    b = raw[2 + n:]
                    if nbytes is None or len(b) == nbytes:
                        return b
            except OSError:
                pass
            if time.time() > t_end:
                raise TimeoutError("rendezvous: %s did not appear in %s within %.0f s" % (name, self.dir, self.timeout))
            time.sleep(0.002)

    def broadcast(self, name, make_blob, nbytes=None):
        """Rank 0 creates the blob (make_blob()), everyone returns it."""
        if self.rank == 0:
            blob = bytes(make_blob())
            self.publish(name, blob)
            return blob
        return self.fetch(name, nbytes)

    def allgather(self, blob):
        """Host-side all-gather of equal-size blobs through the directory.  Not the product's gather (that is RCCL,
        bsr.dist.RcclGather): used by tests that run several ranks on one device, where RCCL refuses to start."""
        seq = self.seq
        self.seq += 1
        self.publish("ag%d_%d" % (seq, self.rank), bytes(blob))
        return [self.fetch("ag%d_%d" % (seq, r), len(blob)) for r in range(self.world)]

    def mark_up(self):
        """This rank's first collective has returned: the start-up phase spawn() may bound (init_timeout) is over."""
        if self.own_dir:      # (an external launcher's job: nobody reads the marker)
            return
        try:
            with open(self._path("up_%d" % self.rank), "wb") as f:
                f.write(self.nonce)
        except OSError:
            pass

    def close(self):
        """Rank 0 removes what it published (call after a collective that proves everyone has read it)."""
        for name in self.published:          # every name this rank published ("uid", "uid1", ..., "ag<seq>_<rank>")
            try:
                os.unlink(self._path(name))
            except OSError:
                pass
        self.published = []
        if self.rank != 0:
            return
        if self.own_dir:
            try:
                os.rmdir(self.dir)
            except OSError:
                pass
