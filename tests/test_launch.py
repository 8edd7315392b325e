"""Torch-free multi-process plumbing of the multi-GPU path, exercised on CPU: the child-process
launcher, the rendezvous directory that carries the RCCL unique id, the record gather on top of an all-gather, and
bench.py's refusal to run when --gpus disagrees with the launcher."""
import os
import subprocess
import sys
import textwrap
import time

import pytest

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mcmc-symreg_amd")

WORKER = textwrap.dedent('''
    import hashlib, os, sys
    sys.path.insert(0, %(pkg)r)
    import numpy as np
    from bsr import dist as D
    from bsr.launch import Rendezvous, rank_env
    rank, world, local = rank_env()
    assert local == rank
    rdv = Rendezvous(rank, world)
    uid = rdv.broadcast("uid", lambda: os.urandom(128), nbytes=128)       # what bsr_comm_unique_id would hand out
    g = D.FileGather(rdv)
    got = g.allgather(np.frombuffer(uid, dtype=np.uint8))
    assert got.shape == (world, 128) and all(np.array_equal(got[r], got[0]) for r in range(world))
    assert D.allreduce_max(g, float(rank) + 0.5) == world - 0.5
    D.barrier(g)
    # the chain-record gather on top of it: 5 chains dealt round-robin, every rank ends with all 5 in chain order
    n_chains = 5
    recs = []
    for c in D.shard(n_chains, world, rank):
        tape = np.zeros(1, dtype=D.NODE_DTYPE)
        tape[0] = (10, -1, -1, c %% 3, 0.0, 0.0)
        recs.append(D.pack_record(c, None, np.arange(3.0) + c, 0.5 + c, [1.0 / (c + 1)] * c, 100 + c, c, 7, 9,
                                  tapes_in=[tape, tape]))
    # ... one of them as large as the reference's restart loop can make it: a depth-12 comb of 8 191 nodes and a
    # 2 000-entry RMSE history (codes/bsr_class.py:270-276 keeps whatever ROOTS.append(Roots) holds)
    big_tape = None
    if 3 in D.shard(n_chains, world, rank):
        from bsr.node import Node
        from bsr.tape import flatten
        def full(depth):
            if depth == 12:
                n = Node(depth); n.type = 0; n.feature = np.array([depth %% 3]); return n
            n = Node(depth); n.type, n.operator = 2, '+' if depth %% 2 else '*'
            n.left, n.right = full(depth + 1), full(depth + 1)
            n.left.parent = n.right.parent = n
            return n
        big_tape = flatten(full(0))
        assert len(big_tape) == 8191
        recs[[int(r[:4].view(np.int32)[0]) for r in recs].index(3)] = D.pack_record(
            3, None, np.arange(3.0) + 3, 3.5, list(np.linspace(9.0, 1.0, 2000)), 103, 3, 7, 9, tapes_in=[big_tape, big_tape[:1]])
    raw = D.gather_raw(g, recs)
    out = [D.unpack_record(r) for r in raw]
    assert [r["chain"] for r in out] == list(range(n_chains))
    assert out[3]["tape_len"] == [8191, 1] and out[3]["n_errs"] == 2000 and len(out[3]["errs"]) == 2000
    assert out[3]["errs"] == [float(v) for v in np.linspace(9.0, 1.0, 2000)] and not out[3]["errs_truncated"]
    if big_tape is not None:
        assert out[3]["tapes"][0].tobytes() == big_tape.tobytes()
    from bsr.node import Express
    assert Express(out[3]["roots"][0]).count("x[") == 4096
    out[3]["errs"] = [1.0 / 4] * 3       # (the checks below: the small record's values)
    for c, r in enumerate(out):
        assert r["n_props"] == 100 + c and r["n_accept"] == c and r["n_rank_rejects"] == 7 and r["n_discarded"] == 9
        assert r["errs"] == [1.0 / (c + 1)] * c and r["sigma"] == 0.5 + c
        assert np.array_equal(r["beta"].reshape(-1), np.arange(3.0) + c)
        assert int(r["tapes"][0]["feature"][0]) == c %% 3
    if rank == 0:
        print("DIGEST", hashlib.sha1(b"".join(r.tobytes() for r in raw)).hexdigest())
''')


def _run(world, tmp_path):
    script = tmp_path / ("worker%d.py" % world)
    script.write_text(WORKER % {"pkg": PKG})
    code = ("import sys; sys.path.insert(0, %r); from bsr.launch import spawn; "
            "codes, text = spawn(%d, [%r], relay_rank0_stdout=False, timeout=120); print(codes); print(text)"
            % (PKG, world, str(script)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == str([0] * world), r.stdout + r.stderr[-4000:]
    return [ln for ln in lines if ln.startswith("DIGEST")][0]


def test_spawned_ranks_meet_and_gather_world_size_invariant(tmp_path):
    three = _run(3, tmp_path)
    one = _run(1, tmp_path)
    assert three == one          # the gathered, ordered records do not depend on how the chains were sharded


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 1 but the launcher started 2" in r.stderr


def test_record_keeps_the_rmse_history_and_counters():
    sys.path.insert(0, PKG)
    from bsr import dist as D
    tape = np.zeros(3, dtype=D.NODE_DTYPE)
    tape[0] = (10, -1, -1, 1, 0.0, 0.0)
    tape[1] = (10, -1, -1, 0, 0.0, 0.0)
    tape[2] = (9, 0, 1, 0, 0.0, 0.0)
    errs = list(np.linspace(2.0, 1.0, 1500))                 # the whole history travels, whatever its length
    rec = D.pack_record(3, None, [0.1, 0.2], 0.7, errs, 5000, 1500, 11, 22, tapes_in=[tape])
    assert rec.size == D.record_bytes(rec) == D.HEADER_BYTES + 1500 * 8 + 3 * D.NODE_DTYPE.itemsize
    u = D.unpack_record(rec)
    assert u["n_errs"] == 1500 and len(u["errs"]) == 1500 and u["errs"][-1] == 1.0
    assert u["errs"] == [float(v) for v in errs]
    assert (u["n_rank_rejects"], u["n_discarded"], u["K"]) == (11, 22, 1)
    from bsr.node import Express
    assert Express(u["roots"][0]) == "(x[1])*(x[0])"


def test_a_dying_rank_ends_the_job_instead_of_hanging_it(tmp_path):
    """a rank that dies early (bad device index, import error) left rank 0 in ncclCommInitRank forever and
    fit() never returned.  spawn() watches every child: the survivors are ended a few seconds after the first
    non-zero exit, and the codes say who failed."""
    sys.path.insert(0, PKG)
    from bsr.launch import spawn
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n"
                      "    sys.exit(3)\n"
                      "print('rank 0 waiting', flush=True)\n"
                      "time.sleep(120)\n")
    t0 = time.time()
    codes, text = spawn(2, [str(script)], relay_rank0_stdout=False)
    assert time.time() - t0 < 30.0
    assert codes[1] == 3 and codes[0] != 0 and "rank 0 waiting" in text
    # ... and a job that outlives its timeout is ended as well
    slow = tmp_path / "slow.py"
    slow.write_text("import time\ntime.sleep(120)\n")
    t0 = time.time()
    codes, _ = spawn(2, [str(slow)], timeout=2.0, relay_rank0_stdout=False)
    assert time.time() - t0 < 30.0 and all(c != 0 for c in codes)
    assert codes.timed_out == "job" and "outlived its timeout of 2 s" in codes.reason   # told apart from a crash


def test_only_the_start_up_is_bounded_by_default(tmp_path, monkeypatch):
    """timeout=None used to mean "one hour", after which a long multi-GPU fit was killed with nothing to
    say why.  None is no limit again (BSR_SPAWN_TIMEOUT sets one); what is bounded is the phase that can hang without
    anyone dying -- rendezvous and communicator set-up -- through the markers the ranks leave once their first
    collective has returned."""
    sys.path.insert(0, PKG)
    from bsr.launch import spawn
    monkeypatch.delenv("BSR_SPAWN_TIMEOUT", raising=False)
    # ranks that come up (marker written) and then work for longer than init_timeout: left alone
    up = tmp_path / "up.py"
    up.write_text("import os, time\n"     # (what Rendezvous.mark_up does, without the package's import time)
                  "open(os.path.join(os.environ['BSR_RDV_DIR'], 'up_' + os.environ['RANK']), 'wb').close()\n"
                  "time.sleep(3.0)\n")
    t0 = time.time()
    codes, _ = spawn(2, [str(up)], init_timeout=1.5, relay_rank0_stdout=False)
    assert list(codes) == [0, 0] and codes.timed_out is None and time.time() - t0 >= 3.0
    # ranks stuck in their start-up (no marker): ended after init_timeout, and the codes say why
    stuck = tmp_path / "stuck.py"
    stuck.write_text("import time\ntime.sleep(120)\n")
    t0 = time.time()
    codes, _ = spawn(2, [str(stuck)], init_timeout=1.5, relay_rank0_stdout=False)
    assert time.time() - t0 < 30.0 and all(c != 0 for c in codes)
    assert codes.timed_out == "init" and "communicator set-up within 2 s" in codes.reason
    # BSR_SPAWN_TIMEOUT bounds the whole job where the caller gave no timeout
    monkeypatch.setenv("BSR_SPAWN_TIMEOUT", "1.5")
    codes, _ = spawn(1, [str(stuck)], relay_rank0_stdout=False)
    assert codes.timed_out == "job"


def test_a_sharded_rank_is_up_before_its_chains_run(tmp_path, monkeypatch):
    """bsr.sharded.run_rank joined the gather only AFTER its chains had run, so fit_sharded's
    init_timeout (900 s) covered the whole fit.  A rank shaped like run_rank is now: connect first (rendezvous,
    communicator, first collective, marker), work for longer than init_timeout, gather -- and is left alone; the old
    order (work, then connect) is what the timeout catches."""
    sys.path.insert(0, PKG)
    from bsr.launch import spawn
    monkeypatch.delenv("BSR_SPAWN_TIMEOUT", raising=False)
    body = ("import sys, time\nsys.path.insert(0, %r)\nimport numpy as np\nfrom bsr import dist as D\n"
            "from bsr.launch import rank_env\nrank, world, local = rank_env()\n" % PKG)
    good = tmp_path / "good.py"
    good.write_text(body + "g, rdv = D.connect(None, rank, world)\ntime.sleep(4.0)\n"
                    "recs = D.gather_raw(g, [D.pack_record(rank, None, [0.5], 0.7, [1.0], 3, 1, tapes_in=[])])\n"
                    "assert [D.unpack_record(r)['chain'] for r in recs] == list(range(world))\n")
    import inspect
    from bsr import sharded
    src = inspect.getsource(sharded.run_rank)
    assert src.index("D.connect(") < src.index("eng.run("), "run_rank must join the gather before its chains run"
    t0 = time.time()
    codes, _ = spawn(2, [str(good)], init_timeout=2.0, relay_rank0_stdout=False, env_extra={"BSR_SHARE_DEVICE": "1"})
    assert list(codes) == [0, 0] and codes.timed_out is None and time.time() - t0 >= 4.0
    late = tmp_path / "late.py"
    late.write_text(body + "time.sleep(30.0)\ng, rdv = D.connect(None, rank, world)\n")
    t0 = time.time()
    codes, _ = spawn(2, [str(late)], init_timeout=2.0, relay_rank0_stdout=False, env_extra={"BSR_SHARE_DEVICE": "1"})
    assert codes.timed_out == "init" and time.time() - t0 < 25.0


def test_an_elastic_restart_does_not_read_the_previous_attempts_blobs(tmp_path, monkeypatch):
    """under an external launcher the nonce was the port and the agent's pid -- the same after the agent
    restarts its workers, so a non-zero rank could fetch the previous attempt's unique id and hang in
    ncclCommInitRank.  The run id and the restart count are part of it now."""
    sys.path.insert(0, PKG)
    from bsr.launch import Rendezvous
    monkeypatch.delenv("BSR_RDV_NONCE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29511")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job7")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    a0 = Rendezvous(0, 2, directory=str(tmp_path), timeout=0.3)
    a0.publish("uid", b"x" * 128)
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    b1 = Rendezvous(1, 2, directory=str(tmp_path), timeout=0.3)
    with pytest.raises(TimeoutError):
        b1.fetch("uid", 128)
    b0 = Rendezvous(0, 2, directory=str(tmp_path), timeout=0.3)
    b0.publish("uid", b"y" * 128)
    assert b1.fetch("uid", 128) == b"y" * 128


def test_rendezvous_times_out_and_ignores_another_jobs_blobs(tmp_path):
    sys.path.insert(0, PKG)
    from bsr.launch import Rendezvous
    os.environ["BSR_RDV_NONCE"] = "job-a"
    try:
        a0 = Rendezvous(0, 2, directory=str(tmp_path), timeout=0.3)
        a0.publish("uid", b"x" * 128)
        a1 = Rendezvous(1, 2, directory=str(tmp_path), timeout=0.3)
        assert a1.fetch("uid", 128) == b"x" * 128
        os.environ["BSR_RDV_NONCE"] = "job-b"                 # a later job in the same directory: job a's uid is not its own
        b1 = Rendezvous(1, 2, directory=str(tmp_path), timeout=0.3)
        t0 = time.time()
        with pytest.raises(TimeoutError):
            b1.fetch("uid", 128)
        assert time.time() - t0 < 5.0
        a0.publish("uid1", b"y" * 128)
        a0.close()                                            # every name the rank published is removed
        assert not [n for n in os.listdir(tmp_path) if n.startswith("uid")]
    finally:
        os.environ.pop("BSR_RDV_NONCE", None)


def test_comm_entry_points_reject_bad_arguments_without_a_device():
    """The RCCL entry points fail with a code, never hang or crash, on arguments no communicator can be built from
    (no GPU needed: argument checks come first)."""
    sys.path.insert(0, PKG)
    import ctypes as C
    from bsr import _lib
    L = _lib.lib()
    buf = (C.c_ubyte * 128)()
    assert L.bsr_comm_init(None, 2, 0, buf) == -1             # BSR_E_ARG: no context
    assert L.bsr_comm_allgather(None, buf, buf, 8) == -1
    assert L.bsr_comm_unique_id(None) == -1


def test_records_of_any_size_survive_the_gather_and_the_file(tmp_path):
    """Nothing is cut: a long RMSE history and a long tape come back whole through split_records (what a rank's payload
    is parsed with), through a world-of-one gather, and through the .npz a sharded fit hands its parent."""
    sys.path.insert(0, PKG)
    from bsr import dist as D
    tape = np.zeros(1, dtype=D.NODE_DTYPE)
    tape[0] = (10, -1, -1, 0, 0.0, 0.0)
    long_tape = np.zeros(5001, dtype=D.NODE_DTYPE)
    long_tape[0] = (10, -1, -1, 1, 0.0, 0.0)
    for i in range(1, 5001, 2):                          # x1, then 2 500 times `x0 +`: a comb 2 501 deep
        long_tape[i] = (10, -1, -1, 0, 0.0, 0.0)
        long_tape[i + 1] = (8, i - 1, i, 0, 0.0, 0.0)
    recs = [D.pack_record(0, None, [0.1, 0.2], 0.7, list(range(5000)), 10, 5, tapes_in=[tape]),
            D.pack_record(1, None, [0.1, 0.2], 0.7, [1.0, 2.0], 10, 2, tapes_in=[long_tape]),
            D.pack_record(2, None, [0.3, 0.4], 0.9, [], 0, 0, tapes_in=[tape])]
    got = D.split_records(np.concatenate(recs))
    assert [g.tobytes() for g in got] == [r.tobytes() for r in recs]
    with pytest.raises(ValueError):
        D.split_records(np.concatenate(recs)[:-8])
    out = D.gather_chains(D.SoloGather(), recs[::-1])
    assert [o["chain"] for o in out] == [0, 1, 2]
    assert out[0]["errs"] == [float(v) for v in range(5000)] and not out[0]["errs_truncated"] and out[0]["truncated"] == 0
    assert out[1]["tape_len"] == [5001] and out[1]["tapes"][0].tobytes() == long_tape.tobytes()
    assert out[2]["errs"] == [] and np.isnan(out[2]["last_rmse"])
    path = str(tmp_path / "g.npz")
    D.save_records(path, recs)
    assert [r.tobytes() for r in D.load_records(path)] == [r.tobytes() for r in recs]
    D.save_records(path, [])
    assert D.load_records(path) == []


def test_a_rank_whose_communicator_fails_ends_the_job_with_its_error_text(tmp_path):
    """VERDICT r3 (de-risking the first N > 1 run): ncclCommInitRank failing on ONE rank -- a stale unique id, a GPU
    RCCL cannot open -- must surface as BSR_E_COMM with RCCL's text on that rank's stderr and end the whole job with
    non-zero codes, not leave the other ranks in the first collective.  The C ABI cannot fail on demand here (no GPU),
    so a stand-in context raises what bsr._lib.check makes of the C return code; everything above it is the product's
    (bsr.dist.connect, bsr.launch.spawn)."""
    sys.path.insert(0, PKG)
    from bsr.launch import spawn
    script = tmp_path / "rank.py"
    script.write_text(
        "import sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from bsr import dist as D, _lib\n"
        "from bsr.launch import rank_env\n"
        "rank, world, _ = rank_env()\n"
        "class Ctx:\n"
        "    def comm_unique_id(self):\n"
        "        return np.zeros(128, dtype=np.uint8)\n"
        "    def comm_init(self, n, r, uid):\n"
        "        if r == 1:\n"
        "            raise _lib.BsrError(-7, 'ncclCommInitRank: unhandled system error')   # what check() raises for BSR_E_COMM\n"
        "    def comm_allgather(self, send):\n"
        "        time.sleep(120)      # rank 0 sits in the first collective\n"
        "D.connect(Ctx(), rank, world)\n" % PKG)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r)\nfrom bsr.launch import spawn\n"
                        "codes, _ = spawn(2, [%r], relay_rank0_stdout=False, init_timeout=60)\nprint(list(codes))"
                        % (PKG, str(script))], capture_output=True, text=True, timeout=120)
    assert time.time() - t0 < 60.0
    assert "BSR_E_COMM (-7): ncclCommInitRank: unhandled system error" in r.stderr, r.stderr[-800:]
    codes = eval(r.stdout.strip().splitlines()[-1])
    assert codes[1] == 1 and codes[0] != 0


def test_local_device_follows_a_narrowed_device_list():
    sys.path.insert(0, PKG)
    from bsr.dist import local_device
    assert [local_device(r, 8) for r in range(8)] == list(range(8))      # the whole node visible: device = LOCAL_RANK
    assert [local_device(r, 1) for r in range(8)] == [0] * 8              # one GPU per rank visible: its index is 0
    assert [local_device(r, 4) for r in (4, 5, 6, 7)] == [0, 1, 2, 3]    # half a node per launcher group
    with pytest.raises(RuntimeError):
        local_device(0, 0)
