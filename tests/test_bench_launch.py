"""CPU tier: bench.py's launch contract.  `python3 bench.py --gpus N` typed bare must itself start N ranks — as a
child process, before torch is imported or the GPU touched — relay rank 0's one JSON line and return the child's status; under a
launcher it must refuse a WORLD_SIZE that is not --gpus; a stalled rank must end as a failed line that says where it waited."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import REPO


def _py(code, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPB_BENCH_WATCHDOG"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=REPO, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_bare_gpus_n_spawns_the_ranks_as_a_child_before_torch_is_imported():
    r = _py("""
        import io, json, subprocess, sys
        seen = {}

        class FakeRanks:                                   # stands in for `python -m torch.distributed.run ... bench.py ...`
            def __init__(self, cmd, env=None, stdout=None):
                seen["cmd"], seen["env"] = cmd, env
                seen["torch_loaded_at_spawn"] = "torch" in sys.modules
                self.stdout = io.BytesIO(b"[Gloo] Rank 0 is connected to 1 peer ranks\\n"
                                         + json.dumps({"metric": "m", "n_gpus": 2}).encode() + b"\\n{not json\\n")
            def wait(self):
                return 0

        subprocess.Popen = FakeRanks
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-extras"]
        import bench
        try:
            bench.main()
        except SystemExit as e:
            seen["rc"] = e.code
        seen["torch_loaded_at_exit"] = "torch" in sys.modules
        seen["env"] = {k: seen["env"].get(k) for k in ("GPB_BENCH_SPAWNED", "HSA_ENABLE_IPC_MODE_LEGACY", "WORLD_SIZE")}
        print("SEEN " + json.dumps(seen), file=sys.stderr)
        """)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = r.stdout.decode().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "m", "n_gpus": 2}      # the one line; the rest went to stderr
    seen = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("SEEN ")][0][5:])
    assert seen["rc"] == 0 and not seen["torch_loaded_at_spawn"] and not seen["torch_loaded_at_exit"]
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-extras"]      # the same arguments
    assert seen["env"] == {"GPB_BENCH_SPAWNED": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "WORLD_SIZE": None}
    assert "[Gloo] Rank 0" in r.stderr.decode() and "{not json" in r.stderr.decode()


def test_bare_gpus_n_returns_the_childs_status_and_wants_exactly_one_line():
    code = """
        import io, subprocess, sys
        class FakeRanks:
            def __init__(self, cmd, env=None, stdout=None):
                self.stdout = io.BytesIO(%r)
            def wait(self):
                return %d
        subprocess.Popen = FakeRanks
        sys.argv = ["bench.py", "--gpus", "4"]
        import bench
        bench.main()
        """
    r = _py(code % (b"", 7))
    assert r.returncode == 7                                              # a failed child is a failed bench
    r = _py(code % (b"", 0))
    assert r.returncode == 3 and b"printed 0 result lines" in r.stderr    # status 0 without the line is not a result


def test_under_a_launcher_world_size_must_be_gpus():
    r = _py("import sys; sys.argv = ['bench.py', '--gpus', '2']; import bench; bench.main()",
            env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and b"refusing" in r.stderr
    r = _py("import sys; sys.argv = ['bench.py', '--gpus', '1']; import bench; bench.main()",
            env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2


def test_nccl_backend_refuses_more_ranks_than_gpus():
    # (no GPU here: zero visible devices, so any nccl world is too large — the message must carry both numbers)
    r = _py("""
        import os, sys
        os.environ.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", LOCAL_WORLD_SIZE="2")
        import torch
        torch.cuda.is_available = lambda: True
        torch.cuda.device_count = lambda: 1
        from gpbayestools_hic_amd.dist import init_from_env
        try:
            init_from_env(backend="nccl")
        except RuntimeError as e:
            print("REFUSED", e)
        """)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    msg = r.stdout.decode()
    assert "REFUSED" in msg and "2 ranks" in msg and "1 visible GPU" in msg


def test_watchdog_says_where_every_thread_waited_and_exits():
    r = _py("""
        import sys, time
        import bench
        dog = bench.Watchdog("rank 1 of 2")
        dog.arm(0.5, "stuck phase")
        time.sleep(30)
        print("not reached")
        """, timeout=60)
    err = r.stderr.decode()
    assert r.returncode == 124 and b"not reached" not in r.stdout
    assert "rank 1 of 2 made no progress in phase 'stuck phase'" in err
    assert "time.sleep" in err or "<module>" in err                       # the Python stacks
    assert "OS threads of pid" in err                                     # and the kernel-side view of every OS thread:
    rows = [ln.split() for ln in err[err.index("OS threads of pid"):].splitlines()[1:] if ln.strip()]
    assert len(rows) >= 2 and all(r[0].isdigit() and r[2] in "RSDTtZ" for r in rows), rows
    # the main thread sleeps in the kernel (state S, a wait channel, a system call number): what Python frames cannot show
    assert any(r[2] == "S" and r[3] != "0" for r in rows), rows


def test_watchdog_rearmed_and_disarmed_does_not_fire():
    r = _py("""
        import time
        import bench
        dog = bench.Watchdog("rank 0 of 2")
        dog.arm(0.4, "a"); time.sleep(0.2)
        dog.arm(0.4, "b"); time.sleep(0.2)
        dog.arm(0.0, "off"); time.sleep(0.6)
        print("done")
        """, timeout=60)
    assert r.returncode == 0 and r.stdout.decode().strip() == "done", r.stderr.decode()[-2000:]
