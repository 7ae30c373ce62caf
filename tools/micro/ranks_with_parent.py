"""Rehearsal of bench.py's N-rank gloo run on ONE GPU with a parent process that also holds a GPU context (what
tests/test_gpu_bench_ranks.py is: pytest + N ranks).  usage: ranks_with_parent.py <tag> <parent: none|torch|engine> <R> [KEY=VAL ...]
Ranks that stall dump their stacks after 60 s (GPB_BENCH_WATCHDOG) and exit; output in gpurun_out/<tag>.{out,err}."""
import os, socket, subprocess, sys, time

tag, parent, R = sys.argv[1], sys.argv[2], int(sys.argv[3])
extra = dict(kv.split("=", 1) for kv in sys.argv[4:])
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
keep = None
if parent == "torch":
    import torch
    keep = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
elif parent == "engine":
    import numpy as np, torch
    from gpbayestools_hic_amd import GPEngine, synth
    keep = GPEngine(0)
    X = synth.lhs(64, 3, seed=1)
    keep.set_data(X, np.sin(X.sum(1))[None, :], "RBF", 0.1)
    keep.set_theta(synth.fixed_theta(3, 1)); keep.factor(); keep.predict(X[:5]); torch.cuda.synchronize()
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPB_DIST_BACKEND="gloo", GPB_BENCH_WATCHDOG="60", **extra)
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(R), "--master-addr", "127.0.0.1",
       "--master-port", str(port), "bench.py", "--gpus", str(R), "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline"]
t0 = time.time()
out = os.path.join(ROOT, "gpurun_out")
with open(os.path.join(out, tag + ".out"), "wb") as fo, open(os.path.join(out, tag + ".err"), "wb") as fe:
    rc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=fo, stderr=fe, timeout=200).returncode
print("%s parent=%s R=%d %s rc=%d %.1fs" % (tag, parent, R, extra, rc, time.time() - t0), flush=True)
sys.exit(rc)
