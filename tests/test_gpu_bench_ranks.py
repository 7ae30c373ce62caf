"""GPU tier: bench.py's N > 1 branch (SURVEY §4 tier 4), started the two ways a driver may start it — bare
(`python3 bench.py --gpus N`: bench.py spawns the N ranks itself as a child process) and under an outer launcher
(`python -m torch.distributed.run ... bench.py --gpus N`).  The build box has ONE GPU, so the ranks share it and the
exchange is torch.distributed over gloo (GPB_DIST_BACKEND=gloo; RCCL refuses two ranks on one device): everything of the
sharded bench except the wire runs — rendezvous, WalkerSharding, the per-rank row shares, the max-over-ranks timing, the
consistency check — and the ensemble after the same steps must equal the single-GPU run's bit for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

ARGS = ["--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline"]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(cmd, env, tag):
    # a rank that stalls for 100 s in one phase writes its Python stacks and the kernel-side state of its OS threads, then exits
    # (bench.py: Watchdog, GPB_BENCH_WATCHDOG); the launcher ends the others: a failed test with the evidence, not a hung suite
    env = dict(env, GPB_BENCH_WATCHDOG="100")
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    if r.returncode != 0:
        out = os.path.join(REPO, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "bench_ranks_%s.err" % tag), "wb") as f:
                f.write(r.stderr)
    assert r.returncode == 0, r.stderr.decode()[-6000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]              # ONE JSON line, from rank 0
    return json.loads(lines[0])


# R = 3: a ragged split (2048 proposal rows in shares of 683 / 683 / 682).  R = 4 and 8 are not rehearsed this way: round 4 saw two
# of four ranks that SHARED this one GPU stall inside a blocking pageable host->device copy of the gloo staging branch
# (profiles/r05_ranks_stall.txt: not the product path — RCCL, one GPU per rank, no host staging); 4 and 8 ranks are covered by
# test_sharded_c_loop_with_R_ranks_in_one_process (tests/test_gpu_sampler.py, the C loop's sharded form, R = 2/3/4/5/8) and by
# tests/test_dist_gloo.py (WalkerSharding at world 2/3/4/8 on the CPU).
@pytest.mark.parametrize("R,bare", [(2, True), (3, False)])
def test_bench_ranks_on_one_gpu_equal_the_single_gpu_run(R, bare):
    # torch.distributed.run gives every rank OMP_NUM_THREADS=1, the single-GPU process keeps all cores: no pin on either side.
    # The host linear algebra of a training (the scaler / PCA SVD, two small products; the synthetic observables) runs on ONE
    # BLAS thread whatever the process's threading (preprocess.single_thread_blas), so separate processes fit the same bits; on
    # top of that rank 0's fitted state is broadcast (WalkerSharding.replicate) and every sharded run starts with a digest check
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env, "one")
    env2 = dict(env, GPB_DIST_BACKEND="gloo")
    if bare:                # what a driver types: no launcher around it, WORLD_SIZE unset — bench.py starts the ranks itself
        two = _run([sys.executable, "bench.py", "--gpus", str(R)] + ARGS, env2, "R%d_bare" % R)
        assert "bench.py itself" in two["config"]["launched_by"]
    else:
        two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(R), "--master-addr",
                    "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(R)] + ARGS, env2, "R%d" % R)
        assert "outer launcher" in two["config"]["launched_by"]
    assert one["n_gpus"] == 1 and two["n_gpus"] == R
    assert one["config"]["ranks"] == 1 and two["config"]["ranks"] == R          # the world size the communicator reports
    import torch
    ndev = torch.cuda.device_count()                                            # (gloo rehearsal: the ranks share the box's GPU(s))
    assert two["config"]["devices_used"] == [r % ndev for r in range(R)]
    assert one["config"]["step_loop"] == "gpb_chain_emcee_run"
    assert two["config"]["step_loop"] == "host-driven"               # no in-stream RCCL collective under gloo
    assert two["config"]["parallelism"] == "walker-shard x%d" % R and "gloo" in two["config"]["allgather"]
    assert two["ranks_hold_identical_ensemble"] is True and one["ranks_hold_identical_ensemble"] is None
    assert two["extras"]["allgather_probe"]["us_per_allgather"] is None and "error" not in two["extras"]["allgather_probe"]   # gloo: no wire
    assert "allgather_probe" not in one.get("extras", {})
    # replicated draws + gathered log-probabilities: the sharded ensemble IS the unsharded one
    assert two["ensemble_checksum"] == one["ensemble_checksum"]
    assert two["acceptance_fraction"] == one["acceptance_fraction"]
    for out in (one, two):
        assert out["scaling"] == "strong" and out["steps"] == 4 and out["warmup"] == 2 and out["dtype"] == "f64"
        assert out["rows_inside_box_fraction"] >= 0.95               # burnt-in: every proposal row is evaluated ...
        assert abs(out["value_evaluated"] / out["value"] - out["rows_inside_box_fraction"]) < 1e-9
        assert out["tflops_algorithmic"] <= 78.6                     # ... and the headline stays under the fp64 peak
        uni = out["extras"]["uniform_start"]
        assert 0.3 < uni["rows_inside_box_fraction"] < 0.7 and uni["value_evaluated"] < uni["value"]
    # every rank evaluated its share of every batch
    assert two["roofline"]["launches"] == one["roofline"]["launches"]


def test_bare_bench_refuses_more_nccl_ranks_than_gpus():
    """`python3 bench.py --gpus 2` on a one-GPU box with the production backend: every rank refuses before the rendezvous (one
    process per GPU; no rank is folded onto another rank's device), the launcher fails, bench.py returns non-zero and prints no
    result line — never a line that timed a different job from the one asked for."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPB_BENCH_WATCHDOG="100")
    for k in ("GPB_DIST_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + ARGS, cwd=REPO, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert "2 ranks on this node but 1 visible GPU" in r.stderr.decode(), r.stderr.decode()[-3000:]


def test_bench_sharded_branch_over_a_one_rank_rccl_communicator():
    """Every nccl-only line of bench.py's sharded branch on the one GPU of a build box (GPB_BENCH_ONE_RANK_SHARDED=1): rank 0's
    state replicated through torch.distributed's nccl backend, the C ABI's own communicator formed (try_direct), the C loop with
    the in-stream ncclAllGather checked against the host-driven loop, the max-over-ranks timing, the ensemble consistency
    all-reduce and the all-gather probe.  Only the wire between ranks is missing — and the ensemble must equal the plain run's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "WORLD_SIZE", "RANK", "LOCAL_RANK", "GPB_DIST_BACKEND"):
        env.pop(k, None)
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env, "one_plain")
    sh = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS,
              dict(env, GPB_BENCH_ONE_RANK_SHARDED="1", MASTER_PORT=str(_free_port())), "one_rank_rccl")
    cfg = sh["config"]
    assert sh["n_gpus"] == 1 and cfg["ranks"] == 1 and "one-rank RCCL rehearsal" in cfg["parallelism"]
    assert cfg["allgather"].startswith("gpb_dist_allgather"), cfg["allgather"]       # the direct path passed its self-check
    assert cfg["step_loop"] == "gpb_chain_emcee_run"                                  # ... and the sharded C loop its own
    assert cfg["stream"] == "a non-blocking stream per rank" and cfg["devices_used"] == [0]
    assert sh["ranks_hold_identical_ensemble"] is True
    probe = sh["extras"]["allgather_probe"]
    assert "error" not in probe and 0.0 < probe["us_per_allgather"] < 500.0
    assert sh["ensemble_checksum"] == one["ensemble_checksum"] and sh["acceptance_fraction"] == one["acceptance_fraction"]
