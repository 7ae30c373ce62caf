"""GPU tier: short runs of the randomised soaks of tools/ (random shapes and inputs against the oracle; the long runs are in
profiles/r04_parity_soak.txt) — a few dozen cases each, a few seconds, no violation allowed:
  gpu_parity_soak       GP predict / LML / gradient, random N, d, number of GPs, kernel family, theta also at the edges of the search box
  gpu_posterior_soak    the whole log-posterior path from reference-format files
  gpu_multi_chain_soak  chains of 2..7 emulators (and the same bits with the launch batching off)
  gpu_sampler_soak      the device stretch move against emcee's algorithm, bit for bit
  gpu_shard_soak        the sharded resident loop on 2..8 loopback ranks (debug library), bit-identical to the unsharded run; its
                        self-test — one rank with another experiment vector — must be caught
"""
import contextlib
import importlib
import io
import json
import os
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(tool, cases, seed, *extra):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    try:
        mod = importlib.import_module(tool)
    finally:
        sys.path.pop(0)
    buf, old = io.StringIO(), sys.argv
    sys.argv = [tool, str(cases), str(seed), *extra]
    try:
        with contextlib.redirect_stdout(buf):
            mod.main()
    finally:
        sys.argv = old
    lines = [json.loads(ln) for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    return lines[-1], lines[:-1]


@pytest.mark.parametrize("tool,cases", [("gpu_parity_soak", 40), ("gpu_posterior_soak", 60), ("gpu_multi_chain_soak", 20),
                                        ("gpu_sampler_soak", 40)])
def test_randomised_soak_against_the_oracle(tool, cases):
    summary, before = _run(tool, cases, 11)
    assert summary["cases"] == cases and summary["violations"] == 0, [ln for ln in before if "done" not in ln]


def test_sharded_loop_soak_on_loopback_ranks(debug_lib):
    summary, before = _run("gpu_shard_soak", 30, 11)
    assert summary["cases"] == 30 and summary["violations"] == 0, [ln for ln in before if "done" not in ln]
    summary, _ = _run("gpu_shard_soak", 4, 12, "--selftest")
    assert summary["violations"] == 4                  # a rank that differs is caught every time
