#!/usr/bin/env python3
"""Block log-likelihood kernels at small batch sizes: one wave per walker (k_loglike_reg) vs one workgroup
per walker (k_loglike_wg).  Checks that both give the same bits and prints the whole log-posterior call time;
run under `rocprofv3 --kernel-trace` for the per-kernel durations."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    chain, emu, info = build_chain(cfg)
    eng = emu._engine_ready()
    for W in (128, 256, 512, 1024, 2048, 4096):
        X = torch.as_tensor(synth.walkers(W, info["d"]), device="cuda")
        X[::7] = 1.5                                   # some walkers outside the prior box
        row = {"W": W}
        outs = {}
        for name, sw in (("reg", 0), ("wg", 1 << 30)):
            eng.tune("mvn_wg_switch", sw)
            for _ in range(2):
                lp = chain.log_prob_device(X)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lp = chain.log_prob_device(X)
            e1.record(); torch.cuda.synchronize()
            row[name + "_call_ms"] = round(e0.elapsed_time(e1) / 10, 4)
            outs[name] = lp.clone()
        row["bit_identical"] = bool(torch.equal(outs["reg"], outs["wg"]))
        row["finite"] = int(torch.isfinite(outs["wg"]).sum())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
