#!/usr/bin/env python3
"""Short table of a bench.py line's extras: python tools/print_bench_extras.py <bench.json>"""
import json
import sys

d = json.load(open(sys.argv[1])); e = d["extras"]
print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 4), "k_predict frac", round(d["roofline"]["frac"], 3))
for k in ("fit_fixed_theta_cfg2", "fit_fixed_theta_cfg4", "fit_fixed_theta_cfg5"):
    print(k, "ms", round(e[k]["ms"], 4), "cholesky-only frac", round(e[k]["roofline"]["frac_cholesky_only"], 4))
for k in ("k_build_cfg4", "k_build_cfg5"):
    print(k, "us", round(e[k]["us"], 1), "frac", round(e[k]["roofline"]["frac"], 3))
print("lml_grad_cfg4 ms", round(e["lml_grad_cfg4"]["ms"], 3), "frac", round(e["lml_grad_cfg4"]["roofline"]["frac"], 3))
print("gp_predict_cfg2 ms", round(e["gp_predict_cfg2"]["ms"], 3), "host predict ms", round(e["emulator_predict_cfg2_host"]["ms"], 3))
n = e["nine_emulator_chain"]
print("nine emulators ms/step burnt-in", round(n["burnt_in"]["ms_per_step"], 3), "uniform", round(n["uniform_start"]["ms_per_step"], 3))
print("uniform start", round(e["uniform_start"]["value"]), round(e["uniform_start"]["ms_per_step"], 4))
if "cfg3_step" in e:
    c = e["cfg3_step"]
    print("cfg3_step ms/step", round(c["ms_per_step"], 4), "M walker-evals/s", round(c["walker_evals_per_s"] / 1e6, 3), "k_predict frac", round(c["roofline"]["frac"], 3))
    if "int8_predict" in c:
        print("  cfg3 int8:", c["int8_predict"])
if "cfg5_batch" in e:
    c = e["cfg5_batch"]
    print("cfg5_batch ms/batch", round(c["ms_per_batch"], 3), "rows/s", round(c["rows_per_s"]), "whole-batch frac", round(c["frac_of_peak_whole_batch"], 3),
          "k_predict frac", round(c["roofline"]["frac"], 3))
    if "int8_predict" in c:
        print("  cfg5 int8:", c["int8_predict"])
if "train_nine_emulators" in e:
    c = e["train_nine_emulators"]
    print("train nine emulators: one after the other", round(c["one_after_the_other_s"], 3), "s, together", round(c["train_emulators_s"], 3), "s, theta identical", c["theta_identical"])
