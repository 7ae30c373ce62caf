"""k_predict on 2048-row batches (BASELINE cfg 4 shapes) in bursts of 40 launches and sustained for several seconds: HIP-event time per launch of
every block of 200 consecutive predict calls.  usage: predict_sustained.py [seconds=6]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import GPEngine, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
c = synth.CONFIGS[4]
N, d, P = c["N"], c["d"], c["P"]
eng = GPEngine(0)
eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), c["kernel"], 0.1)
eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
Xs = torch.as_tensor(synth.walkers(2048, d), device="cuda")
for _ in range(5):
    eng.predict(Xs)
out = []
t0 = time.time()
while time.time() - t0 < secs:
    eng.profile(True)
    for _ in range(200):
        eng.predict(Xs)
    eng.sync()
    n_l, ms_l, _u = eng.profile_read()
    eng.profile(False)
    out.append(round(ms_l / n_l * 1e3, 1))
print(json.dumps({"k_predict_us_per_launch_by_block_of_200": out, "seconds": round(time.time() - t0, 1)}))
