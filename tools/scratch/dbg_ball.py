import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import synth
from gpbayestools_hic_amd.sampler import StretchSampler
from gpbayestools_hic_amd.workload import build_chain

class Split:
    def __init__(self, world): self.world = world
    def logprob(self, fn, X, out):
        W = X.shape[0]; c = W // self.world
        for r in range(self.world):
            fn(X[r*c:(r+1)*c], out[r*c:(r+1)*c])
        return out

chain, emu, info = build_chain(4)
eng = emu._engine_ready()
nw = 4096
ball = float(min(1e-3, max(1e-13, 10.0 ** (-3.0 - 0.16 * 6))))
X0 = synth.walkers_ball(nw, info["xstar"], ball, lo=info["lo"], hi=info["hi"])
res = {}
for tag in ("c6", "c2+4", "c2+4prof", "host2+4", "split2+4"):
    s = StretchSampler(chain, nw, seed=12345, sharding=Split(2) if tag.startswith("split") else None)
    if not tag.startswith("c"): s._resident_engine = lambda: None
    if tag == "c6":
        s.run(X0, 6, status=10**9, store=False)
    else:
        s.run(X0, 2, status=10**9, store=False)
        if tag.endswith("prof"): eng.profile(True)
        torch.cuda.synchronize()
        s.run(None, 4, status=10**9, store=False)
        if tag.endswith("prof"): print(eng.profile_read()); eng.profile(False)
    res[tag] = (s.pos.cpu().numpy().copy(), s.lp.cpu().numpy().copy(), s.naccept.cpu().numpy().copy())
    print(tag, hashlib.sha256(res[tag][0].tobytes() + res[tag][1].tobytes()).hexdigest()[:16], "acc", res[tag][2].mean() / 6, flush=True)
for tag in res:
    print(tag, "pos==c6", np.array_equal(res[tag][0], res["c6"][0]), "lp==c6", np.array_equal(res[tag][1], res["c6"][1]),
          "nacc==", np.array_equal(res[tag][2], res["c6"][2]), np.abs(res[tag][0]-res["c6"][0]).max())
