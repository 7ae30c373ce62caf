"""Where does the parameterTrafoPCA path lose digits against the reference golden (tests/golden/g7_param_pca.npz)?
Prints LML / mean / cov errors of the HIP path with both forms of the cross-kernel distance."""
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, maxrel, relerr  # noqa: E402
from gpbayestools_hic_amd import Emulator, synth  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

g = golden("g7_param_pca.npz")
tmp = tempfile.mkdtemp()
tp, pf = os.path.join(tmp, "t.pkl"), os.path.join(tmp, "p.txt")
synth.write_training_pickle(tp, g["X"], g["Y"], 0.01)
synth.write_parameter_file(pf, g["lo"], g["hi"])
emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]), parameterTrafoPCA=True)
print("new_design_points maxrel", maxrel(emu.PCA_new_design_points, g["new_design_points"]))
emu.trainEmulator([True] * emu.nev, thetas=g["thetas"])
print("thetas", np.round(g["thetas"], 3))
print("lml relerr", relerr(emu.lml_, g["lml"]), emu.lml_, g["lml"])
eng = emu._engine_ready()
for dot in (1, 0):
    eng.tune("kcross_dot", dot)
    mean, cov = emu.predict(g["Xs"], return_cov=True, extra_std=0.0)
    print(f"kcross_dot={dot}: mean relerr {relerr(mean, g['mean']):.3e} maxrel {maxrel(mean, g['mean']):.3e}  cov maxrel {maxrel(cov, g['cov']):.3e}")
# the oracle on the SAME mapped inputs the device sees
Xg = emu._ppca.transform(g["Xs"])
oe = O.OracleEmulator(emu._X_train, g["Y"], emu.design_min, emu.design_max, int(g["npc"])).fit(g["thetas"])
m_ref, v_ref = oe.gp_predict(Xg)
gm, gv = eng.predict(Xg)
print("per-GP mean vs oracle maxrel", maxrel(gm, m_ref), " var relerr", relerr(gv, v_ref))
L = eng.get("L"); a = eng.get("alpha")
for p in range(len(oe.L)):
    print(p, "L maxrel", maxrel(L[p], oe.L[p]), "alpha maxrel", maxrel(a[p], oe.a[p]),
          "cond-ish", float(np.max(np.diag(oe.L[p])) / np.min(np.diag(oe.L[p]))) ** 2)
m2, c2 = oe.predict(Xg, True, 0.0)
print("oracle(device inputs) vs golden: mean", relerr(m2, g["mean"]), "cov", maxrel(c2, g["cov"]))
