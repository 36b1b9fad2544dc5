#!/usr/bin/env python3
"""hip-ros4 tolerance scan on the reference's own test inputs (zNo=20, 0.5 s) vs golden G4."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import rmtExe
print("| input | rtol | atol | steps (acc) | rejected | wall s | max rel outlet err (per output time) |")
print("|---|---|---|---|---|---|---|")
for name in ("dme_script", "dme_nb"):
    g = np.load(os.path.join(ROOT, "tests", "golden", "g4_tight_%s_lsoda.npz" % name))
    for rtol in (1e-5, 3e-6, 1e-6, 3e-7, 1e-7, 1e-8):
        mi = INP.ALL_N2_INPUTS[name](ivp="hip-ros4")
        mi["solver-config"].update({"quiet": True, "rtol": rtol, "atol": 1e-3*rtol})
        t0 = time.perf_counter(); res = rmtExe(mi)["resModel"]; w = time.perf_counter() - t0
        errs = []
        for k in range(5):
            a, b = res["dataPack"][k]["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
            errs.append(float(np.max(np.abs(a - b)/np.abs(b))))
        st = res["device-stats"]
        print("| %s | %g | %g | %d | %d | %.3f | %s |" % (name, rtol, 1e-3*rtol, st["steps"], int(np.sum(st["rejected"])), w,
              " ".join("%.1e" % e for e in errs)), flush=True)
