#!/usr/bin/env python3
"""Steady-state model N1 on one MI355X: a 64x32 inlet-T/P sweep of the
reference's TEST1.ipynb reactor, one reactor per lane, all 2048 profiles (101 points each) in ONE
launch, against the reference's own single profile (golden G6) and its wall time (BASELINE configs[0])."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP                      # noqa: E402
from rmt_app_amd import rmtExe            # noqa: E402

out = {}
mi = INP.n1_notebook_input()
mi["solver-config"]["quiet"] = True
rmtExe(mi)                                 # JIT
t0 = time.perf_counter()
res = rmtExe(mi)["resModel"][0]
out["single"] = {"wall_s": time.perf_counter() - t0, "steps": res["device-stats"],
                 "outlet_T": float(res["dataYs"][-1, -1]), "outlet_P_bar": float(res["dataYs"][-2, -1])}
g = np.load(os.path.join(ROOT, "tests", "golden", "g6_n1.npz"))
key = [k for k in g.files if "dataYs" in k][0]
ref = g[key]
out["single"]["max_rel_vs_reference_default_lsoda"] = float(np.max(np.abs(res["dataYs"] - ref)/np.maximum(np.abs(ref), 1e-30)))
for nT, nP in ((8, 8), (64, 32), (128, 128)):
    mi = INP.n1_notebook_input()
    mi["solver-config"].update({"quiet": True, "ensemble": {"temperature": list(np.linspace(503.0, 543.0, nT)),
                                                            "pressure": list(np.linspace(3e6, 7e6, nP))}})
    t0 = time.perf_counter()
    packs = rmtExe(mi)["resModel"]
    w = time.perf_counter() - t0
    acc = np.array([p["device-stats"]["accepted"] + p["device-stats"]["rejected"] for p in packs])
    out["sweep_%dx%d" % (nT, nP)] = {"profiles": len(packs), "wall_s_incl_host_packing": w,
                                      "profiles_per_s": len(packs)/w,
                                      "steps_min_median_max": [int(acc.min()), int(np.median(acc)), int(acc.max())],
                                      "outlet_T_range": [float(min(p["dataYs"][-1, -1] for p in packs)),
                                                         float(max(p["dataYs"][-1, -1] for p in packs))]}
print(json.dumps(out))
