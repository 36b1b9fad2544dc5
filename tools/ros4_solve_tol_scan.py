#!/usr/bin/env python3
"""Accuracy of the stiff stepper against the tight goldens as a function of the linear-solve tolerance
(RMT_ROS_SOLVE_TOL, in units of rtol): outlet error vs G4 (zNo = 20, reference LSODA rtol 1e-10) and whole-profile
error vs G8 (zNo = 1024, DOP853 rtol 1e-10), default rtol 1e-6 / atol 1e-9, with the wall time of the 1024-node run."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP  # noqa: E402
from oracle import n2_oracle as O  # noqa: E402
from rmt_app_amd import plan  # noqa: E402
from rmt_app_amd.n2 import N2Device, pack_interval  # noqa: E402

import torch  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
g4 = np.load(os.path.join(G, "g4_tight_dme_script_lsoda.npz"))
g8 = np.load(os.path.join(G, "g8_mesh1024_dme_nb_dop853.npz"))
print("| RMT_ROS_SOLVE_TOL | zNo=20 outlet rel err vs G4 (5 times) | steps | zNo=1024 max abs dMoFri vs G8 | max rel (all nodes) | steps | wall s (E=256) |")
print("|---|---|---|---|---|---|---|")
for spec in sys.argv[1:] or ["1e-2", "1e-1", "1"]:
    tol = spec.split(",")[0]
    defs = {"RMT_ROS_SOLVE_TOL": tol}
    defs.update(dict(kv.split("=") for kv in spec.split(",")[1:]))
    tol = spec
    # zNo = 20, reference test input
    mi = INP.dme_script_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 20)
    dev = N2Device(mech, row, 20, block=64, npt=1, defines=defs, features=("ros4",))
    y = dev.to_device(plan.initial_state(nm, mech, 20))
    worst, steps = 0.0, 0
    for k in range(5):
        dev.ros4(y, 0.1*k, 0.1*(k + 1), 1e-6, 1e-9, 1e-5 if k == 0 else -1e-5, 10**6)
        st = dev.rk45_stats()
        steps += int(st["accepted"][0] + st["rejected"][0])
        got = pack_interval(y.cpu().numpy()[0], nm, mech, 20, 0.1*(k + 1), "N2")["dataYs"][:, -1]
        ref = g4["dataYs_%d" % k][:, -1]
        worst = max(worst, float(np.max(np.abs(got - ref)/np.abs(ref))))
    assert not dev.status().any()
    dev.close()
    # zNo = 1024, notebook input, 256 identical members for the timing
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 1024)
    E = 256
    dev = N2Device(mech, np.tile(row, (E, 1)), 1024, block=256, npt=1, defines=defs, features=("ros4",))
    y = dev.to_device(np.tile(plan.initial_state(nm, mech, 1024), (E, 1)))
    pr = O.setup_n2(mi, 1024)
    wabs = wrel = 0.0
    steps2, wall = 0, 0.0
    done = int(g8["done"])
    for k in range(done):
        t0, t1 = (0.0 if k == 0 else float(g8["times"][k - 1])), float(g8["times"][k])
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        dev.ros4(y, t0, t1, 1e-6, 1e-9, 1e-5 if k == 0 else -1e-5, 10**6)
        torch.cuda.synchronize()
        wall += time.perf_counter() - w0
        st = dev.rk45_stats()
        steps2 += int(st["accepted"][0] + st["rejected"][0])
        got = pack_interval(y[0].cpu().numpy(), nm, mech, 1024, t1, "N2")["dataYs"]
        ref = O.pack_interval(g8["states"][k], pr, t1)["dataYs"]
        wabs = max(wabs, float(np.max(np.abs(got[:6] - ref[:6]))))
        wrel = max(wrel, float(np.max(np.abs(got - ref)/np.maximum(np.abs(ref), 1e-30))))
    assert not dev.status().any()
    dev.close()
    print("| %s | %.2e | %d | %.2e | %.2e | %d | %.4f |" % (tol, worst, steps, wabs, wrel, steps2, wall), flush=True)
