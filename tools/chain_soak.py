#!/usr/bin/env python3
"""Soak of the chained-workgroup stepper: long launches at the BASELINE sizes (N2 and M2), status
words must stay clear (a stalled hand-over would end the launch with RMT_FLAG_STEP) and the result
must equal the same integration cut into short launches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device
for model, N, E, steps in (("N2", 16384, 64, 20000), ("N2", 4096, 1, 100000), ("M2", 4096, 64, 20000)):
    mi = INP.dme_notebook_input() if model == "N2" else INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    nm, row = (plan.member_constants if model == "N2" else plan.member_constants_m2)(mi, mech, N)
    IV = (plan.initial_state if model == "N2" else plan.initial_state_m2)(nm, mech, N)
    dev = N2Device(mech, np.tile(row, (E, 1)), N)
    y = dev.to_device(np.tile(IV, (E, 1)))
    dev.rk4(y, 2e-6, steps)
    ms = dev.last_kernel_ms()
    f1 = dev.status()
    y2 = dev.to_device(np.tile(IV, (E, 1)))
    for _ in range(20):
        dev.rk4(y2, 2e-6, steps//20)
    f2 = dev.status()
    a, b = y.cpu().numpy(), y2.cpu().numpy()
    print(model, N, E, steps, "%.0f ms" % ms, "flags", int(f1.max()), int(f2.max()), "one launch == 20 launches:", bool(np.array_equal(a, b)),
          "finite:", bool(np.isfinite(a).all()), flush=True)
    dev.close()
