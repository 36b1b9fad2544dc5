#!/usr/bin/env python3
"""Model M2, chained RK4: flag protocol (default) vs tagged-word links (RMT_CHAIN_TAGGED 1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device
mi = INP.m2_dme_input(); mech = plan.Mechanism(mi)
for N, E, steps in ((4096, 64, 5000), (4096, 1, 20000), (16384, 16, 2000)):
    nm, row = plan.member_constants_m2(mi, mech, N)
    IV = np.tile(plan.initial_state_m2(nm, mech, N), (E, 1))
    ref = None
    for tag, mode in (("mem", None), ("flags", {}), ("tagged", {"RMT_CHAIN_TAGGED": "1"})):
        dev = N2Device(mech, np.tile(row, (E, 1)), N, defines=mode or None)
        dev.set_mode("mem" if tag == "mem" else "chain")
        y = dev.to_device(IV)
        dev.rk4(y, 2e-6, steps)
        ms = dev.last_kernel_ms(); fl = dev.status(); out = y.cpu().numpy()
        if ref is None: ref = out
        print("M2 rk4 N=%d E=%d %s %dx%d: %.2f ms %.3e node-steps/s flags %d maxdiff %.1e" % (
            N, E, tag, dev.block, dev.npt, ms, E*N*steps/(ms/1e3), int(fl.max()), float(np.max(np.abs(out - ref)/np.maximum(np.abs(ref), 1e-300)))), flush=True)
        dev.close()
