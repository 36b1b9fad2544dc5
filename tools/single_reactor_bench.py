#!/usr/bin/env python3
"""ONE reactor (BASELINE's target shape) under the stiff stepper: one workgroup vs chained over CUs.
usage: single_reactor_bench.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP  # noqa: E402
from rmt_app_amd import plan  # noqa: E402
from rmt_app_amd.n2 import N2Device  # noqa: E402
from rmt_app_amd.settings import DEVICE_DEFAULTS as D  # noqa: E402

import torch  # noqa: E402

print("| N | E | mode | steps (acc+rej) | wall s (0.5 s job) | us per step | outlet T [K] |")
print("|---|---|---|---|---|---|---|")
for N in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]:
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    IV = plan.initial_state(nm, mech, N)
    for E in (1, 8):
        dev = N2Device(mech, np.tile(row, (E, 1)), N, block=256, npt=1, features=("ros4",))
        for mode in ("mem", "chain"):
            dev.set_mode(mode)
            y = dev.to_device(np.tile(IV, (E, 1)))
            dev.ros4(y, 0.0, 1e-4, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)     # warm-up
            y = dev.to_device(np.tile(IV, (E, 1)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.ros4(y, 0.0, 0.5, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)
            torch.cuda.synchronize()
            w = time.perf_counter() - t0
            st = dev.rk45_stats()
            fl = dev.status()
            n = int(st["accepted"][0] + st["rejected"][0])
            Tout = y.cpu().numpy()[0].reshape(7, N)[6, -1]*nm["Tf"] + nm["Tf"]
            print("| %d | %d | %s | %d | %.4f | %.1f | %.4f %s |" % (N, E, mode, n, w, 1e6*w/n, Tout,
                                                                   "" if not fl.any() else "FLAGS"), flush=True)
        dev.close()
