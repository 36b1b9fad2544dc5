#!/usr/bin/env python3
"""Chained-workgroup stepper: strong scaling of ONE reactor over CUs and long reactors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device


def run(N, E, steps, mode, block=None, npt=None, lds=None):
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    dev = N2Device(mech, np.tile(row, (E, 1)), N, block=block, npt=npt, lds_state=lds)
    dev.set_mode(mode)
    y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
    dev.rk4(y, 2e-6, max(1, steps//10))
    dev.rk4(y, 2e-6, steps)
    ms = dev.last_kernel_ms()
    fl = dev.status()
    W = dev.block*dev.npt
    print("| %d | %d | %s %dx%d (chunks %d) | %d | %.3f | %.2f | %.3e | %s |" % (
        N, E, mode, dev.block, dev.npt, -(-N//W), steps, ms, 1e3*ms/steps, E*N*steps/(ms/1e3),
        "ok" if not fl.any() else hex(int(fl.max()))), flush=True)
    dev.close()


print("| nodes N | reactors E | stepper | steps | kernel ms | us/step | node-steps/s | flags |")
print("|---|---|---|---|---|---|---|---|")
run(1024, 1, 2000, "reg", 512, 2)
for b, n in ((64, 1), (64, 2), (128, 1), (128, 2), (256, 1), (256, 2)):
    run(1024, 1, 2000, "chain", b, n)
run(4096, 1, 1000, "mem")
for b, n in ((64, 1), (128, 1), (128, 2), (256, 2), (512, 2)):
    run(4096, 1, 1000, "chain", b, n)
run(16384, 1, 500, "chain", 64, 1)
run(16384, 1, 500, "chain", 128, 2)
run(16384, 1, 500, "chain", 512, 2)
run(4096, 64, 500, "chain")
run(4096, 256, 200, "chain")
run(16384, 16, 500, "chain")
run(16384, 64, 200, "chain")
