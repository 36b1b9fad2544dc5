#!/usr/bin/env python3
"""Shape sweep of the N2 steppers on one GPU -> markdown table (kept under profiles/).
Rows: BASELINE configs 2/3 (single reactor at 1024/4096/16384 nodes), ensemble sizes, the 12-species
mechanism, RK45.  Times are HIP-event kernel times of one launch (rmt_n2_last_kernel_ms)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP  # noqa: E402
from rmt_app_amd import plan  # noqa: E402
from rmt_app_amd.n2 import N2Device, rk45_geometry  # noqa: E402


def run(name, N, E, steps, mode="auto", dt=2e-6, **kw):
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    dev = N2Device(mech, np.tile(row, (E, 1)), N, **kw)
    dev.set_mode(mode)
    y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
    dev.rk4(y, dt, max(1, steps//10))
    dev.rk4(y, dt, steps)
    ms = dev.last_kernel_ms()
    fl = dev.status()
    W = dev.block*dev.npt
    kern = "mem" if mode == "mem" else ("reg" if N <= W else "chain[%d]" % (-(-N//W)))
    rate = E*N*steps/(ms/1e3)
    print("| %s | %d | %d | rk4_%s %dx%d lds%d | %d | %.3f | %.3e | %s |" % (
        name, N, E, kern, dev.block, dev.npt, dev.lds_state, steps, ms, rate, "ok" if not fl.any() else hex(int(fl.max()))),
        flush=True)
    dev.close()


def run_rk45(name, N, E, t1, rtol):
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    block, npt, defs = rk45_geometry(mech.V, N, E=E)                                      # what rmtExe(ivp="hip-rk45") picks
    dev = N2Device(mech, np.tile(row, (E, 1)), N, block=block, npt=npt, defines=defs)
    y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
    dev.rk45(y, 0.0, 1e-5, rtol, 1e-3*rtol, 1e-6, 10**8)         # warm-up launch
    dev.rk45(y, 1e-5, t1, rtol, 1e-3*rtol, -1e-6, 10**8)
    ms = dev.last_kernel_ms()
    st = dev.rk45_stats()
    fl = dev.status()
    acc, rej = int(st["accepted"].sum()), int(st["rejected"].sum())
    print("| %s | %d | %d | rk45_%s %dx%d | acc %d rej %d (t1=%g, rtol=%g) | %.3f | %.3e | %s |" % (
        name, N, E, ("reg" if N <= dev.block*dev.npt else "chain[%d]" % (-(-N//(dev.block*dev.npt)))) if defs else "mem",
        dev.block, dev.npt, acc, rej, t1, rtol, ms, N*acc/(ms/1e3), "ok" if not fl.any() else hex(int(fl.max()))),
        flush=True)
    dev.close()


print("| mechanism | nodes N | reactors E | kernel | steps | kernel ms | node-steps/s | flags |")
print("|---|---|---|---|---|---|---|---|")
run("dme_nb", 1024, 1, 2000)
run("dme_nb", 1024, 1, 2000, block=1024, npt=1)
run("dme_nb", 1024, 8, 2000)
run("dme_nb", 1024, 32, 2000)
run("dme_nb", 1024, 64, 2000)
run("dme_nb", 1024, 64, 2000, block=512, npt=2)
run("dme_nb", 1024, 100, 2000)
run("dme_nb", 1024, 128, 2000)
run("dme_nb", 1024, 128, 2000, block=512, npt=2)
run("dme_nb", 1024, 256, 2000)
run("dme_nb", 1024, 2048, 500)
run("dme_nb", 1024, 256, 500, mode="mem")
run("dme_nb", 4096, 1, 200)
run("dme_nb", 4096, 256, 100)
run("dme_nb", 16384, 1, 50)
run("dme_nb", 16384, 64, 50)
run("dme_nb", 20, 1, 20000)
run("dme_nb", 20, 2048, 2000)
run("syn12", 1024, 256, 500)
run("syn12", 512, 256, 500)
run("ch4", 1024, 256, 2000, dt=1e-4)
run_rk45("dme_nb", 1024, 256, 8e-3, 1e-6)
run_rk45("dme_nb", 4096, 64, 8e-3, 1e-6)
run_rk45("syn12", 512, 64, 0.1, 1e-6)
