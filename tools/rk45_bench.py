#!/usr/bin/env python3
"""Adaptive RK45 on one GPU: on-chip (rmt_n2_rk45_reg) vs memory-resident (rmt_n2_rk45_mem) stepper over
geometries.  Rows: accepted node-steps/s, agreement of the two kernels' end states and step counts.
usage: rk45_bench.py [quick]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP  # noqa: E402
from rmt_app_amd import plan  # noqa: E402
from rmt_app_amd.n2 import N2Device, rk45_block  # noqa: E402

REF = {}


def run(name, N, E, t1, rtol, mode, block, npt, defines=None, t0=0.0):
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    dev = N2Device(mech, np.tile(row, (E, 1)), N, block=block, npt=npt, defines=defines)
    dev.set_mode(mode)
    y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
    dev.rk45(y, 0.0, 1e-5, rtol, 1e-3*rtol, 1e-6, 10**8)      # warm-up launch (module upload, first steps)
    dev.rk45(y, 1e-5, t1, rtol, 1e-3*rtol, -1e-6, 10**8)
    ms = dev.last_kernel_ms()
    st = dev.rk45_stats()
    fl = dev.status()
    out = y.cpu().numpy()[0]
    key = (name, N, t1, rtol)
    if key not in REF:
        REF[key] = out
    V = mech.V
    scale = np.max(np.abs(REF[key].reshape(V, N)), axis=1, keepdims=True)
    scale[scale == 0] = 1
    diff = float(np.max(np.abs(out.reshape(V, N) - REF[key].reshape(V, N))/scale))
    acc, rej = int(st["accepted"][0]), int(st["rejected"][0])
    print("| %s | %d | %d | rk45_%s %dx%d %s | acc %d rej %d (t1=%g, rtol=%g) | %.3f | %.3e | %.1e | %s |" % (
        name, N, E, mode, dev.block, dev.npt, defines or "", acc, rej, t1, rtol, ms, E*N*acc/(ms/1e3), diff,
        "ok" if not fl.any() else hex(int(fl.max()))), flush=True)
    dev.close()


print("| mechanism | nodes N | reactors E | kernel | steps | kernel ms | accepted node-steps/s | vs first row | flags |")
print("|---|---|---|---|---|---|---|---|---|")
t1 = 8e-3
run("dme_nb", 1024, 256, t1, 1e-6, "mem", rk45_block(7, 1024), 1)
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2)
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 256, 4)
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 1024, 1)
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2, {"RMT_RK45_LDS": "1"})
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2, {"RMT_RK45_LDS": "0"})
if len(sys.argv) < 2:
    run("dme_nb", 1024, 2048, t1, 1e-6, "reg", 512, 2)
    run("dme_nb", 512, 512, t1, 1e-6, "reg", 512, 1)
    run("dme_nb", 512, 512, t1, 1e-6, "reg", 256, 2)
    run("dme_nb", 64, 2048, t1, 1e-6, "reg", 64, 1, {"RMT_RK45_LDS": "4"})
    run("dme_nb", 64, 2048, t1, 1e-6, "reg", 64, 1, {"RMT_RK45_LDS": "0"})
t1 = 0.1
run("syn12", 512, 64, t1, 1e-6, "mem", rk45_block(13, 512), 1)
run("syn12", 512, 64, t1, 1e-6, "reg", 512, 1)
run("syn12", 512, 64, t1, 1e-6, "reg", 256, 2)
run("syn12", 512, 64, t1, 1e-6, "reg", 256, 2, {"RMT_RK45_LDS": "3"})
run("syn12", 512, 256, t1, 1e-6, "reg", 256, 2)
