#!/usr/bin/env python3
"""Model M2 with the chained steppers (tagged links): rk45 chain vs memory-resident, ros4 chain vs one workgroup."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device
from rmt_app_amd.settings import DEVICE_DEFAULTS as D

def members(N, E):
    mi = INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    rows, ivs = [], []
    for e in range(E):
        m2 = INP.m2_dme_input()
        m2["operating-conditions"]["temperature"] = mi["operating-conditions"]["temperature"] + (e % 7)
        nm, row = plan.member_constants_m2(m2, mech, N)
        rows.append(row); ivs.append(plan.initial_state_m2(nm, mech, N))
    return mech, np.array(rows), np.array(ivs)

def run(step, N, E, t1, mode, block, npt, defines=None, ref=None, rtol=1e-6):
    mech, rows, ivs = members(N, E)
    dev = N2Device(mech, rows, N, block=block, npt=npt, defines=defines, features=("ros4",) if step == "ros4" else ())
    dev.set_mode(mode)
    y = dev.to_device(ivs)
    if step == "rk45":
        dev.rk45(y, 0.0, 1e-4, rtol, 1e-3*rtol, 1e-6, 10**8)
        dev.rk45(y, 1e-4, t1, rtol, 1e-3*rtol, -1e-6, 10**8)
    else:
        dev.ros4(y, 0.0, t1, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**8)
    ms = dev.last_kernel_ms(); st = dev.rk45_stats(); fl = dev.status()
    out = y.cpu().numpy()
    d = ""
    if ref is not None:
        sc = np.max(np.abs(ref.reshape(E, mech.V, N)), axis=2, keepdims=True); sc[sc == 0] = 1
        d = "%.1e" % float(np.max(np.abs(out.reshape(E, mech.V, N) - ref.reshape(E, mech.V, N))/sc))
    print("| M2 %s | %d | %d | %s %dx%d %s | acc %d..%d rej %d | %.3f | %.3e | %s | %s |" % (
        step, N, E, mode, dev.block, dev.npt, defines or "", st["accepted"].min(), st["accepted"].max(), st["rejected"].max(),
        ms, N*float(st["accepted"].sum())/(ms/1e3), d, "ok" if not fl.any() else hex(int(fl.max()))), flush=True)
    dev.close()
    return out

print("| stepper | N | E | kernel | steps | ms | accepted node-steps/s | vs first | flags |")
print("|---|---|---|---|---|---|---|---|---|")
ref = run("rk45", 4096, 64, 0.02, "mem", 256, 1)
for blk, npt, lds in ((512, 2, 2), (256, 4, 2)):
    run("rk45", 4096, 64, 0.02, "chain", blk, npt, {"RMT_RK45_LDS": str(lds)}, ref)
RT = 1e-9          # far inside the stability limit: the two kernels must agree to rounding
ref = run("rk45", 4096, 8, 0.002, "mem", 256, 1, rtol=RT)
run("rk45", 4096, 8, 0.002, "chain", 512, 2, {"RMT_RK45_LDS": "2"}, ref, rtol=RT)
ref = run("rk45", 2500, 3, 0.02, "mem", 256, 1)
run("rk45", 2500, 3, 0.02, "chain", 256, 2, {"RMT_RK45_LDS": "4"}, ref)
ref = run("ros4", 4096, 1, 2.0, "mem", 256, 1)
run("ros4", 4096, 1, 2.0, "chain", 256, 1, None, ref)
ref = run("ros4", 2000, 12, 2.0, "mem", 256, 1)
run("ros4", 2000, 12, 2.0, "chain", 256, 1, None, ref)
