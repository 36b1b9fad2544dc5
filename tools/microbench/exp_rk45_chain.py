#!/usr/bin/env python3
"""rmt_n2_rk45_chain vs rmt_n2_rk45_mem: agreement and rate over geometries (DME, N = 4096)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device

def run(name, N, E, t1, mode, block, npt, defines=None, ref=None):
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    rows, ivs = [], []
    for e in range(E):
        m2 = INP.ALL_N2_INPUTS[name]()
        m2["operating-conditions"]["temperature"] = mi["operating-conditions"]["temperature"] + (e % 7)
        nm, row = plan.member_constants(m2, mech, N)
        rows.append(row); ivs.append(plan.initial_state(nm, mech, N))
    dev = N2Device(mech, np.array(rows), N, block=block, npt=npt, defines=defines)
    dev.set_mode(mode)
    y = dev.to_device(np.array(ivs))
    dev.rk45(y, 0.0, 1e-5, 1e-6, 1e-9, 1e-6, 10**8)
    dev.rk45(y, 1e-5, t1, 1e-6, 1e-9, -1e-6, 10**8)
    ms = dev.last_kernel_ms(); st = dev.rk45_stats(); fl = dev.status()
    out = y.cpu().numpy()
    d = ""
    if ref is not None:
        sc = np.max(np.abs(ref.reshape(E, mech.V, N)), axis=2, keepdims=True); sc[sc == 0] = 1
        d = "%.1e" % float(np.max(np.abs(out.reshape(E, mech.V, N) - ref.reshape(E, mech.V, N))/sc))
    print("| %s | %d | %d | %s %dx%d %s | acc %d..%d rej %d | %.3f | %.3e | %s | %s |" % (
        name, N, E, mode, dev.block, dev.npt, defines or "", st["accepted"].min(), st["accepted"].max(), st["rejected"].max(),
        ms, N*float(st["accepted"].sum())/(ms/1e3), d, "ok" if not fl.any() else hex(int(fl.max()))), flush=True)
    dev.close()
    return out

print("| mechanism | N | E | kernel | steps | ms | accepted node-steps/s | vs mem | flags |")
print("|---|---|---|---|---|---|---|---|---|")
t1 = 4e-3
ref = run("dme_nb", 4096, 64, t1, "mem", 256, 1)
for blk, npt, lds in ((512, 2, 2), (256, 2, 4), (512, 1, 4), (256, 2, 2), (256, 4, 2)):
    run("dme_nb", 4096, 64, t1, "chain", blk, npt, {"RMT_RK45_LDS": str(lds)}, ref)
ref = run("dme_nb", 4096, 3, t1, "mem", 256, 1)
run("dme_nb", 4096, 3, t1, "chain", 512, 2, {"RMT_RK45_LDS": "2"}, ref)
run("dme_nb", 4000, 3, t1, "chain", 256, 2, {"RMT_RK45_LDS": "4"})
