#!/usr/bin/env python3
"""BASELINE configs[4]: fp32 vs fp64 on the device - achieved tolerance and throughput.
For each mechanism: fp64 RK4 reference on the device vs the same run with real=float (the pressure
scan stays fp64), and adaptive RK45 in both precisions at several rtol."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device


def outlet_err(a, b, V, N):
    a = a.reshape(V, N); b = b.reshape(V, N)
    conc_a, conc_b = a[:V-1], b[:V-1]
    xa, xb = conc_a/conc_a.sum(0), conc_b/conc_b.sum(0)
    return float(np.max(np.abs(xa[:, -1] - xb[:, -1])/np.maximum(np.abs(xb[:, -1]), 1e-300))), float(abs(a[V-1, -1] - b[V-1, -1])/(1 + abs(b[V-1, -1])))


print("| mechanism | N | E | integrator | dtype | kernel ms | node-steps/s | max rel dMoFri (outlet) vs fp64 | dT'/(1+T') | flags |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, N, E, dt, steps in (("dme_nb", 1024, 256, 2e-6, 2000), ("syn12", 512, 256, 2e-6, 2000)):
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    ref = None
    for fp32 in (False, True):
        dev = N2Device(mech, np.tile(row, (E, 1)), N, fp32=fp32)
        y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
        dev.rk4(y, dt, steps)
        ms = dev.last_kernel_ms(); fl = dev.status()
        got = y.cpu().numpy().astype(np.float64)[0]
        if ref is None:
            ref = got
        e1, e2 = outlet_err(got, ref, mech.V, N)
        print("| %s | %d | %d | rk4 dt=%g x%d | %s | %.3f | %.3e | %.2e | %.2e | %s |" % (
            name, N, E, dt, steps, "fp32" if fp32 else "fp64", ms, E*N*steps/(ms/1e3), e1, e2,
            "ok" if not fl.any() else hex(int(fl.max()))), flush=True)
        dev.close()
    # adaptive
    t1 = 4e-3
    for rtol in (1e-4, 1e-6, 1e-8):
        for fp32 in (False, True):
            dev = N2Device(mech, np.tile(row, (8, 1)), N, fp32=fp32)
            y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (8, 1)))
            dev.rk45(y, 0.0, t1, rtol, 1e-3*rtol, 1e-6, 10**6)
            ms = dev.last_kernel_ms(); fl = dev.status(); st = dev.rk45_stats()
            got = y.cpu().numpy().astype(np.float64)[0]
            if not fp32 and rtol == 1e-4:
                pass
            if not fp32:
                ref45 = got
            # reference for the error: fp64 at rtol 1e-10 computed once
            if rtol == 1e-4 and not fp32:
                d2 = N2Device(mech, np.tile(row, (1, 1)), N)
                y2 = d2.to_device(plan.initial_state(nm, mech, N))
                d2.rk45(y2, 0.0, t1, 1e-11, 1e-14, 1e-6, 10**7)
                tight = y2.cpu().numpy()[0]; d2.close()
            e1, e2 = outlet_err(got, tight, mech.V, N)
            print("| %s | %d | 8 | rk45 rtol=%g t1=%g (acc %d rej %d) | %s | %.3f | %.3e | %.2e | %.2e | %s |" % (
                name, N, rtol, t1, int(st["accepted"][0]), int(st["rejected"][0]), "fp32" if fp32 else "fp64", ms,
                N*8*int(st["accepted"][0])/(ms/1e3), e1, e2, "ok" if not fl.any() else hex(int(fl.max()))), flush=True)
            dev.close()
