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

# chained STIFF stepper: the whole 0.5 s transient, repeatedly, one reactor and a small ensemble; step history and
# result must equal the one-workgroup kernel's every time (bitwise identical across repeats of the same kernel)
from rmt_app_amd.settings import DEVICE_DEFAULTS as D
for N, E, reps in ((4096, 1, 10), (1024, 40, 10), (16384, 2, 3)):
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    IV = np.tile(plan.initial_state(nm, mech, N), (E, 1))
    dev = N2Device(mech, np.tile(row, (E, 1)), N, block=256, npt=1, features=("ros4",))
    ref = None
    ok = True
    for mode, n in (("mem", 1), ("chain", reps)):
        dev.set_mode(mode)
        for r in range(n):
            y = dev.to_device(IV)
            dev.ros4(y, 0.0, 0.5, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)
            fl = dev.status()
            st = dev.rk45_stats()
            out = y.cpu().numpy()
            if mode == "mem":
                ref, ref_acc = out, st["accepted"].copy()
            else:
                if r == 0:
                    first = out
                ok = ok and not fl.any() and np.array_equal(st["accepted"], ref_acc) and np.array_equal(out, first) \
                    and float(np.max(np.abs(out - ref)/np.maximum(np.abs(ref), 1e-300))) < 1e-6
    print("ros4 chain soak N=%d E=%d x%d: %s (accepted %d)" % (N, E, reps, "ok" if ok else "MISMATCH", int(ref_acc[0])), flush=True)
    dev.close()

# chained ADAPTIVE EXPLICIT stepper: long integrations (tens of thousands of steps = hundreds of thousands of link
# messages per chunk), one reactor over 4 / 16 chunks and ensembles with more reactors than teams; step history and
# result must equal the memory-resident kernel's, and repeats of the chained launch must be bitwise identical
from rmt_app_amd.n2 import rk45_geometry
for name, N, E, t1, reps in (("dme_nb", 4096, 1, 0.05, 3), ("dme_nb", 16384, 3, 0.02, 2), ("dme_nb", 2048, 300, 0.01, 2),
                             ("syn12", 1024, 200, 0.2, 2)):
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    rows, ivs = [], []
    for e in range(E):
        m2 = INP.ALL_N2_INPUTS[name]()
        m2["operating-conditions"]["temperature"] = mi["operating-conditions"]["temperature"] + (e % 11)
        nm, row = plan.member_constants(m2, mech, N)
        rows.append(row), ivs.append(plan.initial_state(nm, mech, N))
    block, npt, defs = rk45_geometry(mech.V, N)
    dev = N2Device(mech, np.array(rows), N, block=block, npt=npt, defines=defs)
    ok, ref, ref_acc, first, ms = True, None, None, None, 0.0
    for mode, n in (("mem", 1), ("chain", reps)):
        dev.set_mode(mode)
        for r in range(n):
            y = dev.to_device(np.array(ivs))
            dev.rk45(y, 0.0, t1, 1e-6, 1e-9, 1e-6, 10**8)
            fl, st, out = dev.status(), dev.rk45_stats(), y.cpu().numpy()
            if mode == "mem":
                ref, ref_acc = out, st["accepted"].copy()
            else:
                ms = dev.last_kernel_ms()
                if r == 0:
                    first = out
                ok = ok and not fl.any() and np.array_equal(st["accepted"], ref_acc) and np.array_equal(out, first) \
                    and float(np.max(np.abs(out - ref)/np.maximum(np.abs(ref), 1e-30))) < 1e-6
    print("rk45 chain soak %s N=%d E=%d x%d: %s (accepted %d..%d, %.0f ms per chained launch)" % (
        name, N, E, reps, "ok" if ok else "MISMATCH", int(ref_acc.min()), int(ref_acc.max()), ms), flush=True)
    dev.close()
