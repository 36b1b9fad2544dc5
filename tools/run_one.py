#!/usr/bin/env python3
"""One launch of one stepper on one shape (for rocprofv3):
   run_one.py <rk4|rk45|ros4> <mechanism> <N> <E> <t1-or-steps> [block] [npt] [mode] [NAME=VAL ...]
   COPT="<hipRTC options>", RTOL=<rtol>, SPECIALIZE=0|1 and LDS=0|1|2 are taken out of the NAME=VAL list (not kernel macros)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP  # noqa: E402
from rmt_app_amd import plan  # noqa: E402
from rmt_app_amd.n2 import N2Device  # noqa: E402

step, name, N, E, amount = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
pos = [a for a in sys.argv[6:] if "=" not in a]
defines = dict(a.split("=", 1) for a in sys.argv[6:] if "=" in a)
copt = defines.pop("COPT", "")
rtol = float(defines.pop("RTOL", 1e-6))
spec = defines.pop("SPECIALIZE", None)          # 0 / 1: member fields as run-time values / literals (default: n2's rule)
lds = defines.pop("LDS", None)                  # RK4 vectors kept in LDS (0, 1, 2; default: plan.Mechanism.lds_state)
block = int(pos[0]) if len(pos) > 0 and pos[0] != "-" else None
npt = int(pos[1]) if len(pos) > 1 and pos[1] != "-" else None
mode = pos[2] if len(pos) > 2 else "auto"
mi = INP.ALL_N2_INPUTS[name]()
mech = plan.Mechanism(mi)
nm, row = plan.member_constants(mi, mech, N)
dev = N2Device(mech, np.tile(row, (E, 1)), N, block=block, npt=npt, defines=defines, extra_opts=copt,
               lds_state=None if lds is None else int(lds),
               specialize=None if spec is None else bool(int(spec)),
               features=("ros4",) if step == "ros4" else ())
dev.set_mode(mode)
y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
if step == "rk4":
    dev.rk4(y, 2e-6, int(amount))
    work = E*N*int(amount)
else:
    getattr(dev, step)(y, 0.0, amount, rtol, 1e-3*rtol, 1e-6 if step == "rk45" else 1e-5, 10**8)
ms = dev.last_kernel_ms()
fl = dev.status()
if step != "rk4":
    st = dev.rk45_stats()
    work = N*int(st["accepted"].sum())
    print("accepted %d rejected %d per reactor: %.2f us per attempted step" % (
        int(st["accepted"][0]), int(st["rejected"][0]), 1e3*ms/max(1, int(st["accepted"][0]) + int(st["rejected"][0]))))
print("%s %s N=%d E=%d block=%dx%d mode=%s %s: %.3f ms, %.3e node-steps/s, flags %s" % (
    step, name, N, E, dev.block, dev.npt, mode, dict(defines, **({"COPT": copt} if copt else {})), ms, work/(ms/1e3), "ok" if not fl.any() else hex(int(fl.max()))))
dev.close()
