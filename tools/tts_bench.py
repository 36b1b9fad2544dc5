#!/usr/bin/env python3
"""Time-to-solution of the whole 0.5 s DME transient (the user-visible job): explicit RK4 at its
stability-limited dt vs the stiff Rosenbrock stepper, 256 x 1024-node sweep members on one GPU,
plus the outlet agreement between the two."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import bench as B
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device

DEFS = dict(a.split("=", 1) for a in sys.argv[4:])
SKIP_RK4 = bool(DEFS.pop("SKIP_RK4", ""))
ROS_BLOCK = int(DEFS.pop("ROS_BLOCK", 256))
RTOLS = [float(v) for v in DEFS.pop("RTOLS", "1e-5,1e-6,1e-7").split(",")]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T_END = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
inputs = B.sweep_member_inputs(0, E)
mech = plan.Mechanism(inputs[0])
pairs = [plan.member_constants(mi, mech, N) for mi in inputs]
rows = np.array([r for _, r in pairs]); IV = np.array([plan.initial_state(nm, mech, N) for nm, _ in pairs])
print("| integrator | E | N | t_end | steps (min/median/max per reactor) | kernel s | outlet max rel diff vs rk4 |")
print("|---|---|---|---|---|---|---|")
dev = N2Device(mech, rows, N)
y = dev.to_device(IV)
n = int(round((T_END if not SKIP_RK4 else 0.004)/2e-6))
import torch
t0 = time.perf_counter(); dev.rk4(y, 2e-6, n); torch.cuda.synchronize(); w = time.perf_counter() - t0
assert not dev.status().any()
ref = y.cpu().numpy().reshape(E, mech.V, N)[:, :, -1]
print("| hip-rk4 dt=2e-6 | %d | %d | %g | %d | %.3f | - |" % (E, N, T_END, n, w), flush=True)
dev.close()
for rtol in RTOLS:
    dev = N2Device(mech, rows, N, block=ROS_BLOCK, npt=1, defines=DEFS, features=("ros4",))
    y = dev.to_device(IV)
    t0 = time.perf_counter(); dev.ros4(y, 0.0, T_END, rtol, 1e-3*rtol, 1e-5, 10**7); torch.cuda.synchronize(); w = time.perf_counter() - t0
    fl = dev.status(); st = dev.rk45_stats()
    got = y.cpu().numpy().reshape(E, mech.V, N)[:, :, -1]
    tot = st["accepted"] + st["rejected"]
    print("| hip-ros4 rtol=%g | %d | %d | %g | %d/%d/%d (+%d rejected max) | %.3f | %.2e %s |" % (
        rtol, E, N, T_END, tot.min(), int(np.median(tot)), tot.max(), st["rejected"].max(), w,
        np.max(np.abs(got - ref)/np.maximum(np.abs(ref), 1e-300)), "" if not fl.any() else "FLAGS"), flush=True)
    dev.close()
