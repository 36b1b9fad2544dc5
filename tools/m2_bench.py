#!/usr/bin/env python3
"""Model M2 on one MI355X: RK4 throughput of an E x N ensemble
(inlet T/P sweep like bench.py's), the host-emulation CPU number beside it, the reference's own
M2 test job end to end (rmtExe, zNo/tNo defaults, period 10 s) and the accuracy against the
tight reference run (golden G9).  usage: m2_bench.py [E=256] [N=1024] [steps=2000]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP                               # noqa: E402
import bench as B                                  # noqa: E402
from rmt_app_amd import plan, rmtExe               # noqa: E402
from rmt_app_amd.ensemble import expand_members    # noqa: E402
from rmt_app_amd.n2 import N2Device                # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
DT = 2e-6
import torch                                       # noqa: E402

base = INP.m2_dme_input()
Ts = np.linspace(503.0, 543.0, 64)
Ps = np.linspace(3e6, 7e6, 32)
members = expand_members(base, {"temperature": Ts, "pressure": Ps})[:E]
mech = plan.Mechanism(base)
pairs = [plan.member_constants_m2(mi, mech, N) for mi in members]
rows = np.array([r for _, r in pairs])
IV = np.array([plan.initial_state_m2(nm, mech, N) for nm, _ in pairs])
out = {"E": E, "N": N, "steps": K, "dt": DT}

dev = N2Device(mech, rows, N)
y = dev.to_device(IV)
dev.rk4(y, DT, 200)
torch.cuda.synchronize()
t0 = time.perf_counter()
dev.rk4(y, DT, K)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = dev.last_kernel_ms()
assert not dev.status().any()
out["rk4"] = {"kernel": "rmt_n2_rk4_reg block=%d npt=%d (RMT_MODEL 2, %s Newton sweeps per RHS)" % (dev.block, dev.npt, dev.defines.get("RMT_M2_NEWTON")),
              "node_steps_per_s": E*N*K/(ms*1e-3), "kernel_ms": ms, "wall_s": wall,
              "hbm_fraction_128B": E*N*K/(ms*1e-3)*128/8e12}
dev.close()

if "--no-cpu" not in sys.argv:
    from oracle.hostemu import HostEmu
    from rmt_app_amd import hipbind
    emu = HostEmu(mech.source(hipbind.kernel_template(), False, 64, 1), tag="m2bench")
    cores = emu.set_threads(B.host_cores())
    Ec = min(E, 2*cores)
    yc = IV[:Ec].copy()
    emu.rk4(yc, rows[:Ec], N, DT, 2)
    n, steps, used = 20, 0, 0.0
    while used < 8.0:
        t0 = time.perf_counter()
        yc, _ = emu.rk4(yc, rows[:Ec], N, DT, n)
        used += time.perf_counter() - t0
        steps += n
    out["cpu_port"] = {"node_steps_per_s": Ec*N*steps/used, "cores": cores,
                       "sample": "%d members x %d nodes x %d RK4 steps" % (Ec, N, steps)}

# the reference's own M2 job (tests/test_rmt_DME.py): zNo=100, tNo=10, period 10 s, ivp LSODA
mi = INP.m2_dme_input(ivp="LSODA")
mi["solver-config"]["quiet"] = True
rmtExe(mi)                                          # JIT warm-up
t0 = time.perf_counter()
res = rmtExe(mi)["resModel"]
out["reference_job"] = {"config": "zNo=100 tNo=10 period=10 s ivp=LSODA -> hip-ros4 defaults",
                        "wall_s": time.perf_counter() - t0, "steps": int(res["device-stats"]["steps"]),
                        "outlet_T": float(res["dataPack"][-1]["dataYs"][-1, -1])}
g = np.load(os.path.join(ROOT, "tests", "golden", "g9_m2_tight_lsoda.npz"))
zNo, tNo = int(g["zNo"]), int(g["tNo"])
acc = {}
for ivp in ("LSODA", "hip-rk4"):
    mi = INP.m2_dme_input(ivp=ivp)
    mi["solver-config"].update({"zNo": zNo, "tNo": tNo, "quiet": True, "dt": 2e-6})
    t0 = time.perf_counter()
    r = rmtExe(mi)["resModel"]
    w = time.perf_counter() - t0
    worst = 0.0
    for k in range(tNo):
        ref = g["states"][k].reshape(7, zNo)
        ref_ys = np.concatenate((ref[:6]/np.sum(ref[:6], axis=0), ref[6:7]), axis=0)[:, -1]
        got = r["dataPack"][k]["dataYs"][:, -1]
        worst = max(worst, float(np.max(np.abs(got - ref_ys)/np.maximum(np.abs(ref_ys), 1e-300))))
    acc["hip-ros4" if ivp == "LSODA" else ivp] = {"max_rel_outlet_MoFri_T": worst, "wall_s": w,
                                                   "steps": int(r["device-stats"]["steps"])}
out["accuracy_vs_tight_reference"] = acc

# ensemble, stiff stepper, whole 10 s job
devr = N2Device(mech, rows, N, block=256, npt=1, features=("ros4",))
y = devr.to_device(IV)
t0 = time.perf_counter()
devr.ros4(y, 0.0, 10.0, 1e-6, 1e-9, 1e-5, 10**7)
torch.cuda.synchronize()
w = time.perf_counter() - t0
st = devr.rk45_stats()
tot = st["accepted"] + st["rejected"]
out["ros4_ensemble_10s"] = {"wall_s": w, "flags_ok": not devr.status().any(),
                            "steps_min_median_max": [int(tot.min()), int(np.median(tot)), int(tot.max())],
                            "rk4_equivalent_wall_s": (10.0/DT)*out["rk4"]["kernel_ms"]*1e-3/K}
devr.close()
print(json.dumps(out))
