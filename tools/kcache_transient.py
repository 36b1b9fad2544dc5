#!/usr/bin/env python3
"""How often does the cached on-chip RK4 stepper fall back to its plain twin during the START-UP of the bench sweep
(256 reactors x 1024 nodes from the reference's initial state: bed at the feed composition and temperature), for
different refresh intervals of the cache's reference point (RMT_KC_REFRESH)?  usage: kcache_transient.py [K ...]
FIRST=<member index> in the environment picks another rank's block of the 2048-member sweep (hotter inlets)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B                          # noqa: E402
from rmt_app_amd import plan               # noqa: E402
from rmt_app_amd.n2 import N2Device        # noqa: E402
import torch                               # noqa: E402

E, N, STEPS, LAUNCHES = 256, 1024, 250, 40
FIRST = int(os.environ.get("FIRST", 0))
inputs = B.sweep_member_inputs(FIRST, E, total=2048)
mech = plan.Mechanism(inputs[0])
pairs = [plan.member_constants(mi, mech, N) for mi in inputs]
rows = np.array([r for _, r in pairs])
IV = np.array([plan.initial_state(nm, mech, N) for nm, _ in pairs])
print("members %d..%d of the sweep" % (FIRST, FIRST + E - 1))
print("| refresh K | launches x steps | fallbacks per launch (first 10) | total fallbacks | wall ms | max dT/dt seen K/s |")
print("|---|---|---|---|---|---|")
for K in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]:
    dev = N2Device(mech, rows, N, defines={"RMT_KC_REFRESH": str(K)} if K >= 0 else {"RMT_KCACHE": "0"})
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 1)
    y = dev.to_device(IV)
    base = dev.fallbacks()
    counts, rate = [], 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(LAUNCHES):
        before = y[:, 6*N:7*N].clone()
        dev.rk4(y, 2e-6, STEPS)
        n = dev.fallbacks()
        counts.append(n - base)
        base = n
        rate = max(rate, float((y[:, 6*N:7*N] - before).abs().max())*float(rows[:, 1].max())/(STEPS*2e-6))
    torch.cuda.synchronize()
    wall = 1e3*(time.perf_counter() - t0)
    assert not dev.status().any()
    print("| %s | %d x %d | %s | %d | %.1f | %.0f |" % (K if K >= 0 else "no cache", LAUNCHES, STEPS, counts[:10], sum(counts), wall, rate))
    dev.close()
