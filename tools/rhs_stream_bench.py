#!/usr/bin/env python3
"""BASELINE configs[2] "HBM-roofline sweep": the one kernel of this path that streams its state
through HBM on every call is the bare RHS evaluation rmt_n2_rhs (y in, dy/dt out; the steppers
keep the state on chip instead).  E replicas of the 1024-node DME reactor with the state well
beyond the 256 MB Infinity Cache; HIP-event kernel times; algorithmic traffic 2*V*8 B per node."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B                          # noqa: E402
from rmt_app_amd import plan               # noqa: E402
from rmt_app_amd.n2 import N2Device        # noqa: E402
import torch                               # noqa: E402

DEFS = dict(a.split("=", 1) for a in sys.argv[1:])

print("| reactors E | nodes N | state in + out | block | kernel ms | node-RHS/s | GB/s (2*V*8 B per node) | of 8 TB/s | fp64 VALU T instr/s (444 per node) |")
print("|---|---|---|---|---|---|---|---|---|")
for E, N, block in ((256, 1024, 256), (2048, 1024, 256), (16384, 1024, 256), (16384, 1024, 512), (16384, 1024, 1024), (4096, 4096, 256)):
    inputs = B.sweep_member_inputs(0, min(E, 2048))
    mech = plan.Mechanism(inputs[0])
    rows = np.array([plan.member_constants(mi, mech, N)[1] for mi in inputs])
    rows = np.tile(rows, (E//len(rows), 1))
    nm, _ = plan.member_constants(inputs[0], mech, N)
    IV = np.tile(plan.initial_state(nm, mech, N), (E, 1))
    dev = N2Device(mech, rows, N, block=block, npt=1, specialize=False, defines=DEFS)
    y = dev.to_device(IV)
    out = dev.rhs(y)
    ms = []
    for _ in range(10):
        out = dev.rhs(y)
        ms.append(dev.last_kernel_ms())
    assert not dev.status().any()
    t = float(np.median(ms))
    by = 2*mech.V*8*E*N
    print("| %d | %d | %.2f GB | %d | %.3f | %.3e | %.0f | %.2f | %.1f |" % (
        E, N, by/1e9, block, t, E*N/(t*1e-3), by/(t*1e-3)/1e9, by/(t*1e-3)/8e12, 444*E*N/(t*1e-3)/1e12), flush=True)
    dev.close()
    del y, out
    torch.cuda.empty_cache()
