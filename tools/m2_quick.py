import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.ensemble import expand_members
from rmt_app_amd.n2 import N2Device
import torch
E, N, K = 256, 1024, 2000
base = INP.m2_dme_input()
members = expand_members(base, {"temperature": np.linspace(503.0, 543.0, 64), "pressure": np.linspace(3e6, 7e6, 32)})[:E]
mech = plan.Mechanism(base)
pairs = [plan.member_constants_m2(mi, mech, N) for mi in members]
rows = np.array([r for _, r in pairs]); IV = np.array([plan.initial_state_m2(nm, mech, N) for nm, _ in pairs])
for lds, defs in ((0, {}), (1, {}), (2, {}), (2, {"RMT_M2_NEWTON": 2})):
    dev = N2Device(mech, rows, N, lds_state=lds, defines=defs)
    y = dev.to_device(IV); dev.rk4(y, 2e-6, 200); dev.rk4(y, 2e-6, K); ms = dev.last_kernel_ms()
    print(lds, defs, "%.2f G node-steps/s" % (E*N*K/(ms*1e-3)/1e9), dev.status().any(), flush=True)
    dev.close()
