import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device
mi = INP.m2_dme_input(); mech = plan.Mechanism(mi)
for N, E, steps in ((4096, 64, 200), (4096, 1, 200), (16384, 16, 100)):
    nm, row = plan.member_constants_m2(mi, mech, N)
    dev = N2Device(mech, np.tile(row, (E, 1)), N)
    y = dev.to_device(np.tile(plan.initial_state_m2(nm, mech, N), (E, 1)))
    dev.rk4(y, 2e-6, steps//10); dev.rk4(y, 2e-6, steps); ms = dev.last_kernel_ms()
    print(N, E, dev.block, dev.npt, "%.3e node-steps/s" % (E*N*steps/(ms/1e3)), dev.status().any(), flush=True)
    dev.close()
