import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs as INP
from rmt_app_amd import plan
from rmt_app_amd.n2 import N2Device
np.set_printoptions(linewidth=200, precision=6)
g = np.load(os.path.join(ROOT, "tests/golden/g2_rhs.npz"))
Y, F = g["dme_nb_20_y"], g["dme_nb_20_f"]
mi = INP.dme_notebook_input()
mech = plan.Mechanism(mi)
nm, row = plan.member_constants(mi, mech, 20)
for block in (64, 256):
    dev = N2Device(mech, np.tile(row, (len(Y), 1)), 20, block=block, npt=1)
    out = dev.rhs(dev.to_device(Y)).cpu().numpy()
    print("block", block, "flags", dev.status())
    print("device:\n", out[0].reshape(7, 20)[:, :6])
    print("golden:\n", F[0].reshape(7, 20)[:, :6])
    dev.close()
