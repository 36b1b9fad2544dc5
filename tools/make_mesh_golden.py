#!/usr/bin/env python3
"""Golden G8: SciPy on the host at a benchmark mesh size.

The reference's per-node Python RHS is infeasible at N >= 1024 (one LSODA Jacobian = 7 169 RHS calls
of ~0.65 s), so - as the survey prescribes - SciPy's `solve_ivp` drives the ORACLE's vectorised
restatement of modelEquationN2 (oracle/n2_oracle.py:make_rhs_vec, itself pinned <= 1e-12 against
the reference RHS at this N by fixture G2) with an explicit high-order pair at tight tolerances.
Nothing from the product (rmt_app_amd) is imported here.

usage: make_mesh_golden.py [input=dme_nb] [zNo=1024] [tNo=5] [rtol=1e-10] [atol=1e-13]
writes tests/golden/g8_mesh<zNo>_<input>_dop853.npz (re-written after every output interval, so a
partial run is usable: key `done` = number of intervals finished).
"""
import os
import sys
import time

import numpy as np
from scipy.integrate import solve_ivp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP                      # noqa: E402
from oracle import n2_oracle as O         # noqa: E402

kw = dict(a.split("=") for a in sys.argv[1:])
name = kw.get("input", "dme_nb")
zNo = int(kw.get("zNo", 1024))
tNo = int(kw.get("tNo", 5))
rtol = float(kw.get("rtol", 1e-10))
atol = float(kw.get("atol", 1e-13))
method = kw.get("method", "DOP853")
mi = {"dme_nb": INP.dme_notebook_input, "dme_script": INP.dme_script_input,
      "ch4": INP.ch4_input, "syn12": INP.syn12_input}[name]()
pr = O.setup_n2(mi, zNo=zNo)
f = O.make_rhs_vec(pr)
opT = mi["operating-conditions"]["period"]
span = np.linspace(0.0, opT, tNo + 1)
y = np.array(pr["IV"], dtype=float)
out = os.path.join(ROOT, "tests", "golden", "g8_mesh%d_%s_%s.npz" % (zNo, name, method.lower()))
states, nfev, wall = [], [], []
for i in range(tNo):
    t0 = time.time()
    sol = solve_ivp(f, (span[i], span[i + 1]), y, method=method, rtol=rtol, atol=atol)
    if not sol.success:
        raise RuntimeError(sol.message)
    y = sol.y[:, -1]
    states.append(y.copy())
    nfev.append(sol.nfev)
    wall.append(time.time() - t0)
    np.savez_compressed(out, input=name, zNo=zNo, tNo=tNo, rtol=rtol, atol=atol, method=method,
                        times=span[1:i + 2], states=np.array(states), nfev=np.array(nfev),
                        wall_s=np.array(wall), done=i + 1)
    print("interval %d: t=%.3f nfev=%d wall=%.0f s" % (i + 1, span[i + 1], sol.nfev, wall[-1]), flush=True)
