#!/usr/bin/env python3
"""rmtExe over a matrix of models / integrators / meshes beyond one workgroup: everything must run, flags clear, and the
adaptive results must agree with each other."""
import os, sys, time, copy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import rmtExe

def outlet(res):
    dp = res["resModel"]["dataPack"] if isinstance(res["resModel"], dict) and "dataPack" in res["resModel"] else None
    if dp is not None:
        return np.asarray(dp[-1]["dataYs"])[:, -1]
    rm = res["resModel"]
    return np.asarray(rm["dataList"][-1]["y"] if isinstance(rm, dict) and "dataList" in rm else rm[0]["dataYs"])[..., -1].ravel()

def run(tag, mi, **cfg):
    mi = copy.deepcopy(mi)
    mi["solver-config"].update({"quiet": True, **cfg})
    t = time.time()
    res = rmtExe(mi)
    dt = time.time() - t
    o = outlet(res)
    print("%-46s %6.2f s  outlet %s" % (tag, dt, np.array2string(o[-3:], precision=9)), flush=True)
    return o

base = INP.dme_notebook_input()
base["operating-conditions"]["period"] = 0.05
ref = run("N2 zNo=3000 hip-ros4", base, ivp="hip-ros4", zNo=3000)
for tag, cfg in (("N2 zNo=3000 hip-auto (default)", dict(ivp="default", zNo=3000)),
                 ("N2 zNo=3000 hip-rk45", dict(ivp="hip-rk45", zNo=3000, rtol=1e-8, atol=1e-11)),
                 ("N2 zNo=3000 BDF", dict(ivp="BDF", zNo=3000)),
                 ("N2 zNo=2048 hip-rk45 fp32", dict(ivp="hip-rk45", zNo=3000, dtype="fp32"))):
    o = run(tag, base, **cfg)
    print("    max rel diff vs hip-ros4: %.2e" % float(np.max(np.abs(o - ref)/np.abs(ref))))
ens = [copy.deepcopy(base) for _ in range(5)]
for k, m in enumerate(ens):
    m["operating-conditions"]["temperature"] = 523 + 4*k
b2 = copy.deepcopy(base)
b2["solver-config"].update({"quiet": True, "ivp": "hip-rk45", "zNo": 1500, "ensemble": ens})
t = time.time(); res = rmtExe(b2); print("N2 zNo=1500 ensemble of 5 hip-rk45              %6.2f s  members %d" % (time.time() - t, len(res["resModel"]["ensemble"])))
m2 = INP.m2_dme_input()
m2["operating-conditions"]["period"] = 0.5
o1 = run("M2 zNo=2500 hip-rk45", m2, ivp="hip-rk45", zNo=2500, rtol=1e-8, atol=1e-11)
o2 = run("M2 zNo=2500 hip-ros4", m2, ivp="hip-ros4", zNo=2500)
print("    M2 rk45 vs ros4: %.2e" % float(np.max(np.abs(o1 - o2)/np.abs(o2))))
