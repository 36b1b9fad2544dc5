#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (PyREMOT).

Runs only in the build container, where /root/reference exists:

    PYTHONPATH=/root/reference MPLBACKEND=Agg python3 tools/make_golden.py <what> [...]

<what> in: setup rhs rk4 tight=<case> default=<case> multistep n1 helpers plot m2 m2run setting  (see SURVEY.md section 8(c), G1..G7).
The reference never travels to the GPU box; only the small .npz/.json files written here do.
Inputs come from tests/inputs.py (this repo's restatement of the reference's test inputs).
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

import inputs as INP  # noqa: E402

import PyREMOT  # noqa: E402  (the reference)
from PyREMOT import rmtExe  # noqa: E402
import PyREMOT.docs.pbHomoReactor as PBH  # noqa: E402
from PyREMOT.docs.pbHomoReactor import PackedBedHomoReactorClass as PB  # noqa: E402
from PyREMOT.solvers.solSetting import solverSetting  # noqa: E402
from PyREMOT.solvers import odeSolver as ODES  # noqa: E402
import scipy.integrate  # noqa: E402

REAL_SOLVE_IVP = scipy.integrate.solve_ivp


class _Captured(Exception):
    pass


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield


@contextlib.contextmanager
def mesh(zNo=None, tNo=None, model="N2"):
    old = dict(solverSetting[model])
    if zNo is not None:
        solverSetting[model]["zNo"] = zNo
    if tNo is not None:
        solverSetting[model]["tNo"] = tNo
    try:
        yield
    finally:
        solverSetting[model].clear()
        solverSetting[model].update(old)


def capture(mi, zNo):
    """Run rmtExe up to the first solve_ivp call; return (IV, paramsSet)."""
    box = {}

    def fake(fun, t_span, y0, method=None, t_eval=None, args=None, **kw):
        box["IV"] = np.array(y0, dtype=float)
        box["params"] = args[0]
        raise _Captured()

    PBH.solve_ivp = fake
    try:
        with mesh(zNo), quiet():
            try:
                rmtExe(mi)
            except _Captured:
                pass
    finally:
        PBH.solve_ivp = REAL_SOLVE_IVP
    return box["IV"], box["params"]


def rhs(params, y):
    return np.array(PB.modelEquationN2(0.0, np.array(y, dtype=float), params), dtype=float)


def run_with_tol(mi, zNo, method, rtol=None, atol=None):
    nfev = [0]

    def wrapped(fun, t_span, y0, method=None, t_eval=None, args=None, **kw):
        if rtol is not None:
            kw["rtol"] = rtol
        if atol is not None:
            kw["atol"] = atol
        sol = REAL_SOLVE_IVP(fun, t_span, y0, method=method, t_eval=t_eval, args=args, **kw)
        nfev[0] += sol.nfev
        return sol

    mi = dict(mi)
    mi["solver-config"] = dict(mi["solver-config"], ivp=method)
    PBH.solve_ivp = wrapped
    t0 = time.time()
    try:
        with mesh(zNo), quiet():
            res = rmtExe(mi)
    finally:
        PBH.solve_ivp = REAL_SOLVE_IVP
    return res, nfev[0], time.time() - t0


def pack_datapack(res):
    dp = res["resModel"]["dataPack"]
    out = {}
    for k, d in enumerate(dp):
        for key in ("dataYs", "dataYCons1", "dataYCons2", "dataYTemp1", "dataYTemp2", "dataXs"):
            out["%s_%d" % (key, k)] = np.array(d[key], dtype=float)
        out["dataTime_%d" % k] = np.array(float(d["dataTime"]))
    out["n"] = np.array(len(dp))
    return out


def tolist(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [tolist(x) for x in v]
    if isinstance(v, dict):
        return {k: tolist(x) for k, x in v.items() if not callable(x)}
    return v


def synthetic_states(IV, V, N, seed):
    """Deterministic test states of the (V,N) layout: smooth profile, noisy, and one with
    negative / tiny concentrations (exercises the EPS clamp, pbHomoReactor.py:3899,4093)."""
    rng = np.random.default_rng(seed)
    Y0 = IV.reshape(V, N)
    z = np.linspace(0, 1, N)
    states = []
    s1 = Y0.copy()
    for i in range(V):
        if i < V - 1 or V == Y0.shape[0]:
            s1[i] = Y0[i]*(1.0 + 0.15*np.sin(2.0*np.pi*(z + 0.1*i))) + 0.003*(i + 1)*z
    s1[V - 1] = 0.02*z + 0.01*np.sin(3*np.pi*z)
    states.append(s1)
    s2 = np.abs(Y0*(1 + 0.05*rng.standard_normal(Y0.shape))) + 1e-4*rng.random(Y0.shape)
    s2[V - 1] = 0.03*rng.random(N) - 0.005
    states.append(s2)
    s3 = s1.copy()
    idx = rng.integers(0, N, size=max(2, N//10))
    S = V - 1
    s3[min(2, S - 1), idx] = -1e-3*rng.random(len(idx))
    if S > 4:
        s3[4, idx[::2]] = 0.0
    s3[0, idx[1::2]] = -5e-2
    states.append(s3)
    return [s.flatten() for s in states]


# --------------------------------------------------------------------------- G1
def g_setup():
    out = {}
    for name, fn in INP.ALL_N2_INPUTS.items():
        mi = fn()
        IV, params = capture(mi, 20)
        rls, rsc, FunParam, DAP, ptype = params
        out[name] = {
            "input": {
                "concentration": tolist(np.array(mi["feed"]["concentration"], dtype=float)),
                "volumetric-flowrate": float(mi["feed"]["volumetric-flowrate"]),
            },
            "const": tolist(FunParam["const"]),
            "constBC1": tolist(FunParam["constBC1"]),
            "ExHe": tolist(FunParam["ExHe"]),
            "DimensionlessAnalysisParams": tolist(DAP),
            "reactionListSorted": tolist(rls),
            "reactionStochCoeff": tolist(rsc),
            "processType": ptype,
            "IV": IV.tolist(),
        }
    # iso-thermal variant of the notebook case
    mi = INP.dme_notebook_input(process_type="iso-thermal")
    IV, params = capture(mi, 20)
    out["dme_nb_iso"] = {"const": tolist(params[2]["const"]), "IV": IV.tolist(),
                         "DimensionlessAnalysisParams": tolist(params[3])}
    with open(os.path.join(GOLD, "g1_setup.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("G1 written")


# --------------------------------------------------------------------------- G2
def g_rhs():
    out = {}
    cases = [("dme_nb", 20), ("dme_nb", 100), ("dme_nb", 1024), ("dme_script", 20),
             ("dme_script", 100), ("ch4", 20), ("ch4", 100), ("syn12", 20), ("syn12", 100)]
    for name, zNo in cases:
        mi = INP.ALL_N2_INPUTS[name]()
        IV, params = capture(mi, zNo)
        V = params[2]["const"]["varNo"]
        states = [IV] + synthetic_states(IV, V, zNo, seed=zNo + len(name))
        if zNo == 1024:
            states = states[:3]
        Y = np.array(states)
        t0 = time.time()
        F = np.array([rhs(params, y) for y in Y])
        print("G2 %s zNo=%d: %d RHS calls in %.1fs" % (name, zNo, len(Y), time.time() - t0))
        out["%s_%d_y" % (name, zNo)] = Y
        out["%s_%d_f" % (name, zNo)] = F
    # iso-thermal variant
    mi = INP.dme_notebook_input(process_type="iso-thermal")
    IV, params = capture(mi, 20)
    Y = np.array([IV] + [s.reshape(7, 20)[:6].flatten() for s in synthetic_states(
        np.concatenate([IV, np.zeros(20)]), 7, 20, seed=5)])
    out["dme_nb_iso_20_y"] = Y
    out["dme_nb_iso_20_f"] = np.array([rhs(params, y) for y in Y])
    # mid-transient states from the tight run (if present)
    p = os.path.join(GOLD, "g4_tight_dme_script_lsoda.npz")
    if os.path.exists(p):
        g4 = np.load(p)
        mi = INP.dme_script_input()
        IV, params = capture(mi, 20)
        Y = np.array([np.concatenate([g4["dataYCons1_%d" % k].flatten(),
                                      g4["dataYTemp1_%d" % k].flatten()]) for k in range(5)])
        out["dme_script_20_transient_y"] = Y
        out["dme_script_20_transient_f"] = np.array([rhs(params, y) for y in Y])
    np.savez_compressed(os.path.join(GOLD, "g2_rhs.npz"), **out)
    print("G2 written")


# --------------------------------------------------------------------------- G3
def g_rk4():
    out = {}
    for name, zNo, h, n, stride in [("dme_nb", 20, 1e-5, 200, 1), ("dme_script", 20, 1e-5, 200, 1),
                                    ("dme_nb", 100, 1e-5, 100, 10), ("ch4", 20, 1e-3, 200, 1),
                                    ("syn12", 20, 1e-5, 100, 1)]:
        mi = INP.ALL_N2_INPUTS[name]()
        IV, params = capture(mi, zNo)
        t0 = time.time()
        traj = ODES.RK4(0.0, n*h, n, IV, PB.modelEquationN2, params)
        print("G3 %s zNo=%d n=%d: %.1fs" % (name, zNo, n, time.time() - t0))
        key = "%s_%d" % (name, zNo)
        out[key + "_traj"] = traj[:, ::stride]
        out[key + "_h"] = np.array(h)
        out[key + "_n"] = np.array(n)
        out[key + "_stride"] = np.array(stride)
    np.savez_compressed(os.path.join(GOLD, "g3_rk4.npz"), **out)
    print("G3 written")


# --------------------------------------------------------------------------- G4/G5
def g_multistep():
    """Reference PreCorr3 / AdBash3 (PyREMOT/solvers/odeSolver.py:43-102) trajectories."""
    out = {}
    for name, zNo, h, n in [("dme_nb", 20, 2e-6, 120), ("ch4", 20, 1e-3, 120)]:
        mi = INP.ALL_N2_INPUTS[name]()
        IV, params = capture(mi, zNo)
        for meth in ("PreCorr3", "AdBash3"):
            traj = getattr(ODES, meth)(0.0, n*h, n, IV, PB.modelEquationN2, params)
            out["%s_%d_%s" % (name, zNo, meth)] = traj[:, [3, n//2, n]]
        out["%s_%d_h" % (name, zNo)] = np.array(h)
        out["%s_%d_n" % (name, zNo)] = np.array(n)
    np.savez_compressed(os.path.join(GOLD, "g3b_multistep.npz"), **out)
    print("G3b written")


def g_tight(which):
    name, method = which.split(":")
    tol = {"lsoda": ("LSODA", 1e-10, 1e-12), "bdf": ("BDF", 1e-9, 1e-12)}[method]
    mi = INP.ALL_N2_INPUTS[name]()
    res, nfev, wall = run_with_tol(mi, 20, tol[0], tol[1], tol[2])
    out = pack_datapack(res)
    out["nfev"] = np.array(nfev)
    out["wall"] = np.array(wall)
    np.savez_compressed(os.path.join(GOLD, "g4_tight_%s_%s.npz" % (name, method)), **out)
    print("G4 %s %s: nfev=%d wall=%.1fs" % (name, method, nfev, wall))


def g_default(name):
    mi = INP.ALL_N2_INPUTS[name]()
    res, nfev, wall = run_with_tol(mi, 20, "LSODA")
    out = pack_datapack(res)
    out["nfev"] = np.array(nfev)
    out["wall"] = np.array(wall)
    out["computation_time"] = np.array(res["resModel"]["computation-time"])
    np.savez_compressed(os.path.join(GOLD, "g5_default_%s.npz" % name), **out)
    print("G5 %s: nfev=%d wall=%.1fs" % (name, nfev, wall))


# --------------------------------------------------------------------------- G6
def g_n1():
    mi = INP.n1_notebook_input()
    t0 = time.time()
    with quiet():
        res = rmtExe(mi)
    d = res["resModel"][0]
    out = {k: np.array(d[k], dtype=float) for k in
           ("dataYs", "dataYCons1", "dataYCons2", "dataYTemp1", "dataYTemp2", "dataXs")}
    out["wall"] = np.array(time.time() - t0)
    # a few direct RHS probes of modelEquationN1
    box = {}

    def fake(fun, t_span, y0, method=None, t_eval=None, args=None, **kw):
        box["IV"] = np.array(y0, float)
        box["params"] = args[0]
        raise _Captured()
    PBH.solve_ivp = fake
    try:
        with quiet():
            try:
                rmtExe(mi)
            except _Captured:
                pass
    finally:
        PBH.solve_ivp = REAL_SOLVE_IVP
    IV = box["IV"]
    ys = [IV, d["dataYCons1"][:, 50].tolist() + [d["dataYs"][6, 50]/5e6, d["dataYTemp1"][50]],
          d["dataYCons1"][:, 100].tolist() + [d["dataYs"][6, 100]/5e6, d["dataYTemp1"][100]]]
    ys = np.array([np.array(y, float) for y in ys])
    with quiet():
        fs = np.array([PB.modelEquationN1(0.37, y, box["params"]) for y in ys])
    out["rhs_y"] = ys
    out["rhs_f"] = fs
    np.savez_compressed(os.path.join(GOLD, "g6_n1.npz"), **out)
    print("G6 written")


# --------------------------------------------------------------------------- G7
def g_helpers():
    from PyREMOT.docs.rmtThermo import (calHeatCapacityAtConstantPressure,
                                        calMeanHeatCapacityAtConstantPressure,
                                        calMixtureHeatCapacityAtConstantPressure,
                                        calStandardEnthalpyOfReaction,
                                        calEnthalpyChangeOfReaction)
    from PyREMOT.docs.gasTransPor import calGasViscosity, calMixturePropertyM1
    from PyREMOT.docs.rmtUtility import rmtUtilityClass as U
    from PyREMOT.data import componentSymbolList, componentDataStore
    comps = list(componentSymbolList)
    out = {"components": comps, "MW": [c["MW"] for c in componentDataStore["payload"]],
           "dHf25": [c["dHf25"]["val"] for c in componentDataStore["payload"]], "probes": []}
    mf = np.arange(1, len(comps) + 1, dtype=float)
    mf = mf/mf.sum()
    MW = np.array(out["MW"])
    for T in (298.15, 400.0, 523.0, 700.0, 973.0):
        cp = calHeatCapacityAtConstantPressure(comps, T)
        cpm = calMeanHeatCapacityAtConstantPressure(comps, T)
        vis = calGasViscosity(comps, T)
        out["probes"].append({
            "T": T, "Cp": cp.tolist(), "CpMean": cpm.tolist(),
            "CpMix": float(calMixtureHeatCapacityAtConstantPressure(mf, cpm)),
            "GaVii": vis.tolist(),
            "GaMiVi": float(calMixturePropertyM1(len(comps), vis, mf, MW)),
        })
    out["molefrac"] = mf.tolist()
    rx = dict(INP.SYN12_REACTIONS)
    rls = U.buildReactionCoefficient(rx)
    out["reactions"] = rx
    out["reactionListSorted"] = tolist(rls)
    out["reactionStochCoeff"] = tolist(U.buildReactionCoeffVector(rls))
    out["StHeRe25"] = [float(calStandardEnthalpyOfReaction(r)) for r in rx.values()]
    out["EnChList_600"] = [float(v) for v in calEnthalpyChangeOfReaction(rls, 600.0)]
    with open(os.path.join(GOLD, "g7_helpers.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("G7 written")


# --------------------------------------------------------------------------- G9 (model M2)
def m2_capture(mi, zNo):
    """rmtExe(model M2) up to the first solve_ivp call (pbReactor.py:719): IV and the args tuple."""
    import PyREMOT.docs.pbReactor as PBR
    box = {}

    def fake(fun, t_span, y0, method=None, t_eval=None, args=None, **kw):
        box["IV"] = np.array(y0, dtype=float)
        box["args"] = args
        box["fun"] = fun
        raise _Captured()

    PBR.solve_ivp = fake
    try:
        with mesh(zNo, model="S2"), quiet():
            try:
                rmtExe(mi)
            except _Captured:
                pass
    finally:
        PBR.solve_ivp = REAL_SOLVE_IVP
    return box["IV"], box["args"], box["fun"]


def m2_states(IV, V, N, seed):
    """dimensional test states: rows C_i [kmol/m^3], last row T [K]"""
    rng = np.random.default_rng(seed)
    Y0 = IV.reshape(V, N)
    z = np.linspace(0, 1, N)
    s1 = Y0.copy()
    for i in range(V - 1):
        s1[i] = Y0[i]*(1.0 + 0.15*np.sin(2.0*np.pi*(z + 0.1*i))) + 0.003*(i + 1)*z*Y0[0]
    s1[V - 1] = Y0[V - 1] + 12.0*z + 4.0*np.sin(3*np.pi*z)
    s2 = np.abs(Y0*(1 + 0.05*rng.standard_normal(Y0.shape))) + 1e-4*rng.random(Y0.shape)*Y0[0]
    s2[V - 1] = Y0[V - 1] + 15.0*rng.random(N) - 3.0
    s3 = s1.copy()
    idx = rng.integers(0, N, size=max(2, N//10))
    s3[2, idx] = -1e-3*rng.random(len(idx))
    s3[4, idx[::2]] = 0.0
    s3[0, idx[1::2]] = -5e-2
    return [s.flatten() for s in (s1, s2, s3)]


def g_m2():
    out, setup = {}, {}
    mi = INP.m2_dme_input()
    for zNo in (20, 100, 1024):
        IV, args, fun = m2_capture(mi, zNo)
        rls, rsc, FunParam = args
        V = FunParam["const"]["varNo"]
        if zNo == 20:
            setup = {"const": tolist(FunParam["const"]), "constBC1": tolist(FunParam["constBC1"]),
                     "ExHe": tolist(FunParam["ExHe"]), "ReSpec": tolist(FunParam["ReSpec"]),
                     "IV": IV.tolist()}
        states = [IV] + m2_states(IV, V, zNo, seed=zNo + 2)
        if zNo == 1024:
            states = states[:2]
        Y = np.array(states)
        t0 = time.time()
        with quiet():
            F = np.array([np.array(fun(0.0, y, *args), dtype=float) for y in Y])
        print("G9 M2 zNo=%d: %d RHS calls in %.1fs" % (zNo, len(Y), time.time() - t0))
        out["rhs_%d_y" % zNo] = Y
        out["rhs_%d_f" % zNo] = F
    # the reference's own RK4 on the M2 RHS (odeSolver.py:17-40), zNo=20, 100 steps of 1e-6 s
    IV, args, fun = m2_capture(mi, 20)
    with quiet():
        traj = ODES.RK4(0.0, 100*1e-6, 100, IV, lambda t, y, p: fun(t, y, *p), args)
    out["rk4_20_traj"] = np.array(traj, dtype=float)[:, ::10]
    out["rk4_20_h"] = np.array(1e-6)
    np.savez_compressed(os.path.join(GOLD, "g9_m2.npz"), **out)
    with open(os.path.join(GOLD, "g9_m2_setup.json"), "w") as f:
        json.dump(setup, f, indent=1)
    print("G9 written")


def g_m2_run(zNo=20, tNo=2, rtol=1e-10, atol=1e-13, method="LSODA"):
    """rmtExe(model M2) end to end with tolerances injected at the solve_ivp call site; the
    reference returns plot lists only (pbReactor.py:835-840), so the end state of every output
    interval is recorded at the call site."""
    import PyREMOT.docs.pbReactor as PBR
    mi = INP.m2_dme_input(ivp=method)
    ends, nfev = [], [0]

    def wrapped(fun, t_span, y0, method=None, t_eval=None, args=None, **kw):
        sol = REAL_SOLVE_IVP(fun, t_span, y0, method=method, t_eval=t_eval, args=args,
                             rtol=rtol, atol=atol)
        ends.append((float(t_span[1]), np.array(sol.y[:, -1], dtype=float)))
        nfev[0] += sol.nfev
        return sol

    PBR.solve_ivp = wrapped
    t0 = time.time()
    try:
        with mesh(zNo, tNo, model="S2"), quiet():
            res = rmtExe(mi)
    finally:
        PBR.solve_ivp = REAL_SOLVE_IVP
    np.savez_compressed(os.path.join(GOLD, "g9_m2_tight_%s.npz" % method.lower()),
                        zNo=zNo, tNo=tNo, rtol=rtol, atol=atol, nfev=nfev[0], wall_s=time.time() - t0,
                        times=np.array([t for t, _ in ends]), states=np.array([y for _, y in ends]),
                        last_leg=np.array([d["leg"] for d in res["resModel"]["dataList"]]),
                        last_y=np.array([d["y"] for d in res["resModel"]["dataList"]], dtype=float))
    print("G9 M2 tight run: nfev=%d wall=%.0fs" % (nfev[0], time.time() - t0))


# --------------------------------------------------------------------------- G10 (plot/export layer)
def synthetic_respack():
    """small resPack of the runN2 schema (pbHomoReactor.py:3664-3696); deterministic numbers"""
    S, N, tNo = 3, 6, 5
    xs = np.linspace(0, 1, N)
    packs = []
    for k in range(tNo):
        ys = np.array([[0.1*(i + 1) + 0.01*k + 0.001*j for j in range(N)] for i in range(S)] +
                      [[500.0 + k + 0.5*j for j in range(N)]])
        packs.append({"modelId": "N2", "processType": "non-iso-thermal", "successStatus": True,
                      "dataShape": (S + 1, N), "labelList": ["A", "B", "C", "Temperature"],
                      "indexList": [S, S + 1, S], "dataTime": 0.1*(k + 1), "dataXs": xs, "dataYs": ys})
    return {"computation-time": 1.234, "dataPack": packs}, tNo


def g_plot():
    import tempfile
    import PyREMOT.solvers.solResultAnalysis as SRA
    from PyREMOT.core.utilities import selectRandomForList, selectFromListByIndex
    from PyREMOT.library.saveResult import saveResultClass as sRes
    out = {"picks": {}}
    for seed in (0, 1, 7, 1234):
        np.random.seed(seed)
        out["picks"][str(seed)] = [int(v) for v in selectRandomForList(list(range(5)), 2)]
    np.random.seed(3)
    out["picks10"] = [int(v) for v in selectRandomForList(list(range(10)), 2)]
    out["select"] = [selectFromListByIndex([], [1, 2, 3]), selectFromListByIndex([2, 0], [1, 2, 3])]
    figs = []
    real = SRA.pltc.plots2D
    SRA.pltc.plots2D = staticmethod(lambda data, xLabel, yLabel, title="": figs.append(
        {"title": title, "xlabel": xLabel, "ylabel": yLabel,
         "lines": [{"leg": d["leg"], "x": tolist(d["x"]), "y": tolist(d["y"])} for d in
                   (data if isinstance(data, list) else [data])]}))
    try:
        resPack, tNo = synthetic_respack()
        np.random.seed(11)
        SRA.plotResultsDynamic(resPack, tNo)
        out["dynamic"] = list(figs)
        figs.clear()
        d = dict(resPack["dataPack"][0])
        d.update({"modelId": "N1", "computation-time": 0.5, "labelList": ["A", "B", "C", "Pressure", "Temperature"],
                  "indexList": [3, 3, 4], "dataYs": np.vstack([d["dataYs"][:3], np.linspace(50, 49, 6), d["dataYs"][3:]])})
        SRA.plotResultsSteadyState([d])
        out["steady"] = list(figs)
    finally:
        SRA.pltc.plots2D = real
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            sRes.saveListToText([1.5, "abc", [1, 2], np.float64(2.25)])
            sRes.saveListToCSV([[1, 2.5, "x"], [3, 4.0, "y,z"]], ["a", "b", "c"])
            out["txt"] = open("saveFile.txt", newline="").read()
            out["csv"] = open("saveFile.csv", newline="").read()
        finally:
            os.chdir(cwd)
    with open(os.path.join(GOLD, "g10_plot_export.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("G10 written")


def g_model_setting():
    """G11: what the reference does when MODEL_SETTING['GaMaCoTe0'] is not "MAX" (the per-species scaling branch of
    pbHomoReactor.py:3461-3463 / :3901-3904 / :3159-3160 and solResultAnalysis.py:285-288).  Recorded, not assumed:
    rmtExe is run end to end for N2 and N1 (zNo = 20) with the setting mutated, and for M2 (which never reads it);
    the outcome - exception type and message, or success - is the fixture the device build is held to."""
    import PyREMOT.docs.modelSetting as MS
    out = {"setting": "FIX", "reference": "PyREMOT %s" % getattr(PyREMOT, "__version__", "1.0.17")}
    cases = {"N2": (INP.dme_notebook_input(period=0.01), "N2"), "N1": (INP.n1_notebook_input(), "N1"),
             "M2": (INP.m2_dme_input(period=0.01), "S2")}
    old = MS.MODEL_SETTING["GaMaCoTe0"]
    MS.MODEL_SETTING["GaMaCoTe0"] = "FIX"            # the one dict object every module imported
    try:
        for name, (mi, key) in cases.items():
            try:
                with mesh(20, 2, key) if key != "N1" else mesh(20, None, key), quiet():
                    rmtExe(mi)
                out[name] = {"raises": None}
            except Exception as e:                   # noqa: BLE001 - the outcome IS the fixture
                out[name] = {"raises": type(e).__name__, "message": str(e)}
        # N1 runs under "FIX" (per-species scale SpCoi0[i], GaMaCoTe0[i] = (vf/zf) SpCoi0[i], while the initial
        # values stay SpCoi0[i]/max(SpCoi0), pbHomoReactor.py:2819-2834, 3159-3162): record its profile and RHS probes
        mi = INP.n1_notebook_input()
        with quiet():
            d = rmtExe(mi)["resModel"][0]
        prof = {k: np.array(d[k], dtype=float) for k in
                ("dataYs", "dataYCons1", "dataYCons2", "dataYTemp1", "dataYTemp2", "dataXs")}
        box = {}

        def fake(fun, t_span, y0, method=None, t_eval=None, args=None, **kw):
            box["IV"] = np.array(y0, float)
            box["params"] = args[0]
            raise _Captured()
        PBH.solve_ivp = fake
        try:
            with quiet():
                try:
                    rmtExe(mi)
                except _Captured:
                    pass
        finally:
            PBH.solve_ivp = REAL_SOLVE_IVP
        ys = [box["IV"], d["dataYCons1"][:, 50].tolist() + [d["dataYs"][6, 50]/5e6, d["dataYTemp1"][50]],
              d["dataYCons1"][:, 100].tolist() + [d["dataYs"][6, 100]/5e6, d["dataYTemp1"][100]]]
        ys = np.array([np.array(y, float) for y in ys])
        with quiet():
            prof["rhs_f"] = np.array([PB.modelEquationN1(0.37, y, box["params"]) for y in ys])
        prof["rhs_y"] = ys
        np.savez_compressed(os.path.join(GOLD, "g11_n1_fix.npz"), **prof)
    finally:
        MS.MODEL_SETTING["GaMaCoTe0"] = old
    with open(os.path.join(GOLD, "g11_model_setting.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(out)


def main(argv):
    os.makedirs(GOLD, exist_ok=True)
    for what in argv:
        if what == "setup":
            g_setup()
        elif what == "rhs":
            g_rhs()
        elif what == "rk4":
            g_rk4()
        elif what.startswith("tight="):
            g_tight(what.split("=", 1)[1])
        elif what.startswith("default="):
            g_default(what.split("=", 1)[1])
        elif what == "multistep":
            g_multistep()
        elif what == "n1":
            g_n1()
        elif what == "helpers":
            g_helpers()
        elif what == "plot":
            g_plot()
        elif what == "setting":
            g_model_setting()
        elif what == "m2":
            g_m2()
        elif what.startswith("m2run"):
            kw = dict(a.split("=") for a in what.split(":")[1:])
            g_m2_run(int(kw.get("zNo", 20)), int(kw.get("tNo", 2)), float(kw.get("rtol", 1e-10)),
                     float(kw.get("atol", 1e-13)), kw.get("method", "LSODA"))
        else:
            raise SystemExit("unknown target " + what)


if __name__ == "__main__":
    main(sys.argv[1:])
