"""Pins the CPU oracle (oracle/n2_oracle.py) against golden vectors produced by the reference
itself (tools/make_golden.py, run in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)/np.maximum(np.abs(b), 1e-300))


def rowwise_err(a, b, V):
    """max |a-b| / max|b| per state row (derivatives can pass through zero)."""
    a = np.asarray(a, float).reshape(V, -1)
    b = np.asarray(b, float).reshape(V, -1)
    return np.max(np.max(np.abs(a - b), axis=1)/np.maximum(np.max(np.abs(b), axis=1), 1e-300))


@pytest.fixture(scope="module")
def g1():
    with open(os.path.join(G, "g1_setup.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", list(INP.ALL_N2_INPUTS))
def test_setup_constants(g1, name):
    mi = INP.ALL_N2_INPUTS[name]()
    g = g1[name]
    np.testing.assert_array_equal(np.array(mi["feed"]["concentration"], float), g["input"]["concentration"])
    assert float(mi["feed"]["volumetric-flowrate"]) == g["input"]["volumetric-flowrate"]
    pr = O.setup_n2(mi, 20)
    c, bc, dap = g["const"], g["constBC1"], g["DimensionlessAnalysisParams"]
    tol = 1e-14
    assert relerr(pr["CrSeAr"], c["CrSeAr"]) < tol
    assert relerr(pr["MoWei"], c["MoWei"]) == 0
    assert np.max(np.abs(pr["StHeRe25"] - np.array(c["StHeRe25"]))) < 1e-9
    assert relerr(pr["GaMiVi"], c["GaMiVi"]) < tol
    assert pr["dz"] == c["dz"] and pr["varNo"] == c["varNo"]
    for k in ("SuGaVe0", "GaDe0", "GaCpMeanMix0", "SpCo0"):
        assert relerr(pr[k], bc[k]) < tol, k
    for k in ("vf", "Cpf", "GaHeCoTe0", "GaMaCoTe0", "Cpif"):
        assert relerr(pr[k], dap[k]) < tol, k
    assert relerr(pr["a"], g["ExHe"]["EfHeTrAr"]) < tol
    np.testing.assert_allclose(pr["IV"], g["IV"], rtol=0, atol=0)
    assert pr["reactionListSorted"] == g["reactionListSorted"]
    assert pr["reactionStochCoeff"] == g["reactionStochCoeff"]


def test_setup_isothermal(g1):
    pr = O.setup_n2(INP.dme_notebook_input(process_type="iso-thermal"), 20)
    assert pr["varNo"] == g1["dme_nb_iso"]["const"]["varNo"] == 6
    np.testing.assert_array_equal(pr["IV"], g1["dme_nb_iso"]["IV"])


def test_helper_probes():
    with open(os.path.join(G, "g7_helpers.json")) as f:
        g = json.load(f)
    comps = g["components"]
    assert tuple(comps) == O.COMPONENT_SYMBOLS
    assert [O._DB[s]["MW"] for s in comps] == g["MW"]
    assert [O._DB[s]["dHf25"] for s in comps] == g["dHf25"]
    mf = np.array(g["molefrac"])
    for p in g["probes"]:
        T = p["T"]
        assert relerr(O.cp(comps, T), p["Cp"]) < 1e-15
        assert relerr(O.cp_mean(comps, T), p["CpMean"]) < 1e-15
        assert relerr(np.dot(mf, O.cp_mean(comps, T)), p["CpMix"]) < 1e-14
        vis = O.gas_viscosity(comps, T)
        assert relerr(vis, p["GaVii"]) < 1e-15
        assert relerr(O.wilke_mixture(len(comps), vis, mf, np.array(g["MW"])), p["GaMiVi"]) < 1e-14
    srt, vec = O.parse_reactions(g["reactions"])
    assert srt == g["reactionListSorted"] and vec == g["reactionStochCoeff"]
    st = [O.standard_enthalpy_of_reaction(r) for r in g["reactions"].values()]
    assert np.max(np.abs(np.array(st) - np.array(g["StHeRe25"]))) < 1e-9
    assert relerr(O.enthalpy_change_of_reaction(srt, 600.0), g["EnChList_600"]) < 1e-13


RHS_CASES = [("dme_nb", 20), ("dme_nb", 100), ("dme_nb", 1024), ("dme_script", 20),
             ("dme_script", 100), ("ch4", 20), ("ch4", 100), ("syn12", 20), ("syn12", 100)]


@pytest.mark.parametrize("name,zNo", RHS_CASES)
def test_rhs_vs_reference(name, zNo):
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["%s_%d_y" % (name, zNo)], g["%s_%d_f" % (name, zNo)]
    pr = O.setup_n2(INP.ALL_N2_INPUTS[name](), zNo)
    fv = O.make_rhs_vec(pr)
    for k, (y, f) in enumerate(zip(Y, F)):
        e = rowwise_err(fv(0.0, y), f, pr["varNo"])
        assert e < 2e-13, (k, e)
        if zNo <= 100:
            e = rowwise_err(O.rhs_loop(0.0, y, pr), f, pr["varNo"])
            assert e < 2e-13, ("loop", k, e)
    # ensemble-batched evaluation equals per-member evaluation
    fb = fv(0.0, Y)
    for k in range(len(Y)):
        np.testing.assert_array_equal(fb[k], fv(0.0, Y[k]))


def test_rhs_isothermal_and_transient():
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    pr = O.setup_n2(INP.dme_notebook_input(process_type="iso-thermal"), 20)
    fv = O.make_rhs_vec(pr)
    for y, f in zip(g["dme_nb_iso_20_y"], g["dme_nb_iso_20_f"]):
        assert rowwise_err(fv(0.0, y), f, 6) < 2e-13
        assert rowwise_err(O.rhs_loop(0.0, y, pr), f, 6) < 2e-13
    from parity import backward_ok
    pr = O.setup_n2(INP.dme_script_input(), 20)
    fv = O.make_rhs_vec(pr)
    for y, f in zip(g["dme_script_20_transient_y"], g["dme_script_20_transient_f"]):
        ok, d = backward_ok(fv(0.0, y), f, fv, y, 7)     # near-steady states: see parity.py
        assert ok, d
        np.testing.assert_array_equal(O.rhs_loop(0.0, y, pr), f)   # the node loop is bit-exact


@pytest.mark.parametrize("name,zNo", [("dme_nb", 20), ("dme_script", 20), ("dme_nb", 100),
                                      ("ch4", 20), ("syn12", 20)])
def test_rk4_trajectory_vs_reference(name, zNo):
    g = np.load(os.path.join(G, "g3_rk4.npz"))
    key = "%s_%d" % (name, zNo)
    traj, h, n, stride = g[key + "_traj"], float(g[key + "_h"]), int(g[key + "_n"]), int(g[key + "_stride"])
    pr = O.setup_n2(INP.ALL_N2_INPUTS[name](), zNo)
    mine = O.rk4(0.0, n*h, n, pr["IV"], O.make_rhs_vec(pr))[:, ::stride]
    assert mine.shape == traj.shape
    scale = np.maximum(np.max(np.abs(traj), axis=1, keepdims=True), 1e-300)
    assert np.max(np.abs(mine - traj)/scale) < 1e-11


def test_tight_end_state_ch4():
    """Oracle RHS under the same scipy BDF settings reproduces the reference run (G4)."""
    g = np.load(os.path.join(G, "g4_tight_ch4_bdf.npz"))
    res = O.run_n2(INP.ch4_input(), zNo=20, method="BDF", rtol=1e-9, atol=1e-12)
    for k in range(5):
        assert relerr(res["dataPack"][k]["dataYs"], g["dataYs_%d" % k]) < 1e-9


def test_n1_profile_vs_reference():
    g = np.load(os.path.join(G, "g6_n1.npz"))
    pr = O.setup_n1(INP.n1_notebook_input())
    for y, f in zip(g["rhs_y"], g["rhs_f"]):
        assert relerr(O.n1_rhs(0.37, y, pr), f) < 1e-12
    res = O.run_n1(INP.n1_notebook_input())
    # same scipy LSODA defaults as the reference -> same step sequence up to rounding
    assert relerr(res["dataYs"], g["dataYs"]) < 1e-7
    out = res["dataYs"][:, -1]
    assert abs(out[7] - 620.85663988) < 1e-4 and abs(out[6] - 4992662.9644) < 1.0


@pytest.mark.parametrize("name", ["dme_nb", "ch4"])
def test_multistep_vs_reference(name):
    """Oracle AdBash3 / PreCorr3 against the reference's own (odeSolver.py:43-102), golden G3b."""
    g = np.load(os.path.join(G, "g3b_multistep.npz"))
    h, n = float(g[name + "_20_h"]), int(g[name + "_20_n"])
    pr = O.setup_n2(INP.ALL_N2_INPUTS[name](), 20)
    f = O.make_rhs_vec(pr)
    for meth, fn in (("PreCorr3", O.precorr3), ("AdBash3", O.adbash3)):
        want = g["%s_20_%s" % (name, meth)]
        got = fn(0.0, n*h, n, pr["IV"], f)[:, [3, n//2, n]]
        scale = np.maximum(np.max(np.abs(want), axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(got - want)/scale) < 1e-11, meth
