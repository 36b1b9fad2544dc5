"""CPU tests of model M2: the oracle's restatement against the
reference-generated golden G9, the host plan, and the generated kernel source's node physics
compiled for the host (oracle/hostemu) - no GPU needed."""
import json
import os

import numpy as np
import pytest

import inputs as INP
from oracle import m2_oracle as M2O
from oracle import n2_oracle as O
from oracle.hostemu import HostEmu
from rmt_app_amd import hipbind, plan, rmtExe
from rmt_app_amd.m2 import pack_interval, result_lists

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rowwise_err(a, b, V):
    a = np.asarray(a, float).reshape(V, -1)
    b = np.asarray(b, float).reshape(V, -1)
    den = np.max(np.abs(b), axis=1)
    den[den == 0] = 1.0
    return np.max(np.max(np.abs(a - b), axis=1)/den)


@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(G, "g9_m2.npz"))


def test_oracle_setup_vs_reference():
    su = json.load(open(os.path.join(G, "g9_m2_setup.json")))
    pr = M2O.setup_m2(INP.m2_dme_input(), 20)
    assert np.array_equal(pr["IV"], np.array(su["IV"]))
    c = su["const"]
    assert pr["CrSeAr"] == c["CrSeAr"] and pr["dz"] == c["dz"] and pr["GaMiVi"] == c["GaMiVi"]
    np.testing.assert_allclose(pr["StHeRe25"], c["StHeRe25"], rtol=1e-15)
    np.testing.assert_allclose(pr["MoWei"], c["MoWei"], rtol=0)
    b = su["constBC1"]
    assert pr["VoFlRa0"] == b["VoFlRa0"] and pr["SpCo0"] == b["SpCo0"] and pr["P0"] == b["P0"]


@pytest.mark.parametrize("zNo", [20, 100, 1024])
def test_oracle_rhs_vs_reference(g9, zNo):
    pr = M2O.setup_m2(INP.m2_dme_input(), zNo)
    Y, F = g9["rhs_%d_y" % zNo], g9["rhs_%d_f" % zNo]
    fv = M2O.make_rhs_vec(pr)
    for y, f in zip(Y, F):
        assert rowwise_err(fv(0.0, y), f, 7) < 1e-13
        if zNo <= 100:
            assert np.array_equal(M2O.rhs_loop(0.0, y, pr), f)        # same operation order: bit-exact
    assert rowwise_err(fv(0.0, Y), F, 7*len(Y)) < 1e-13               # ensemble form


def test_oracle_rk4_vs_reference(g9):
    pr = M2O.setup_m2(INP.m2_dme_input(), 20)
    traj = np.array(O.rk4(0.0, 100e-6, 100, pr["IV"], M2O.make_rhs_vec(pr)))[:, ::10]
    ref = g9["rk4_20_traj"]
    scale = np.max(np.abs(ref), axis=1, keepdims=True)
    assert np.max(np.abs(traj - ref)/scale) < 1e-13


def test_plan_constants_and_initial_state(g9):
    mi = INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    assert mech.model == "M2" and mech.V == 7 and not mech.iso
    nm, row = plan.member_constants_m2(mi, mech, 20)
    assert np.array_equal(plan.initial_state_m2(nm, mech, 20), g9["rhs_20_y"][0])
    F = plan.MEMBER_FIELDS
    su = json.load(open(os.path.join(G, "g9_m2_setup.json")))
    assert row[F["INV_DZ"]] == 1.0/su["const"]["dz"] and row[F["P0"]] == su["constBC1"]["P0"]
    np.testing.assert_array_equal(row[F["CIN"]:F["CIN"] + 6], su["constBC1"]["SpCoi0"])
    assert "#define RMT_MODEL 2" in mech.prelude()
    assert "#define RMT_MODEL 0" in plan.Mechanism(INP.dme_script_input()).prelude()


@pytest.mark.parametrize("zNo", [20, 100, 1024])
def test_generated_m2_node_physics_vs_reference(g9, zNo):
    mi = INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    emu = HostEmu(mech.source(hipbind.kernel_template(), False, 64, 1), tag="m2")
    _, row = plan.member_constants_m2(mi, mech, zNo)
    Y, F = g9["rhs_%d_y" % zNo], g9["rhs_%d_f" % zNo]
    out, flags = emu.rhs(Y, np.tile(row, (len(Y), 1)), zNo)
    assert not flags.any()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], 7) < 1e-12


def test_emulated_m2_rk4_vs_reference(g9):
    mi = INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    emu = HostEmu(mech.source(hipbind.kernel_template(), False, 64, 1), tag="m2")
    _, row = plan.member_constants_m2(mi, mech, 20)
    traj = g9["rk4_20_traj"]
    y, flags = emu.rk4(traj[:, 0], row, 20, float(g9["rk4_20_h"]), 100)
    assert not flags.any()
    assert rowwise_err(y[0], traj[:, -1], 7) < 1e-12


@pytest.mark.parametrize("zNo", [20, 100])
def test_m2_analytic_node_jacobian_and_coupling_vs_forward_differences(g9, zNo):
    """Model M2's rmt_node_jac (host build of the generated source): -d f_z/d y_z and the upwind coupling
    d f_r/d up_r against the forward differences they replace in the stiff stepper, at the reference-generated G9
    states: equal to FD accuracy (relative to the node's largest entry)."""
    mi = INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    emu = HostEmu(mech.source(hipbind.kernel_template(), False, 64, 1, None, {"RMT_WITH_ROS4": "1"}), tag="m2jac",
                  openmp=False)
    _, row = plan.member_constants_m2(mi, mech, zNo)
    V = mech.V
    checked = 0
    for Y in g9["rhs_%d_y" % zNo]:
        jan, jfd, lan, lfd = emu.node_jac(Y, row, zNo, coupling=True)
        ok = np.all(Y.reshape(V, zNo)[:mech.S] > 1e-30, axis=0)
        ok[1:] &= ok[:-1]
        scale = np.max(np.abs(jfd), axis=(1, 2), keepdims=True)
        assert np.max((np.abs(jan - jfd)/scale)[ok]) < 2e-5
        # coupling: one convective coefficient for all species, one for T; the forward difference is only
        # well conditioned for the abundant species (a trace species moves its balance by ~1e-11) and for T
        big = np.argmax(Y.reshape(V, zNo)[:mech.S, 0])
        for col in (big, V - 1):
            assert np.max(np.abs(lan[ok, col] - lfd[ok, col])/np.abs(lfd[ok, col])) < 2e-5
        assert np.all(lan[ok] > 0) and np.all(lan[:, :mech.S] == lan[:, :1])
        checked += int(ok.sum())
    assert checked >= zNo


def test_m2_kernels_cross_compile_for_gfx950():
    mech = plan.Mechanism(INP.m2_dme_input())
    tpl = hipbind.kernel_template()
    code, _ = hipbind.compile_source(mech.source(tpl, False, 128, 1, None, {"RMT_WITH_ROS4": "1"}))
    for sym in (b"rmt_n2_rhs", b"rmt_n2_rk4_reg", b"rmt_n2_rk4_chain", b"rmt_n2_rk4_mem", b"rmt_n2_rk45_mem",
                b"rmt_n2_ros4_mem"):
        assert sym in code


def test_m2_result_lists_match_reference_schema():
    """runM2 returns the temperature series per output time (pbReactor.py:806-840)"""
    p = os.path.join(G, "g9_m2_tight_lsoda.npz")
    if not os.path.exists(p):
        pytest.skip("tight reference run of M2 not generated")
    g = np.load(p)
    zNo, tNo = int(g["zNo"]), int(g["tNo"])
    mech = plan.Mechanism(INP.m2_dme_input())
    packs = [pack_interval(s, mech, zNo, t) for s, t in zip(g["states"], g["times"])]
    res = result_lists(packs, 1, zNo, np.linspace(0, 10, tNo + 1))
    assert [d["leg"] for d in res["dataList"]] == [str(s) for s in g["last_leg"]]
    np.testing.assert_allclose(np.array([d["y"] for d in res["dataList"]]), g["last_y"], rtol=1e-15)
    # and the oracle's own run reproduces the tight reference end states
    want, ends = M2O.run_m2(INP.m2_dme_input(), zNo, tNo, "LSODA", rtol=1e-10, atol=1e-13)
    assert rowwise_err(ends[-1], g["states"][-1], 7) < 1e-7


def test_m2_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    mi = INP.m2_dme_input(ivp="hip-rk4")
    mi["solver-config"]["quiet"] = True
    with pytest.raises(hipbind.RmtN2Error):
        rmtExe(mi)
