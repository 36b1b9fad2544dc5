"""GPU parity tests of model M2 (the dimensional dynamic model, pbReactor.py:552-1165; SURVEY.md
section 8(f) rank 3) through the C-ABI library: RHS against the reference-generated golden G9
(<= 1e-12 row-relative), fixed-step RK4 against the reference's own RK4 (<= 1e-9), whole runs
against the tight-tolerance reference run (<= 1e-6 on outlet mole fractions and temperature) and
against SciPy driving the oracle's RHS."""
import os

import numpy as np
import pytest

import inputs as INP
from oracle import m2_oracle as M2O
from oracle import n2_oracle as O
from rmt_app_amd import plan, rmtExe
from rmt_app_amd.n2 import N2Device

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rowwise_err(a, b, V):
    a = np.asarray(a, float).reshape(V, -1)
    b = np.asarray(b, float).reshape(V, -1)
    den = np.max(np.abs(b), axis=1)
    den[den == 0] = 1.0
    return np.max(np.max(np.abs(a - b), axis=1)/den)


def make_device(zNo, E=1, mi=None, **kw):
    mi = mi or INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants_m2(mi, mech, zNo)
    return mi, mech, nm, N2Device(mech, np.tile(row, (E, 1)), zNo, **kw)


@pytest.mark.parametrize("zNo,block,npt", [(20, None, None), (100, None, None), (100, 64, 1),
                                           (1024, None, None), (1024, 256, 1), (1024, 1024, 1)])
def test_m2_rhs_vs_reference_golden(zNo, block, npt):
    """incl. N > workgroup (carry between node blocks, Newton sweeps per block) and a ragged tail"""
    g = np.load(os.path.join(G, "g9_m2.npz"))
    Y, F = g["rhs_%d_y" % zNo], g["rhs_%d_f" % zNo]
    _, mech, _, dev = make_device(zNo, E=len(Y), block=block, npt=npt)
    out = dev.rhs(dev.to_device(Y)).cpu().numpy()
    assert not dev.status().any()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], mech.V) < 1e-12, k
    dev.close()


@pytest.mark.parametrize("mode", ["reg", "mem"])
def test_m2_rk4_vs_reference_trajectory(mode):
    g = np.load(os.path.join(G, "g9_m2.npz"))
    traj, h = g["rk4_20_traj"], float(g["rk4_20_h"])          # every 10th of 100 steps
    _, mech, nm, dev = make_device(20)
    dev.set_mode(mode)
    y = dev.to_device(plan.initial_state_m2(nm, mech, 20))
    scale = np.maximum(np.max(np.abs(traj), axis=1), 1e-300)
    for col in range(1, traj.shape[1]):
        dev.rk4(y, h, 10)
        assert np.max(np.abs(y.cpu().numpy()[0] - traj[:, col])/scale) < 1e-9, col
    assert not dev.status().any()
    dev.close()


@pytest.mark.parametrize("N,block,npt", [(1000, 512, 2), (300, 64, 1), (1021, 256, 1)])
def test_m2_rk4_geometries_agree_with_oracle(N, block, npt):
    mi, mech, nm, dev = make_device(N, block=block, npt=npt)
    y = dev.to_device(plan.initial_state_m2(nm, mech, N))
    dev.rk4(y, 2e-6, 12)
    pr = M2O.setup_m2(mi, N)
    want = O.rk4(0.0, 12*2e-6, 12, pr["IV"], M2O.make_rhs_vec(pr), keep=False)
    assert rowwise_err(y.cpu().numpy()[0], want, mech.V) < 1e-11
    assert not dev.status().any()
    dev.close()


def test_m2_pressure_newton_flag_and_large_drop():
    """A bed with a pressure drop of 11 % of P (60 um particles): the default three Newton sweeps still
    reproduce the sequential march to rounding; with ONE sweep the kernel must raise RMT_FLAG_PRESSURE."""
    mi = INP.m2_dme_input()
    mi["reactor"] = dict(mi["reactor"], PaDi=6e-5)
    N = 200
    pr = M2O.setup_m2(mi, N)
    want = M2O.make_rhs_vec(pr)(0.0, pr["IV"])
    _, mech, nm, dev = make_device(N, mi=mi)
    out = dev.rhs(dev.to_device(pr["IV"])).cpu().numpy()[0]
    assert not dev.status().any()
    assert rowwise_err(out, want, mech.V) < 1e-12
    dev.close()
    _, mech, nm, dev = make_device(N, mi=mi, defines={"RMT_M2_NEWTON": 1})
    dev.rhs(dev.to_device(pr["IV"]))
    assert dev.status()[0] & 32
    dev.close()


def test_m2_ensemble_members_differ():
    """T/P sweep through rmtExe's ensemble option: every member equals its own single run"""
    mi = INP.m2_dme_input(ivp="hip-rk4", period=2e-3)
    mi["solver-config"].update({"zNo": 64, "tNo": 1, "dt": 2e-6, "quiet": True,
                                "ensemble": {"temperature": [513.0, 533.0], "pressure": [4e6, 6e6]}})
    res = rmtExe(mi)["resModel"]
    assert len(res["ensemble"]) == 4
    from rmt_app_amd.ensemble import expand_members
    members = expand_members(mi, mi["solver-config"]["ensemble"])
    for e in (0, 3):
        single = dict(members[e])
        single["solver-config"] = {k: v for k, v in mi["solver-config"].items() if k != "ensemble"}
        one = rmtExe(single)["resModel"]
        np.testing.assert_allclose(res["ensemble"][e]["dataPack"][-1]["dataYs"], one["dataPack"][-1]["dataYs"],
                                   rtol=1e-12, atol=0)
    assert np.max(np.abs(res["ensemble"][0]["dataPack"][-1]["dataYs"] -
                         res["ensemble"][3]["dataPack"][-1]["dataYs"])) > 1e-3


def _tight():
    p = os.path.join(G, "g9_m2_tight_lsoda.npz")
    if not os.path.exists(p):
        pytest.skip("tight reference run of M2 not generated")
    return np.load(p)


@pytest.mark.parametrize("ivp,tol", [("LSODA", 1e-6), ("hip-rk45", 1e-6)])
def test_m2_end_to_end_vs_tight_reference(ivp, tol):
    """rmtExe(model M2) with the reference's own test input (period 10 s) against the reference run
    under LSODA rtol 1e-10: outlet mole fractions and temperature, and the returned plot lists."""
    g = _tight()
    zNo, tNo = int(g["zNo"]), int(g["tNo"])
    mi = INP.m2_dme_input(ivp=ivp)
    mi["solver-config"].update({"zNo": zNo, "tNo": tNo, "quiet": True})
    res = rmtExe(mi)["resModel"]
    V = 7
    for k in range(tNo):
        ref = g["states"][k].reshape(V, zNo)
        ref_ys = np.concatenate((ref[:6]/np.sum(ref[:6], axis=0), ref[6:7]), axis=0)
        got = res["dataPack"][k]["dataYs"]
        err = np.max(np.abs(got[:, -1] - ref_ys[:, -1])/np.maximum(np.abs(ref_ys[:, -1]), 1e-3))
        assert err < tol, (k, err)
        assert abs(res["dataPack"][k]["dataTime"] - float(g["times"][k])) < 1e-12
    assert [d["leg"] for d in res["dataList"]] == [str(s) for s in g["last_leg"]]
    np.testing.assert_allclose(np.array([d["y"] for d in res["dataList"]]), g["last_y"], rtol=1e-6)
    np.testing.assert_allclose(res["XYList"][0][0], np.linspace(0, 1, zNo))


def test_m2_rk4_end_to_end_vs_scipy_on_oracle_rhs():
    """explicit device RK4 over 20 ms against SciPy LSODA (rtol 1e-11) on the oracle's M2 RHS"""
    from scipy.integrate import solve_ivp
    mi = INP.m2_dme_input(ivp="hip-rk4", period=0.02)
    mi["solver-config"].update({"zNo": 40, "tNo": 2, "dt": 2e-6, "quiet": True})
    res = rmtExe(mi)["resModel"]
    pr = M2O.setup_m2(mi, 40)
    sol = solve_ivp(M2O.make_rhs_vec(pr), (0, 0.02), pr["IV"], method="LSODA", rtol=1e-11, atol=1e-14)
    ref = M2O.pack_interval(sol.y[:, -1], pr)["dataYs"]
    got = res["dataPack"][-1]["dataYs"]
    assert np.max(np.abs(got - ref)/np.maximum(np.abs(ref), 1e-6)) < 1e-7


@pytest.mark.parametrize("N,E,block,npt,tagged", [(1000, 1, 128, 1, 1), (4096, 3, 512, 2, 1), (4100, 2, 256, 1, 1),
                                                  (1000, 1, 128, 1, 0), (4096, 3, 512, 2, 0)])
def test_m2_chained_workgroups_match_memory_stepper(N, E, block, npt, tagged):
    """model M2 spread over several workgroups (upstream record awaited before the Newton sweeps):
    same result as the memory-resident stepper and as the oracle's RK4 - over the tagged-word links (default)
    and over the flag protocol they replaced (RMT_CHAIN_TAGGED 0)"""
    mi, mech, nm, dev = make_device(N, E=E, block=block, npt=npt, defines={"RMT_CHAIN_TAGGED": str(tagged)})
    IV = np.tile(plan.initial_state_m2(nm, mech, N), (E, 1))
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 40)
    assert not dev.status().any()
    dev.set_mode("mem")
    y2 = dev.to_device(IV)
    dev.rk4(y2, 2e-6, 40)
    a, b = y.cpu().numpy(), y2.cpu().numpy()
    for e in range(E):
        assert rowwise_err(a[e], b[e], mech.V) < 1e-12
    if N <= 1000:
        pr = M2O.setup_m2(mi, N)
        want = O.rk4(0.0, 40*2e-6, 40, pr["IV"], M2O.make_rhs_vec(pr), keep=False)
        assert rowwise_err(a[0], want, mech.V) < 1e-11
    dev.close()


def _m2_members(N, E):
    mi = INP.m2_dme_input()
    mech = plan.Mechanism(mi)
    rows, ivs = [], []
    for e in range(E):
        m = INP.m2_dme_input()
        m["operating-conditions"]["temperature"] = mi["operating-conditions"]["temperature"] + 3*(e % 5)
        nm, row = plan.member_constants_m2(m, mech, N)
        rows.append(row), ivs.append(plan.initial_state_m2(nm, mech, N))
    return mech, np.array(rows), np.array(ivs)


@pytest.mark.parametrize("N,E", [(2500, 4), (2048, 150)])
def test_m2_chained_rk45_matches_memory_resident_kernel(N, E):
    """Model M2 under rmt_n2_rk45_chain (tagged-word links; the upstream chunk's record is awaited BEFORE the
    pressure sweeps, sent right after them): same step sequences as rmt_n2_rk45_mem, end states equal to rounding
    at a tolerance that keeps the explicit pair inside its stability region (at rtol 1e-6 the stability-limited
    steps amplify rounding differences to ~1e-7, tools/microbench/exp_m2_chain.py)."""
    from rmt_app_amd.n2 import rk45_geometry
    mech, rows, IV = _m2_members(N, E)
    block, npt, defs = rk45_geometry(mech.V, N)
    assert block*npt < N
    out, stats = {}, {}
    for mode in ("mem", "chain"):
        dev = N2Device(mech, rows, N, block=block, npt=npt, defines=defs)
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.rk45(y, 0.0, 4e-4, 1e-9, 1e-12, 1e-6, 10**7)
        dev.rk45(y, 4e-4, 1e-3, 1e-9, 1e-12, -1e-6, 10**7)
        assert not dev.status().any(), mode
        out[mode], stats[mode] = y.cpu().numpy(), dev.rk45_stats()
        dev.close()
    assert np.array_equal(stats["chain"]["accepted"], stats["mem"]["accepted"])
    assert np.array_equal(stats["chain"]["rejected"], stats["mem"]["rejected"])
    for e in range(E):
        assert rowwise_err(out["chain"][e], out["mem"][e], mech.V) < 1e-11


@pytest.mark.parametrize("N,E", [(2000, 1), (1500, 9)])
def test_m2_chained_stiff_stepper_matches_one_workgroup(N, E):
    """Model M2 under rmt_n2_ros4_chain: the controller's step history of the one-workgroup kernel and its state to
    the accuracy of the linear solves."""
    from rmt_app_amd.settings import DEVICE_DEFAULTS as D
    mech, rows, IV = _m2_members(N, E)
    out, stats = {}, {}
    for mode in ("mem", "chain"):
        dev = N2Device(mech, rows, N, block=256, npt=1, features=("ros4",))
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.ros4(y, 0.0, 1.0, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)
        assert not dev.status().any(), mode
        out[mode], stats[mode] = y.cpu().numpy(), dev.rk45_stats()
        dev.close()
    assert np.array_equal(stats["chain"]["accepted"], stats["mem"]["accepted"])
    for e in range(E):
        assert rowwise_err(out["chain"][e], out["mem"][e], mech.V) < 1e-8


# ----------------------------------------------------------------------------- chained steppers vs the oracle (not vs each other)
def test_m2_chained_rk45_vs_oracle_controller():
    """Model M2 under rmt_n2_rk45_chain at a mesh that forces 5 chunks (N = 300, 64-node chunks, ragged tail), two
    reactors with their own step sequences: the accept / reject history of the oracle's Dormand-Prince controller
    driving the ORACLE's M2 right-hand side (itself pinned to the reference's modelEquationM2, golden G9), end state
    within 50 rtol."""
    N, t1, rtol, atol, h0 = 300, 6e-3, 1e-7, 1e-10, 1e-6
    base = INP.m2_dme_input()
    mech = plan.Mechanism(base)
    mis, rows, ivs = [], [], []
    for dT in (0.0, 12.0):
        mi = INP.m2_dme_input()
        mi["operating-conditions"]["temperature"] = base["operating-conditions"]["temperature"] + dT
        nm, row = plan.member_constants_m2(mi, mech, N)
        mis.append(mi), rows.append(row), ivs.append(plan.initial_state_m2(nm, mech, N))
    dev = N2Device(mech, np.array(rows), N, block=64, npt=1, defines={"RMT_RK45_LDS": "2"})
    dev.set_mode("chain")
    y = dev.to_device(np.array(ivs))
    dev.rk45(y, 0.0, t1, rtol, atol, h0, 10**7)
    assert not dev.status().any()
    st, got = dev.rk45_stats(), y.cpu().numpy()
    for e, mi in enumerate(mis):
        pr = M2O.setup_m2(mi, N)
        want, ost = O.rk45(M2O.make_rhs_vec(pr), 0.0, t1, pr["IV"], rtol, atol, h0)
        assert st["t_end"][e] == t1
        assert abs(int(st["accepted"][e]) - ost["accepted"]) <= max(2, 0.02*ost["accepted"]), (e, st, ost)
        assert abs(int(st["rejected"][e]) - ost["rejected"]) <= max(3, 0.05*ost["accepted"]), (e, st, ost)
        assert rowwise_err(got[e], want, mech.V) < 50*rtol, e
    dev.close()


_M2_TIGHT = {}


def _m2_tight_reference(N, t1):
    """SciPy LSODA at rtol 1e-11 on the oracle's M2 right-hand side (computed once per session: ~20 s of CPU)."""
    from scipy.integrate import solve_ivp
    if (N, t1) not in _M2_TIGHT:
        pr = M2O.setup_m2(INP.m2_dme_input(), N)
        sol = solve_ivp(M2O.make_rhs_vec(pr), (0, t1), pr["IV"], method="LSODA", rtol=1e-11, atol=1e-14)
        assert sol.success
        _M2_TIGHT[(N, t1)] = (pr, sol.y[:, -1])
    return _M2_TIGHT[(N, t1)]


@pytest.mark.parametrize("stepper,tol", [("rk4", 1e-8), ("rk45", 2e-7), ("ros4", 1e-6)])
def test_m2_chained_steppers_vs_scipy_on_the_oracle_rhs(stepper, tol):
    """The three chained M2 steppers (3 chunks of 64 nodes) against SciPy's LSODA at rtol 1e-11 integrating the oracle's
    M2 right-hand side over the first 10 ms: whole state and outlet mole fractions / temperature."""
    from rmt_app_amd.settings import DEVICE_DEFAULTS as D
    N, t1 = 192, 0.01
    kw = {"features": ("ros4",)} if stepper == "ros4" else {"defines": {"RMT_RK45_LDS": "2"}}
    mi, mech, nm, dev = make_device(N, block=64, npt=1, **kw)
    dev.set_mode("chain")
    y = dev.to_device(plan.initial_state_m2(nm, mech, N))
    if stepper == "rk4":
        dev.rk4(y, 2e-6, 5000)
    elif stepper == "rk45":
        dev.rk45(y, 0.0, t1, 1e-9, 1e-12, 1e-6, 10**7)
    else:
        dev.ros4(y, 0.0, t1, 1e-8, 1e-11, D["ros4-h0"], 10**7)
    assert not dev.status().any()
    got = y.cpu().numpy()[0]
    dev.close()
    pr, want = _m2_tight_reference(N, t1)
    assert rowwise_err(got, want, mech.V) < tol
    a, b = M2O.pack_interval(got, pr)["dataYs"][:, -1], M2O.pack_interval(want, pr)["dataYs"][:, -1]
    assert np.max(np.abs(a - b)/np.abs(b)) < tol
