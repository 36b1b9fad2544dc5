"""GPU parity tests for BASELINE configs[4] (12-species / 8-reaction mechanism, adaptive RK45 with
per-reactor step control, fp32 vs fp64) and for the full sizes of configs[2]/[3] that round 1 only
covered through properties.

References: the 12-species (`syn12`) and the adiabatic CH4 runs of the REFERENCE itself under
SciPy BDF at tight tolerances (`tests/golden/g4_tight_syn12_bdf.npz`, `g4_tight_ch4_bdf.npz`,
written by tools/make_golden.py from `solve_ivp` at PyREMOT/docs/pbHomoReactor.py:3609) and the CPU
oracle's restatement of the Dormand-Prince controller (oracle/n2_oracle.py rk45).
Tolerances: <= 1e-6 relative on outlet mole fractions and temperature (north_star) for fp64;
fp32 is asserted against the band it actually reaches (profiles/round1_fp32.md)."""
import os

import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import plan, rmtExe
from rmt_app_amd.n2 import N2Device

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def outlet_rel(dp, g, ntimes=5):
    worst = 0.0
    for k in range(ntimes):
        a, b = dp[k]["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
        worst = max(worst, float(np.max(np.abs(a - b)/np.abs(b))))
    return worst


@pytest.mark.parametrize("name", ["syn12", "ch4"])
@pytest.mark.parametrize("ivp,extra", [("hip-rk45", {"rtol": 1e-8, "atol": 1e-11}), ("hip-ros4", {}),
                                       ("RK45", {"rtol": 1e-8, "atol": 1e-11}), ("BDF", {})])
def test_rmtexe_other_mechanisms_end_to_end_vs_reference_bdf(name, ivp, extra):
    """rmtExe on the 12-species mechanism and on the adiabatic CH4 case, adaptive explicit pair and
    stiff stepper (also under SciPy's method names, which the reference passes to solve_ivp,
    pbHomoReactor.py:3609), all five output times, against the reference run under BDF."""
    g = np.load(os.path.join(G, "g4_tight_%s_bdf.npz" % name))
    mi = INP.ALL_N2_INPUTS[name](ivp=ivp)
    mi["solver-config"].update(dict(extra, quiet=True))
    res = rmtExe(mi)
    dp = res["resModel"]["dataPack"]
    assert len(dp) == 5
    for k in range(5):
        assert abs(dp[k]["dataTime"] - float(g["dataTime_%d" % k])) < 1e-12
        assert dp[k]["dataYs"].shape == g["dataYs_%d" % k].shape
    worst = outlet_rel(dp, g)
    assert worst < 1e-6, worst
    # whole profiles: absolute mole-fraction difference (BASELINE's max |dMoFri|) and |dT|/T
    for k in range(5):
        a, b = dp[k]["dataYs"], g["dataYs_%d" % k]
        assert np.max(np.abs(a[:-1] - b[:-1])) < 1e-6, k
        assert np.max(np.abs(a[-1] - b[-1])/b[-1]) < 1e-6, k


@pytest.mark.parametrize("rtol", [1e-4, 1e-6, 1e-8])
def test_rk45_rtol_sweep_vs_oracle_controller_syn12(rtol):
    """configs[4]'s tolerance sweep, fp64: same accept/reject history as the oracle's controller and
    an end state within 50 rtol, for rtol = 1e-4, 1e-6, 1e-8 on the 12-species mechanism; two
    reactors with different inlet temperatures in ONE launch keep their own step sequences."""
    zNo, t1 = 20, 2e-2
    mech = plan.Mechanism(INP.syn12_input())
    rows, named, mis = [], [], []
    for T in (600, 640):
        mi = INP.syn12_input()
        mi["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(mi, mech, zNo)
        rows.append(row), named.append(nm), mis.append(mi)
    dev = N2Device(mech, np.array(rows), zNo)
    y = dev.to_device(np.array([plan.initial_state(nm, mech, zNo) for nm in named]))
    atol, h0 = 1e-3*rtol, 1e-6
    dev.rk45(y, 0.0, t1, rtol, atol, h0, 10**7)
    assert not dev.status().any()
    st = dev.rk45_stats()
    got = y.cpu().numpy()
    V = mech.V
    for e, mi in enumerate(mis):
        pr = O.setup_n2(mi, zNo)
        want, ost = O.rk45(O.make_rhs_vec(pr), 0.0, t1, pr["IV"], rtol, atol, h0)
        assert st["t_end"][e] == t1
        assert abs(int(st["accepted"][e]) - ost["accepted"]) <= max(2, 0.02*ost["accepted"]), (e, st, ost)
        assert abs(int(st["rejected"][e]) - ost["rejected"]) <= max(3, 0.05*ost["accepted"]), (e, st, ost)
        scale = np.max(np.abs(want.reshape(V, zNo)), axis=1, keepdims=True)
        scale[scale == 0] = 1.0
        assert np.max(np.abs(got[e].reshape(V, zNo) - want.reshape(V, zNo))/scale) < 50*rtol, e
    dev.close()


def _outlet_x_theta(y, V, N):
    a = np.asarray(y, dtype=np.float64).reshape(V, N)
    x = a[:V - 1]/a[:V - 1].sum(0)
    return x[:, -1], a[V - 1, -1]


def test_fp32_vs_fp64_tolerance_study_syn12():
    """configs[4]'s precision leg on the device (what tools/fp32_study.py tabulates in
    profiles/round1_fp32.md): `dtype: fp32` = state and kinetics in float with the hardware
    transcendentals, pressure scan in fp64.
      * RK45, 4 ms, rtol 1e-4 / 1e-6: fp32 lands within ~1e-6 relative of the tight fp64 run on the
        outlet mole fractions (measured 1.4e-7 / 1.6e-7) - single-precision accurate, not better;
      * rtol 1e-8 in fp32 cannot be reached: the controller takes >= 2x the fp64 steps (58 vs 13)
        and the error does not fall below the rtol 1e-6 result;
      * 2000 fixed RK4 steps in fp32 accumulate rounding to ~1e-4 (measured 8.9e-5): fp32 cannot carry
        the 1e-6 requirement for this mechanism."""
    name, N, E, t1 = "syn12", 512, 4, 4e-3
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    IV = np.tile(plan.initial_state(nm, mech, N), (E, 1))
    V = mech.V

    def run45(fp32, rtol):
        dev = N2Device(mech, np.tile(row, (E, 1)), N, fp32=fp32)
        y = dev.to_device(IV)
        dev.rk45(y, 0.0, t1, rtol, 1e-3*rtol, 1e-6, 10**7)
        assert not dev.status().any()
        st = dev.rk45_stats()
        out = y.cpu().numpy().astype(np.float64)[0]
        dev.close()
        return out, int(st["accepted"][0]) + int(st["rejected"][0])

    tight, _ = run45(False, 1e-11)
    xt, tt = _outlet_x_theta(tight, V, N)
    errs, steps = {}, {}
    for fp32 in (False, True):
        for rtol in (1e-4, 1e-6, 1e-8):
            out, n = run45(fp32, rtol)
            x, th = _outlet_x_theta(out, V, N)
            errs[(fp32, rtol)] = max(float(np.max(np.abs(x - xt)/xt)), abs(th - tt)/(1 + abs(tt)))
            steps[(fp32, rtol)] = n
    print("fp32 study (outlet error vs fp64 rtol 1e-11): ", errs, steps)
    for rtol in (1e-4, 1e-6, 1e-8):
        assert errs[(False, rtol)] < 50*rtol                       # fp64 follows the tolerance
        assert 1e-9 < errs[(True, rtol)] < 5e-6, (rtol, errs)      # fp32: single-precision band
    assert errs[(False, 1e-8)] < 1e-2*errs[(True, 1e-8)]           # fp64 at 1e-8 is far below the fp32 floor
    assert steps[(True, 1e-8)] >= 2*steps[(False, 1e-8)], steps    # unreachable tolerance: steps multiply
    # fixed-step RK4: 2000 steps of rounding
    outs = {}
    for fp32 in (False, True):
        dev = N2Device(mech, np.tile(row, (E, 1)), N, fp32=fp32)
        y = dev.to_device(IV)
        dev.rk4(y, 2e-6, 2000)
        assert not dev.status().any()
        outs[fp32] = y.cpu().numpy().astype(np.float64)[0]
        dev.close()
    xa, ta = _outlet_x_theta(outs[True], V, N)
    xb, tb = _outlet_x_theta(outs[False], V, N)
    e4 = float(np.max(np.abs(xa - xb)/xb))
    print("fp32 RK4 x2000 outlet error vs fp64: %.3e" % e4)
    assert 1e-6 < e4 < 1e-3, e4
    xq, _ = _outlet_x_theta(tight, V, N)
    assert float(np.max(np.abs(xb - xq)/xq)) < 1e-6               # fp64 RK4 agrees with tight RK45


def test_rmtexe_fp32_syn12_end_to_end_band():
    """rmtExe(dtype=fp32) on the 12-species case against the reference's BDF run: runs to the end on
    both adaptive steppers and lands in the single-precision band (NOT within the 1e-6 requirement -
    that is what the study is for)."""
    g = np.load(os.path.join(G, "g4_tight_syn12_bdf.npz"))
    for ivp, extra in (("hip-rk45", {"rtol": 1e-5, "atol": 1e-8}), ("hip-ros4", {"rtol": 1e-4, "atol": 1e-7})):
        mi = INP.syn12_input(ivp=ivp)
        mi["solver-config"].update(dict(extra, quiet=True, dtype="fp32"))
        dp = rmtExe(mi)["resModel"]["dataPack"]
        worst = outlet_rel(dp, g)
        print("fp32 %s syn12 outlet error vs reference BDF: %.3e" % (ivp, worst))
        assert worst < 2e-3, (ivp, worst)


def test_full_size_16384_nodes_vs_oracle_direct():
    """BASELINE configs[2] at 16384 nodes, directly against the vectorised oracle (10 RK4 steps) in the
    default (chained on-chip) and the memory-resident stepper."""
    N = 16384
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    pr = O.setup_n2(mi, N)
    want = O.rk4(0.0, 10*2e-6, 10, pr["IV"], O.make_rhs_vec(pr), keep=False).reshape(7, N)
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    dev = N2Device(mech, row, N)
    for mode in ("auto", "mem"):
        dev.set_mode(mode)
        y = dev.to_device(plan.initial_state(nm, mech, N))
        dev.rk4(y, 2e-6, 10)
        assert not dev.status().any()
        assert np.max(np.abs(y.cpu().numpy()[0].reshape(7, N) - want)/scale) < 1e-11, mode
    f = dev.rhs(dev.to_device(pr["IV"])).cpu().numpy()[0].reshape(7, N)
    wf = O.make_rhs_vec(pr)(0.0, pr["IV"]).reshape(7, N)
    assert np.max(np.max(np.abs(f - wf), axis=1)/np.max(np.abs(wf), axis=1)) < 1e-12
    dev.close()


def test_full_size_ensemble_2048_members_vs_oracle():
    """BASELINE configs[3] at its full size on one GPU: 2048 reactors x 1024 nodes (the 64 x 32 inlet-T /
    pressure sweep), 8 RK4 steps; sampled members against the oracle, neighbours differ."""
    from rmt_app_amd.ensemble import expand_members
    N = 1024
    base = INP.dme_notebook_input()
    members = expand_members(base, {"temperature": list(np.linspace(503.0, 543.0, 64)),
                                    "pressure": list(np.linspace(3.0e6, 7.0e6, 32))})
    assert len(members) == 2048
    mech = plan.Mechanism(base)
    pairs = [plan.member_constants(mi, mech, N) for mi in members]
    rows = np.array([r for _, r in pairs])
    IV = np.array([plan.initial_state(nm, mech, N) for nm, _ in pairs])
    dev = N2Device(mech, rows, N)
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 8)
    assert not dev.status().any()
    got = y.cpu().numpy()
    for e in (0, 777, 2047):
        pr = O.setup_n2(members[e], N)
        want = O.rk4(0.0, 8*2e-6, 8, pr["IV"], O.make_rhs_vec(pr), keep=False).reshape(7, N)
        scale = np.max(np.abs(want), axis=1, keepdims=True)
        assert np.max(np.abs(got[e].reshape(7, N) - want)/scale) < 1e-11, e
    assert np.max(np.abs(got[0] - got[1])) > 1e-9
    dev.close()


@pytest.mark.parametrize("name,expect", [("dme_nb", "ros4"), ("syn12", "rk45"), ("ch4", "rk45")])
def test_default_ivp_chooses_the_integrator_like_lsoda(name, expect):
    """The reference's default integrator is LSODA - automatic stiff / non-stiff switching (pbHomoReactor.py:3576).
    An UNMODIFIED modelInput (ivp = "default") probes with the explicit pair and settles on the Rosenbrock stepper
    for the stiff DME case, on Dormand-Prince for the 12-species and CH4 mechanisms (which the stiff stepper
    integrates 30x slower: 13 x 13 node Jacobians for a problem that is not stiff) - and meets the 1e-6 outlet
    requirement against the reference's tight runs either way."""
    g = np.load(os.path.join(G, "g4_tight_%s.npz" % {"dme_nb": "dme_nb_lsoda", "syn12": "syn12_bdf", "ch4": "ch4_bdf"}[name]))
    mi = INP.ALL_N2_INPUTS[name]()
    assert mi["solver-config"]["ivp"] == "default"
    mi["solver-config"]["quiet"] = True
    res = rmtExe(mi)["resModel"]
    st = res["device-stats"]
    assert st["method-per-interval"] == [expect]*5, st["method-per-interval"]
    assert outlet_rel(res["dataPack"], g) < 1e-6
    assert st["rhs_evals"] > 0 and st["steps"] > 0


def test_auto_switches_to_the_stiff_stepper_when_the_bed_heats_up():
    """The DME transient is mild at first (the explicit pair's stability limit is ~1e-5 s on the cold bed) and
    stiff once the bed heats up (~5e-6 s at zNo = 20): with 20 ms output intervals the automatic choice starts
    explicit, finds an interval too expensive (> 2000 steps) and hands over to the stiff stepper for good."""
    mi = INP.dme_notebook_input(period=0.2)
    mi["solver-config"].update({"quiet": True, "tNo": 10})
    res = rmtExe(mi)["resModel"]
    seq = res["device-stats"]["method-per-interval"]
    assert seq[0] == "rk45" and seq[-1] == "ros4", seq
    k = seq.index("ros4")
    assert all(m == "ros4" for m in seq[k:]), seq
    ref = INP.dme_notebook_input(ivp="hip-ros4", period=0.2)
    ref["solver-config"].update({"quiet": True, "tNo": 10, "rtol": 1e-8, "atol": 1e-11})
    a = res["dataPack"][-1]["dataYs"]
    b = rmtExe(ref)["resModel"]["dataPack"][-1]["dataYs"]
    assert np.max(np.abs(a - b)/np.abs(b)) < 2e-6


# ----------------------------------------------------------------------------- stiff stepper, wide mechanism (one node on four lanes)
@pytest.mark.parametrize("N,mode", [(40, "mem"), (150, "mem"), (150, "chain")])
def test_ros4_quad_layout_vs_oracle_controller(N, mode):
    """The 12-species mechanism (13 x 13 node blocks) runs the stiff stepper in its one-node-on-four-lanes layout
    (kernels/61_ros4_quad.inc: matrix rows split over a quad, the iterate through LDS).  Against the oracle's RODAS4
    controller (exact bidiagonal solves): same step history, same end state - for a mesh inside one workgroup, one that
    walks several node blocks with a ragged tail (150 = 64 + 64 + 22), and the same mesh chained over three CUs; two
    reactors with their own step sequences."""
    from rmt_app_amd.n2 import N2Device, ros4_block
    mech = plan.Mechanism(INP.syn12_input())
    assert ros4_block(mech.V, N) == 256 and ros4_block(mech.V, 20) == 128        # 64 / 32 nodes per workgroup
    mis, rows, ivs = [], [], []
    for T in (600.0, 615.0):
        mi = INP.syn12_input()
        mi["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(mi, mech, N)
        mis.append(mi), rows.append(row), ivs.append(plan.initial_state(nm, mech, N))
    dev = N2Device(mech, np.array(rows), N, block=ros4_block(mech.V, N), npt=1, features=("ros4",))
    assert dev.ros_quad and dev.defines["RMT_ROS_QUAD"] == "1"
    dev.set_mode(mode)
    y = dev.to_device(np.array(ivs))
    rtol, atol, h0, t1 = 1e-6, 1e-9, 1e-5, 0.05
    dev.ros4(y, 0.0, t1, rtol, atol, h0, 10**6)
    assert not dev.status().any()
    assert dev.last_geometry()[0] == (1 if mode == "mem" else 3)
    st, got = dev.rk45_stats(), y.cpu().numpy()
    for e, mi in enumerate(mis):
        pr = O.setup_n2(mi, N)
        want, ost = O.ros4(pr, pr["IV"], 0.0, t1, rtol, atol, h0, scheme="rodas4")
        assert st["t_end"][e] == t1
        assert abs(int(st["accepted"][e]) - ost["accepted"]) <= max(3, 0.03*ost["accepted"]), (e, st, ost)
        scale = np.max(np.abs(want.reshape(13, N)), axis=1, keepdims=True)
        assert np.max(np.abs(got[e].reshape(13, N) - want.reshape(13, N))/scale) < 20*rtol, e
    dev.close()
