"""GPU parity tests: the HIP path (through the C-ABI library) against the reference-generated
golden vectors and against the CPU oracle on the same seeded inputs.  fp64 tolerances:
RHS <= 1e-12 row-relative; fixed-step RK4 trajectories <= 1e-9; outlet mole fractions and
temperature <= 1e-6 relative versus the tight-tolerance SciPy run of the reference RHS."""
import os

import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import plan, rmtExe, solverSetting
from rmt_app_amd.n2 import N2Device

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rowwise_err(a, b, V):
    a = np.asarray(a, float).reshape(V, -1)
    b = np.asarray(b, float).reshape(V, -1)
    den = np.max(np.abs(b), axis=1)
    den[den == 0] = 1.0
    return np.max(np.max(np.abs(a - b), axis=1)/den)


def make_device(name, zNo, E=1, **kw):
    mi = INP.ALL_N2_INPUTS[name]() if isinstance(name, str) else name
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, zNo)
    dev = N2Device(mech, np.tile(row, (E, 1)), zNo, **kw)
    return mi, mech, nm, dev


RHS_CASES = [("dme_nb", 20), ("dme_nb", 100), ("dme_nb", 1024), ("dme_script", 20),
             ("dme_script", 100), ("ch4", 20), ("ch4", 100), ("syn12", 20), ("syn12", 100)]


@pytest.mark.parametrize("name,zNo", RHS_CASES)
def test_rhs_vs_reference_golden(name, zNo):
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["%s_%d_y" % (name, zNo)], g["%s_%d_f" % (name, zNo)]
    _, mech, _, dev = make_device(name, zNo, E=len(Y))
    out = dev.rhs(dev.to_device(Y)).cpu().numpy()
    assert not dev.status().any()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], mech.V) < 1e-12, k
    dev.close()


@pytest.mark.parametrize("block", [64, 128, 256, 1024])
def test_rhs_block_sizes_and_ragged_tail(block):
    """N not a multiple of the workgroup: several node blocks with carry + a ragged tail."""
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["dme_nb_100_y"], g["dme_nb_100_f"]
    _, mech, _, dev = make_device("dme_nb", 100, E=len(Y), block=block, npt=1)
    out = dev.rhs(dev.to_device(Y)).cpu().numpy()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], mech.V) < 1e-12
    dev.close()


def test_rhs_isothermal():
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["dme_nb_iso_20_y"], g["dme_nb_iso_20_f"]
    _, mech, _, dev = make_device(INP.dme_notebook_input(process_type="iso-thermal"), 20, E=len(Y))
    out = dev.rhs(dev.to_device(Y)).cpu().numpy()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], 6) < 1e-12
    dev.close()


@pytest.mark.parametrize("name,zNo", [("dme_nb", 20), ("dme_script", 20), ("dme_nb", 100),
                                      ("ch4", 20), ("syn12", 20)])
@pytest.mark.parametrize("mode", ["reg", "mem"])
def test_rk4_vs_reference_trajectory(name, zNo, mode):
    g = np.load(os.path.join(G, "g3_rk4.npz"))
    key = "%s_%d" % (name, zNo)
    traj, h, n, stride = g[key + "_traj"], float(g[key + "_h"]), int(g[key + "_n"]), int(g[key + "_stride"])
    _, mech, nm, dev = make_device(name, zNo)
    dev.set_mode(mode)
    y = dev.to_device(plan.initial_state(nm, mech, zNo))
    done = 0
    scale = np.maximum(np.max(np.abs(traj), axis=1), 1e-300)
    for col in (1, traj.shape[1]//2, traj.shape[1] - 1):
        steps = col*stride
        dev.rk4(y, h, steps - done)
        done = steps
        got = y.cpu().numpy()[0]
        assert np.max(np.abs(got - traj[:, col])/scale) < 1e-9, col
    assert not dev.status().any()
    dev.close()


@pytest.mark.parametrize("block,npt", [(64, 1), (64, 2), (128, 1), (256, 4), (512, 2), (1024, 1)])
def test_rk4_geometries_agree_with_oracle(block, npt):
    """Every (workgroup, nodes-per-thread) shape of the register stepper, incl. ragged N."""
    N = min(block*npt, 1000) - 3
    mi, mech, nm, dev = make_device("dme_nb", N, block=block, npt=npt)
    y = dev.to_device(plan.initial_state(nm, mech, N))
    dev.rk4(y, 1e-5, 12)
    pr = O.setup_n2(mi, N)
    want = O.rk4(0.0, 12e-5, 12, pr["IV"], O.make_rhs_vec(pr), keep=False)
    got = y.cpu().numpy()[0]
    scale = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(got.reshape(7, N) - want.reshape(7, N))/scale) < 1e-11
    dev.close()


def test_full_size_1024_ensemble_properties():
    """BASELINE config 2/4 sizes: N=1024, E members.  Identical members stay bit-identical,
    the oracle agrees after a short run, and a T/P-perturbed member differs."""
    N, E = 1024, 8
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    rows, named = [], []
    for e in range(E):
        m2 = INP.dme_notebook_input()
        if e == E - 1:
            m2["operating-conditions"]["temperature"] = 533
        nm, row = plan.member_constants(m2, mech, N)
        rows.append(row), named.append(nm)
    dev = N2Device(mech, np.array(rows), N)
    y = dev.to_device(np.array([plan.initial_state(nm, mech, N) for nm in named]))
    dev.rk4(y, 1e-5, 40)
    assert not dev.status().any()
    got = y.cpu().numpy()
    for e in range(1, E - 1):
        np.testing.assert_array_equal(got[e], got[0])
    assert np.max(np.abs(got[E - 1] - got[0])) > 1e-6
    pr = O.setup_n2(mi, N)
    want = O.rk4(0.0, 40e-5, 40, pr["IV"], O.make_rhs_vec(pr), keep=False)
    scale = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(got[0].reshape(7, N) - want.reshape(7, N))/scale) < 1e-10
    # memory-resident stepper gives the same answer at this size
    dev.set_mode("mem")
    y2 = dev.to_device(np.array([plan.initial_state(nm, mech, N) for nm in named]))
    dev.rk4(y2, 1e-5, 40)
    assert np.max(np.abs(y2.cpu().numpy()[0] - got[0])) < 1e-12
    dev.close()


def test_large_mesh_memory_stepper_4096():
    N = 4096
    mi, mech, nm, dev = make_device("dme_nb", N)
    y = dev.to_device(plan.initial_state(nm, mech, N))
    dev.rk4(y, 1e-5, 10)
    pr = O.setup_n2(mi, N)
    want = O.rk4(0.0, 10e-5, 10, pr["IV"], O.make_rhs_vec(pr), keep=False)
    scale = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(y.cpu().numpy()[0].reshape(7, N) - want.reshape(7, N))/scale) < 1e-10
    dev.close()


def test_flags_become_python_exceptions():
    mi, mech, nm, dev = make_device("dme_nb", 20)
    y0 = plan.initial_state(nm, mech, 20).reshape(7, 20).copy()
    y0[6, 3] = -1.5                       # T < 0 -> math.log(T) raises ValueError in the reference
    dev.rhs(dev.to_device(y0.flatten()))
    with pytest.raises(ValueError, match="math domain error"):
        dev.raise_on_flags()
    assert not dev.status().any()         # flags are cleared once read
    y = dev.to_device(plan.initial_state(nm, mech, 20))
    dev.rk4(y, 5e-3, 50)                  # far beyond the stability limit -> blows up
    with pytest.raises((OverflowError, FloatingPointError, ValueError, ZeroDivisionError)):
        dev.raise_on_flags()
    dev.close()


@pytest.mark.parametrize("name", ["dme_script", "dme_nb"])
def test_rmtexe_rk4_end_to_end_vs_tight_scipy_reference(name):
    """BASELINE metric 'max |dMoFri| vs SciPy ref': rmtExe(hip-rk4) against the reference run
    with LSODA rtol=1e-10/atol=1e-12 (golden G4), all five output times.  The requirement is
    <= 1e-6 relative on outlet mole fractions and temperature; measured ~5e-11, asserted 1e-8.
    dt: the DME case needs dt <= ~3e-6 s once the bed is hot (DESIGN.md 'stability')."""
    g = np.load(os.path.join(G, "g4_tight_%s_lsoda.npz" % name))
    mi = INP.ALL_N2_INPUTS[name](ivp="hip-rk4")
    mi["solver-config"].update({"dt": 2.5e-6, "quiet": True})
    res = rmtExe(mi)
    dp = res["resModel"]["dataPack"]
    assert len(dp) == solverSetting["N2"]["tNo"] == 5
    worst = worst_all = 0.0
    for k in range(5):
        a, b = dp[k]["dataYs"], g["dataYs_%d" % k]
        worst = max(worst, np.max(np.abs(a[:, -1] - b[:, -1])/np.abs(b[:, -1])))
        worst_all = max(worst_all, np.max(np.abs(a - b)/np.abs(b)))
        assert abs(dp[k]["dataTime"] - float(g["dataTime_%d" % k])) < 1e-12
        for key in ("dataYCons1", "dataYCons2", "dataYTemp1", "dataYTemp2", "dataXs"):
            assert np.shape(dp[k][key]) == g["%s_%d" % (key, k)].shape
    assert worst < 1e-8, worst
    assert worst_all < 1e-7, worst_all
    assert res["resModel"]["device-stats"]["steps"] == 200000


def test_rmtexe_reports_blowup_like_the_reference_raises():
    """dt above the stability limit: the reference's own explicit path dies with OverflowError
    inside a lambda; here the device flags come back as an exception."""
    mi = INP.dme_script_input(ivp="hip-rk4")
    mi["solver-config"].update({"dt": 1e-3, "quiet": True})
    with pytest.raises((OverflowError, FloatingPointError, ValueError, ZeroDivisionError)):
        rmtExe(mi)


def test_rhs_near_steady_states_backward_stable():
    """Mid-transient / near-steady states taken from the reference's tight LSODA run: every RHS
    row is a small difference of large terms there, so parity is asserted as backward stability
    (tests/parity.py) - the device result lies inside the oracle's own 1e-13 perturbation band
    around the reference value."""
    from parity import backward_ok
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["dme_script_20_transient_y"], g["dme_script_20_transient_f"]
    mi, mech, _, dev = make_device("dme_script", 20, E=len(Y))
    out = dev.rhs(dev.to_device(Y)).cpu().numpy()
    fv = O.make_rhs_vec(O.setup_n2(mi, 20))
    for k in range(len(Y)):
        ok, d = backward_ok(out[k], F[k], fv, Y[k], 7)
        assert ok, (k, d)
    dev.close()


@pytest.mark.parametrize("name,zNo,t1,rtol", [("dme_nb", 20, 4e-3, 1e-6), ("dme_nb", 100, 2e-3, 1e-5),
                                              ("syn12", 20, 2e-2, 1e-6), ("ch4", 20, 1.0, 1e-7)])
def test_rk45_vs_oracle_controller(name, zNo, t1, rtol):
    """Adaptive Dormand-Prince with per-reactor step control against the oracle's restatement of
    the same controller: same accept/reject history (within 2%), end state within 50*rtol."""
    mi, mech, nm, dev = make_device(name, zNo)
    y = dev.to_device(plan.initial_state(nm, mech, zNo))
    atol, h0 = 1e-3*rtol, 1e-6
    dev.rk45(y, 0.0, t1, rtol, atol, h0, 10**7)
    assert not dev.status().any()
    st = dev.rk45_stats()
    pr = O.setup_n2(mi, zNo)
    want, ost = O.rk45(O.make_rhs_vec(pr), 0.0, t1, pr["IV"], rtol, atol, h0)
    assert st["t_end"][0] == t1
    assert abs(int(st["accepted"][0]) - ost["accepted"]) <= max(2, 0.02*ost["accepted"])
    assert abs(int(st["rejected"][0]) - ost["rejected"]) <= max(3, 0.05*ost["accepted"])
    got = y.cpu().numpy()[0]
    V = mech.V
    scale = np.max(np.abs(want.reshape(V, zNo)), axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert np.max(np.abs(got.reshape(V, zNo) - want.reshape(V, zNo))/scale) < 50*rtol
    dev.close()


def test_rk45_per_reactor_step_control():
    """Members with different inlet temperatures take different numbers of steps in one launch."""
    N = 64
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named = [], []
    for T in (503, 523, 563):
        m2 = INP.dme_notebook_input()
        m2["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(m2, mech, N)
        rows.append(row), named.append(nm)
    dev = N2Device(mech, np.array(rows), N)
    y = dev.to_device(np.array([plan.initial_state(nm, mech, N) for nm in named]))
    dev.rk45(y, 0.0, 5e-3, 1e-6, 1e-9, 1e-6, 10**7)
    assert not dev.status().any()
    st = dev.rk45_stats()
    assert np.all(st["t_end"] == 5e-3)
    assert st["accepted"][0] < st["accepted"][1] < st["accepted"][2]      # hotter = stiffer
    dev.close()


def test_rmtexe_unmodified_input_runs_on_device_and_meets_1e6():
    """A completely unmodified modelInput (ivp='default', no extra keys) runs on the device (stiff
    Rosenbrock, default tolerances) and meets the <= 1e-6 outlet requirement against the
    reference's tight LSODA run (golden G4)."""
    g = np.load(os.path.join(G, "g4_tight_dme_nb_lsoda.npz"))
    mi = INP.dme_notebook_input()          # ivp == "default"
    res = rmtExe(mi)
    dp = res["resModel"]["dataPack"]
    worst = max(np.max(np.abs(dp[k]["dataYs"][:, -1] - g["dataYs_%d" % k][:, -1])/np.abs(g["dataYs_%d" % k][:, -1]))
                for k in range(5))
    assert worst < 1e-6, worst


def test_rmtexe_rk45_meets_1e6():
    """ivp='RK45' (SciPy's name) -> device Dormand-Prince with per-reactor step control."""
    g = np.load(os.path.join(G, "g4_tight_dme_nb_lsoda.npz"))
    mi = INP.dme_notebook_input(ivp="RK45")
    mi["solver-config"].update({"quiet": True, "rtol": 1e-8, "atol": 1e-11})
    res = rmtExe(mi)
    dp = res["resModel"]["dataPack"]
    worst = 0.0
    for k in range(5):
        a, b = dp[k]["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
        worst = max(worst, np.max(np.abs(a - b)/np.abs(b)))
    assert worst < 1e-6, worst
    st = res["resModel"]["device-stats"]
    assert st["steps"] > 1000 and st["rejected"] is not None


@pytest.mark.parametrize("name", ["dme_nb", "ch4"])
@pytest.mark.parametrize("meth", ["PreCorr3", "AdBash3"])
def test_multistep_vs_reference_trajectory(name, meth):
    """The reference's `ivp == "AM"` integrators on the device vs its own Python versions (G3b)."""
    g = np.load(os.path.join(G, "g3b_multistep.npz"))
    h, n = float(g[name + "_20_h"]), int(g[name + "_20_n"])
    want = g["%s_20_%s" % (name, meth)]
    scale = np.maximum(np.max(np.abs(want), axis=1), 1e-300)
    for col, steps in enumerate((3, n//2, n)):
        _, mech, nm, dev = make_device(name, 20)
        y = dev.to_device(plan.initial_state(nm, mech, 20))
        dev.multistep(y, h, steps, meth)
        assert not dev.status().any()
        assert np.max(np.abs(y.cpu().numpy()[0] - want[:, col])/scale) < 1e-9, (steps,)
        dev.close()


def test_rmtexe_am_plug_point():
    """ivp='AM' (PreCorr3, n=100 per interval, pbHomoReactor.py:3572,3598-3607): works for the mild
    CH4 case and, like the reference's own Python path on the DME case, raises
    when h = 1e-3 s is far above the stability limit."""
    mi = INP.ch4_input(ivp="AM", period=2.0)
    mi["solver-config"]["quiet"] = True
    res = rmtExe(mi)
    pr = O.setup_n2(INP.ch4_input(), 20)
    f = O.make_rhs_vec(pr)
    yv = pr["IV"]
    for k in range(5):
        yv = O.precorr3(0.4*k, 0.4*(k + 1), 100, yv, f)[:, -1]
        got = res["resModel"]["dataPack"][k]
        want = O.pack_interval(yv, pr, 0.4*(k + 1))
        np.testing.assert_allclose(got["dataYs"], want["dataYs"], rtol=1e-10)
    mi = INP.dme_script_input(ivp="AM")
    mi["solver-config"]["quiet"] = True
    with pytest.raises((OverflowError, FloatingPointError, ValueError, ZeroDivisionError)):
        rmtExe(mi)


@pytest.mark.parametrize("N,E,block,npt", [(300, 3, 64, 1), (300, 70, 64, 1), (1000, 2, 128, 2),
                                           (4096, 2, None, None), (5000, 1, None, None), (1024, 64, None, None)])
def test_chained_workgroups_agree_with_oracle(N, E, block, npt):
    """One reactor spread over several workgroups (producer->consumer chain of boundary records):
    ragged chunking, more reactors than teams, the default geometry at N = 4096, and a mid-size ensemble
    (64 x 1024 nodes) that the default geometry cuts into 4 chunks per reactor to fill the CUs."""
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    rows, named = [], []
    for e in range(E):
        m2 = INP.dme_notebook_input()
        m2["operating-conditions"]["temperature"] = 523 + (e % 7)
        nm, row = plan.member_constants(m2, mech, N)
        rows.append(row), named.append(nm)
    dev = N2Device(mech, np.array(rows), N, block=block, npt=npt)
    dev.set_mode("chain")
    IV = np.array([plan.initial_state(nm, mech, N) for nm in named])
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 9)
    assert not dev.status().any()
    got = y.cpu().numpy()
    for e in sorted({0, E - 1, min(E - 1, 7)}):
        m2 = INP.dme_notebook_input()
        m2["operating-conditions"]["temperature"] = 523 + (e % 7)
        pr = O.setup_n2(m2, N)
        want = O.rk4(0.0, 9*2e-6, 9, pr["IV"], O.make_rhs_vec(pr), keep=False)
        scale = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
        assert np.max(np.abs(got[e].reshape(7, N) - want.reshape(7, N))/scale) < 1e-11, e
    # and bit-identical to the memory-resident stepper's arithmetic up to rounding
    dev.set_mode("mem")
    y2 = dev.to_device(IV)
    dev.rk4(y2, 2e-6, 9)
    assert np.max(np.abs(y2.cpu().numpy() - got)) < 1e-12
    dev.close()


@pytest.mark.parametrize("scheme,defines", [("rodas4", {}), ("kr4", {"RMT_ROS_SCHEME": 0})])
def test_ros4_vs_oracle_controller(scheme, defines):
    """Stiff Rosenbrock stepper with per-reactor step control vs the oracle's restatement (exact
    bidiagonal solves there, Jacobi sweeps here): same step history, same end state - for the
    default RODAS4 and for the Kaps-Rentrop pair (RMT_ROS_SCHEME 0)."""
    N = 20
    mi, mech, nm, dev = make_device("dme_script", N, block=64, npt=1, features=("ros4",), defines=defines)
    y = dev.to_device(plan.initial_state(nm, mech, N))
    rtol, atol, h0, t1 = 1e-6, 1e-9, 1e-5, 0.1
    dev.ros4(y, 0.0, t1, rtol, atol, h0, 10**6)
    assert not dev.status().any()
    st = dev.rk45_stats()
    pr = O.setup_n2(mi, N)
    want, ost = O.ros4(pr, pr["IV"], 0.0, t1, rtol, atol, h0, scheme=scheme)
    assert st["t_end"][0] == t1
    assert abs(int(st["accepted"][0]) - ost["accepted"]) <= max(3, 0.03*ost["accepted"])
    got = y.cpu().numpy()[0]
    scale = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(got.reshape(7, N) - want.reshape(7, N))/scale) < 20*rtol
    dev.close()


@pytest.mark.parametrize("name", ["dme_script", "dme_nb"])
def test_rmtexe_ros4_end_to_end_vs_tight_scipy_reference(name):
    """rmtExe(ivp='hip-ros4'): a few hundred implicit steps instead of 200 000 explicit ones,
    outlet mole fractions and temperature <= 1e-6 vs the reference under LSODA rtol 1e-10."""
    g = np.load(os.path.join(G, "g4_tight_%s_lsoda.npz" % name))
    mi = INP.ALL_N2_INPUTS[name](ivp="hip-ros4")
    mi["solver-config"].update({"quiet": True})          # default tolerances (1e-6 / 1e-9)
    res = rmtExe(mi)
    dp = res["resModel"]["dataPack"]
    worst = 0.0
    for k in range(5):
        a, b = dp[k]["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
        worst = max(worst, np.max(np.abs(a - b)/np.abs(b)))
    assert worst < 2e-7, worst                           # requirement: 1e-6
    st = res["resModel"]["device-stats"]
    assert 100 < st["steps"] < 3000, st


def test_ros4_1024_nodes_ensemble_matches_explicit():
    """N = 1024 (4 node blocks per workgroup, Jacobi sweeps across them) on a small sweep:
    the stiff stepper agrees with RK4 at dt = 2e-6 after 4 ms."""
    N, E = 1024, 4
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named = [], []
    for T in (513, 523, 533, 543):
        m2 = INP.dme_notebook_input()
        m2["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(m2, mech, N)
        rows.append(row), named.append(nm)
    IV = np.array([plan.initial_state(nm, mech, N) for nm in named])
    dev = N2Device(mech, np.array(rows), N, block=256, npt=1, features=("ros4",))
    y = dev.to_device(IV)
    dev.ros4(y, 0.0, 4e-3, 1e-7, 1e-10, 1e-6, 10**6)
    assert not dev.status().any()
    st = dev.rk45_stats()
    ref = dev.to_device(IV)
    dev.rk4(ref, 2e-6, 2000)
    a, b = y.cpu().numpy().reshape(E, 7, N), ref.cpu().numpy().reshape(E, 7, N)
    scale = np.max(np.abs(b), axis=2, keepdims=True)
    assert np.max(np.abs(a - b)/scale) < 2e-6
    assert np.all(st["accepted"] < 400)
    dev.close()


def test_rmtexe_ensemble_sweep_api():
    """solver-config.ensemble = {"temperature": [...], "pressure": [...]}: one launch per output
    interval for the whole sweep; every member equals its own single-reactor rmtExe run."""
    base = INP.dme_notebook_input(ivp="hip-rk4", period=0.004)
    base["solver-config"].update({"quiet": True, "dt": 2e-6, "zNo": 64, "tNo": 2,
                                  "ensemble": {"temperature": [513.0, 533.0], "pressure": [4.0e6, 5.0e6, 6.0e6]}})
    res = rmtExe(base)
    ens = res["resModel"]["ensemble"]
    assert len(ens) == 6 and len(ens[0]["dataPack"]) == 2
    from rmt_app_amd.ensemble import expand_members
    members = expand_members(base, base["solver-config"]["ensemble"])
    for e in (0, 4):
        single = dict(members[e])
        single["solver-config"] = {k: v for k, v in base["solver-config"].items() if k != "ensemble"}
        one = rmtExe(single)["resModel"]["dataPack"]
        for k in range(2):
            np.testing.assert_allclose(ens[e]["dataPack"][k]["dataYs"], one[k]["dataYs"], rtol=1e-12)
    # the members really differ
    assert abs(ens[0]["dataPack"][1]["dataYs"][6, -1] - ens[5]["dataPack"][1]["dataYs"][6, -1]) > 1.0


def test_rmtexe_n1_steady_state_profile():
    """BASELINE configs[0] (README steady-state example, model N1) on the device: the profile along
    z* against the oracle's modelEquationN1 under LSODA rtol 1e-11 (<= 1e-6 everywhere) and against
    the reference's own default-tolerance run (golden G6; that run is only ~1e-4 accurate)."""
    from scipy.integrate import solve_ivp
    g = np.load(os.path.join(G, "g6_n1.npz"))
    mi = INP.n1_notebook_input()
    res = rmtExe(mi)
    d = res["resModel"][0]
    assert d["dataYs"].shape == g["dataYs"].shape == (8, 101)
    assert d["labelList"] == INP.DME_COMPONENTS + ["Pressure", "Temperature"] and d["indexList"] == [6, 6, 7]
    pr = O.setup_n1(mi)
    tight = solve_ivp(lambda t, y: O.n1_rhs(t, y, pr), [0, 1], pr["IV"], method="LSODA", rtol=1e-11,
                      atol=1e-13, t_eval=np.linspace(0, 1, 101))
    S = 6
    conc = tight.y[:S]*np.max(pr["SpCoi0"])
    want = np.concatenate([conc/conc.sum(0), (tight.y[S]*pr["Pf"]).reshape(1, -1),
                           (tight.y[S + 1]*pr["Tf"] + pr["Tf"]).reshape(1, -1)])
    assert np.max(np.abs(d["dataYs"] - want)/np.abs(want)) < 1e-6
    assert np.max(np.abs(d["dataYs"] - g["dataYs"])/np.abs(g["dataYs"])) < 5e-3
    for key in ("dataYCons1", "dataYCons2", "dataYTemp1", "dataYTemp2", "dataXs"):
        assert np.shape(d[key]) == g[key].shape, key
    # BASELINE.md outlet of the reference run: T = 620.857 K, P = 4.99266 MPa
    assert abs(d["dataYs"][7, -1] - 620.857) < 0.05 and abs(d["dataYs"][6, -1] - 4992663.0) < 50.0


def test_n1_ensemble_one_reactor_per_lane():
    mi = INP.n1_notebook_input()
    mi["solver-config"]["ensemble"] = {"temperature": list(np.linspace(503.0, 543.0, 70)), "pressure": [5.0e6]}
    packs = rmtExe(mi)["resModel"]
    assert len(packs) == 70
    Tout = np.array([p["dataYs"][7, -1] for p in packs])
    assert np.all(np.isfinite(Tout)) and np.all(np.diff(Tout) > 0)       # hotter feed -> hotter outlet
    one = INP.n1_notebook_input()
    one["operating-conditions"]["temperature"] = float(np.linspace(503.0, 543.0, 70)[33])
    c0 = np.array(INP.n1_notebook_input()["feed"]["concentration"])
    one["feed"]["concentration"] = (c0/c0.sum())*5.0e6/(INP.R_CONST*one["operating-conditions"]["temperature"])
    single = rmtExe(one)["resModel"][0]
    np.testing.assert_allclose(packs[33]["dataYs"], single["dataYs"], rtol=1e-9)


def test_set_members_switches_operating_point_without_recompiling():
    N = 128
    mech = plan.Mechanism(INP.dme_notebook_input())
    def rows_for(T):
        m2 = INP.dme_notebook_input()
        m2["operating-conditions"]["temperature"] = T
        return plan.member_constants(m2, mech, N)
    (nmA, rowA), (nmB, rowB) = rows_for(523), rows_for(540)
    dev = N2Device(mech, np.array([rowA, rowA]), N, specialize=False)
    dev.set_members(np.array([rowB, rowA]))
    y = dev.to_device(np.array([plan.initial_state(nmB, mech, N), plan.initial_state(nmA, mech, N)]))
    dev.rk4(y, 2e-6, 50)
    fresh = N2Device(mech, np.array([rowB, rowA]), N, specialize=False)
    y2 = fresh.to_device(np.array([plan.initial_state(nmB, mech, N), plan.initial_state(nmA, mech, N)]))
    fresh.rk4(y2, 2e-6, 50)
    np.testing.assert_array_equal(y.cpu().numpy(), y2.cpu().numpy())
    assert np.max(np.abs(y.cpu().numpy()[0] - y.cpu().numpy()[1])) > 1e-6
    spec = N2Device(mech, np.array([rowA, rowA]), N)          # specialised: refuses
    from rmt_app_amd.hipbind import RmtN2Error
    with pytest.raises(RmtN2Error):
        spec.set_members(np.array([rowB, rowA]))
    for d in (dev, fresh, spec):
        d.close()


@pytest.mark.parametrize("name,N,t1,dt", [("ch4", 100, 2.0, 1e-3), ("syn12", 64, 0.02, 2e-6)])
def test_ros4_other_mechanisms_match_explicit(name, N, t1, dt):
    """The stiff stepper is mechanism-generic: adiabatic 3-species CH4 (Tm == 0) and the
    12-species / 8-reaction mechanism (13x13 node Jacobians) against RK4 at a small dt."""
    mi, mech, nm, dev = make_device(name, N, block=64 if N <= 64 else 128, npt=1, features=("ros4",))
    IV = plan.initial_state(nm, mech, N)
    y = dev.to_device(IV)
    dev.ros4(y, 0.0, t1, 1e-7, 1e-10, 1e-6, 10**6)
    assert not dev.status().any()
    st = dev.rk45_stats()
    ref = dev.to_device(IV)
    n = int(round(t1/dt))
    dev.rk4(ref, t1/n, n)
    assert not dev.status().any()
    V = mech.V
    a, b = y.cpu().numpy().reshape(V, N), ref.cpu().numpy().reshape(V, N)
    scale = np.max(np.abs(b), axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert np.max(np.abs(a - b)/scale) < 5e-6
    assert int(st["accepted"][0]) < n//5
    dev.close()


@pytest.mark.parametrize("ivp,extra,tol", [("hip-rk4", {"dt": 2e-6}, 1e-9), ("hip-ros4", {}, 1e-6),
                                           ("hip-rk45", {"rtol": 1e-8, "atol": 1e-11}, 1e-6)])
def test_rmtexe_isothermal_variant(ivp, extra, tol):
    """process-type 'iso-thermal' (V = S, T == Tf; pbHomoReactor.py:3469-3470, 3883-3884, 4102)
    end to end on every integrator against the oracle's RK4 of the same model."""
    mi = INP.dme_notebook_input(ivp=ivp, process_type="iso-thermal", period=0.01)
    mi["solver-config"].update(dict(extra, quiet=True, tNo=2))
    dp = rmtExe(mi)["resModel"]["dataPack"]
    pr = O.setup_n2(mi, 20)
    f = O.make_rhs_vec(pr)
    yv = pr["IV"]
    for k in range(2):
        yv = O.rk4(0.005*k, 0.005*(k + 1), 2500, yv, f, keep=False)
        want = O.pack_interval(yv, pr, 0.005*(k + 1))
        assert dp[k]["dataYs"].shape == want["dataYs"].shape == (7, 20)
        assert np.max(np.abs(dp[k]["dataYs"] - want["dataYs"])/np.abs(want["dataYs"])) < tol
        np.testing.assert_array_equal(dp[k]["dataYs"][6], 523.0)
        # dataYCons1 = dataYs_Reshaped[:-1] also when V = S (pbHomoReactor.py:3636): 5 rows, like the reference
        assert np.shape(dp[k]["dataYCons1"]) == (5, 20) and np.shape(dp[k]["dataYTemp1"]) == (20,)
        assert np.shape(want["dataYCons1"]) == (5, 20)


def test_rmtexe_fp32_dtype_runs_and_is_single_precision_accurate():
    mi = INP.dme_notebook_input(ivp="hip-rk4", period=0.004)
    mi["solver-config"].update({"dt": 2e-6, "quiet": True, "tNo": 1, "dtype": "fp32", "zNo": 64})
    a = rmtExe(mi)["resModel"]["dataPack"][0]["dataYs"]
    mi["solver-config"]["dtype"] = "fp64"
    b = rmtExe(mi)["resModel"]["dataPack"][0]["dataYs"]
    err = np.max(np.abs(a - b)/np.abs(b))
    assert 1e-9 < err < 2e-5, err


def test_rmtexe_fp32_stiff_stepper_is_single_precision_accurate():
    """dtype fp32 with the stiff stepper: runs to the end and lands within single-precision
    distance of the tight reference run (it cannot carry the 1e-6 requirement)"""
    g = np.load(os.path.join(G, "g4_tight_dme_script_lsoda.npz"))
    mi = INP.dme_script_input(ivp="hip-ros4")
    mi["solver-config"].update({"quiet": True, "dtype": "fp32", "rtol": 1e-4, "atol": 1e-7})
    dp = rmtExe(mi)["resModel"]["dataPack"]
    for k in range(5):
        a, b = dp[k]["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
        assert np.max(np.abs(a - b)/np.abs(b)) < 5e-5, k


def test_full_size_16384_nodes_chain_vs_memory_stepper_and_properties():
    """BASELINE configs[2] size (16384 nodes): the chained on-chip stepper and the memory-resident
    one are independent code paths for the cross-block carries - they must agree; plus
    size-independent properties: mole fractions sum to one, pressure falls monotonically,
    and halving dt changes the answer by O(dt^4)."""
    N = 16384
    mi, mech, nm, dev = make_device("dme_nb", N)
    IV = plan.initial_state(nm, mech, N)
    outs = {}
    for mode, dt, n in (("chain", 2e-6, 40), ("mem", 2e-6, 40), ("chain", 1e-6, 80), ("chain", 4e-6, 20)):
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.rk4(y, dt, n)
        assert not dev.status().any()
        outs[(mode, dt)] = y.cpu().numpy()[0].reshape(7, N)
    a, b = outs[("chain", 2e-6)], outs[("mem", 2e-6)]
    assert np.max(np.abs(a - b)) < 1e-12
    conc = a[:6]*nm["Cmax"]
    x = conc/conc.sum(0)
    assert np.max(np.abs(x.sum(0) - 1.0)) < 1e-14
    e1 = np.max(np.abs(outs[("chain", 4e-6)] - outs[("chain", 2e-6)]))
    e2 = np.max(np.abs(outs[("chain", 2e-6)] - outs[("chain", 1e-6)]))
    assert e2 < e1/8 or e1 < 1e-13          # 4th order: ratio ~16
    # the RHS's pressure profile (recomputed by the oracle from the device state) is monotone
    pr = O.setup_n2(mi, N)
    _, _, _, _, P = O.neighbourhood(pr, a.reshape(1, 7, N))
    assert np.all(np.diff(P[0]) < 0) and P[0, 0] == 5.0e6
    dev.close()


def test_adaptive_steppers_report_step_budget_exhaustion():
    """max-steps too small: RMT_FLAG_STEP -> RuntimeError (the reference does a bare `raise` when
    solve_ivp reports failure, pbHomoReactor.py:3614-3615)."""
    for ivp in ("hip-rk45", "hip-ros4"):
        mi = INP.dme_notebook_input(ivp=ivp)
        mi["solver-config"].update({"quiet": True, "max-steps": 5})
        with pytest.raises(RuntimeError, match="step"):
            rmtExe(mi)


# ----------------------------------------------------------------------------- benchmark mesh (G8)
def _g8():
    p = os.path.join(G, "g8_mesh1024_dme_nb_dop853.npz")
    if not os.path.exists(p):
        pytest.skip("golden G8 not generated")
    return np.load(p)


@pytest.mark.parametrize("ivp,tol", [("hip-rk4", 1e-7), ("hip-ros4", 3e-6)])
def test_benchmark_mesh_1024_vs_scipy_on_oracle_rhs(ivp, tol):
    """at the benchmark mesh (zNo = 1024) the reference's own RHS is
    infeasible, so the trajectory is pinned by SciPy DOP853 (rtol 1e-10) driving the oracle's
    vectorised RHS (tools/make_mesh_golden.py, golden G8; that RHS is pinned <= 1e-12 against the
    reference at this N by G2).  Whole profiles at every output time, default tolerances:
    BASELINE's metric max|dMoFri| (absolute) and |dT|/T <= 1e-6; relative to each value (trace
    species and the steep front included) <= `tol`; outlet values <= 1e-6 relative."""
    g = _g8()
    done = int(g["done"])
    mi = INP.dme_notebook_input(ivp=ivp)
    mi["solver-config"].update({"zNo": 1024, "tNo": 5, "quiet": True, "display-result": "False"})
    dp = rmtExe(mi)["resModel"]["dataPack"]
    pr = O.setup_n2(INP.dme_notebook_input(), 1024)
    for k in range(done):
        ref = O.pack_interval(g["states"][k], pr, float(g["times"][k]))["dataYs"]
        got = dp[k]["dataYs"]
        assert np.max(np.abs(got[:6] - ref[:6])) < 1e-6, k                     # max |dMoFri|
        assert np.max(np.abs(got[6] - ref[6])/ref[6]) < 1e-6, k               # |dT|/T
        rel = np.abs(got - ref)/np.maximum(np.abs(ref), 1e-30)
        assert np.max(rel) < tol, (k, np.max(rel))
        assert np.max(rel[:, -1]) < 1e-6, (k, np.max(rel[:, -1]))


@pytest.mark.parametrize("N", [2, 3, 63, 65])
def test_tiny_and_wave_boundary_meshes_vs_oracle(N):
    """smallest meshes the reference accepts (zNo = 2: inlet-fed node + outlet) and sizes straddling
    one wave: RHS and 10 RK4 steps against the oracle's vectorised restatement"""
    mi, mech, nm, dev = make_device("dme_script", N)
    pr = O.setup_n2(mi, N)
    f = O.make_rhs_vec(pr)
    y = dev.to_device(pr["IV"])
    assert rowwise_err(dev.rhs(y).cpu().numpy()[0], f(0.0, pr["IV"]), mech.V) < 1e-12
    dev.rk4(y, 1e-6, 10)
    want = O.rk4(0.0, 1e-5, 10, pr["IV"], f, keep=False)
    assert rowwise_err(y.cpu().numpy()[0], want, mech.V) < 1e-12
    assert not dev.status().any()
    dev.close()


def test_kcache_opt_in_matches_the_shipped_kernel():
    """defines={"RMT_KCACHE": "1"} (opt-in: the temperature-only rate constants cached per node, a Taylor step in
    d = f(T) - f(T_ref) while |d| <= 2^-9, full evaluation + new reference point otherwise): same trajectory as the
    shipped on-chip RK4 kernel to rounding, over steps small enough to stay on the short path and over steps that
    leave the Taylor range every stage (the refresh path), and the oracle's RK4."""
    N = 200
    mi, mech, nm, dev = make_device("dme_nb", N, E=3, block=128, npt=2, lds_state=0)
    _, _, _, devc = make_device("dme_nb", N, E=3, block=128, npt=2, lds_state=0, defines={"RMT_KCACHE": "1"})
    IV = np.tile(plan.initial_state(nm, mech, N), (3, 1))
    for dt, n in ((2e-6, 400), (2.5e-5, 40)):
        y, yc = dev.to_device(IV), devc.to_device(IV)
        dev.rk4(y, dt, n)
        devc.rk4(yc, dt, n)
        assert not dev.status().any() and not devc.status().any()
        a, b = y.cpu().numpy(), yc.cpu().numpy()
        for e in range(3):
            assert rowwise_err(b[e], a[e], mech.V) < 1e-12, (dt, e)
    pr = O.setup_n2(mi, N)
    want = O.rk4(0.0, 40*2.5e-5, 40, pr["IV"], O.make_rhs_vec(pr), keep=False)
    assert rowwise_err(b[0], want, mech.V) < 1e-11
    dev.close()
    devc.close()
