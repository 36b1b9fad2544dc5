"""GPU: the on-chip RK4 stepper with the temperature-only rate constants cached per node (csrc/kernels/50_rk4.inc
rmt_rk4_reg_body<true>, chosen by n2.kcache_choice for the 512 x 2 geometry).  Stage 1 of a step evaluates the
kinetics in full, stages 2-4 take K(T) = K(T_ref) e^d from the cache; a reactor whose stage temperatures leave the range
that serves is integrated again by the plain stepper (rmt_n2_rk4_reg_redo).  Parity: against the oracle's RK4 (the
reference's arithmetic, PyREMOT/core/pbHomoReactor.py:3706-4066 under solvers/odeSolver... RK4) and against the plain
kernel; the redo path through a shrunken range (RMT_KCACHE_THR)."""
import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import plan
from rmt_app_amd.lowering import FLAG_DOMAIN
from rmt_app_amd.n2 import N2Device

pytestmark = pytest.mark.gpu
N = 1024


def _sweep(E, hot=0.0):
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named, inputs = [], [], []
    for e in range(E):
        mi = INP.dme_notebook_input()
        mi["operating-conditions"]["temperature"] = 513.0 + 4.0*e + hot*(e % 2)
        nm, row = plan.member_constants(mi, mech, N)
        rows.append(row), named.append(nm), inputs.append(mi)
    return mech, np.array(rows), np.array([plan.initial_state(nm, mech, N) for nm in named]), inputs


def _run(mech, rows, IV, steps, dt=2e-6, **kw):
    dev = N2Device(mech, rows, N, block=512, npt=2, **kw)
    y = dev.to_device(IV)
    dev.rk4(y, dt, steps)
    out, flags = y.cpu().numpy(), dev.status().copy()
    info = (dict(dev.defines), dev.lds_state)
    _run.fallbacks = dev.fallbacks()
    dev.close()
    return out, flags, info


def test_cached_stepper_is_the_default_at_512x2_and_matches_plain_and_oracle():
    mech, rows, IV, inputs = _sweep(6)
    got, flags, (defs, lds) = _run(mech, rows, IV, 300)
    assert defs.get("RMT_KCACHE") == "1" and defs.get("RMT_KCACHE_GEN") == "2" and lds == 1      # equilibrium constants too
    assert defs.get("RMT_KC_SMALL_EXP") == "1" and defs.get("RMT_KC_NODE_MAJOR") == "1"
    assert int(defs.get("RMT_KC_REFRESH")) == 8            # the reference point moves every 8th step
    assert not flags.any() and _run.fallbacks == 0
    plain, pflags, (pdefs, plds) = _run(mech, rows, IV, 300, defines={"RMT_KCACHE": "0"})
    assert pdefs["RMT_KCACHE"] == "0" and plds == 0 and not pflags.any()
    scale = np.max(np.abs(plain.reshape(6, 7, N)), axis=2, keepdims=True)
    assert np.max(np.abs(got - plain).reshape(6, 7, N)/scale) < 2e-13
    for e in (0, 5):
        pr = O.setup_n2(inputs[e], N)
        want = O.rk4(0.0, 60*2e-6, 60, pr["IV"], O.make_rhs_vec(pr), keep=False)
        g60, f60, _ = _run(mech, rows[e:e + 1], IV[e:e + 1], 60)
        sc = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
        assert not f60.any()
        assert np.max(np.abs(g60[0].reshape(7, N) - want.reshape(7, N))/sc) < 1e-11, e


@pytest.mark.parametrize("thr", ["1e-12", "3e-7", "2e-5"])
def test_reactors_that_leave_the_cache_range_are_integrated_again_in_full(thr):
    """A shrunken Taylor range (1e-12: every stage of every reactor is out of range; the others: some) - the voided
    reactors come back from the plain stepper of the same code object, bit for bit what a cache-less build of the same
    geometry gives, and no internal flag bit is left behind."""
    mech, rows, IV, _ = _sweep(8, hot=25.0)
    got, flags, (defs, lds) = _run(mech, rows, IV, 150, defines={"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "0",
                                                                   "RMT_KCACHE_THR": thr}, lds_state=1)
    assert defs["RMT_KCACHE_THR"] == thr and lds == 1
    assert not flags.any(), flags
    lo, hi = {"1e-12": (8, 8), "3e-7": (1, 8), "2e-5": (0, 8)}[thr]                      # rmt_n2_fallbacks
    assert lo <= _run.fallbacks <= hi, _run.fallbacks
    plain, pflags, _ = _run(mech, rows, IV, 150, defines={"RMT_KCACHE": "0"}, lds_state=1)
    assert not pflags.any()
    if thr == "1e-12":
        np.testing.assert_array_equal(got, plain)
    else:
        scale = np.max(np.abs(plain.reshape(8, 7, N)), axis=2, keepdims=True)
        assert np.max(np.abs(got - plain).reshape(8, 7, N)/scale) < 2e-13


def test_second_launch_continues_a_redone_reactor_and_a_cached_one_alike():
    """Two launches back to back equal one of twice the length (the state in memory after a voided launch is the plain
    stepper's; the flag word carries nothing over)."""
    mech, rows, IV, _ = _sweep(4, hot=25.0)
    kw = dict(defines={"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "0", "RMT_KCACHE_THR": "3e-7"}, lds_state=1)
    dev = N2Device(mech, rows, N, block=512, npt=2, **kw)
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 80)
    dev.rk4(y, 2e-6, 80)
    assert not dev.status().any()
    two = y.cpu().numpy()
    dev.close()
    one, flags, _ = _run(mech, rows, IV, 160, **kw)
    assert not flags.any()
    scale = np.max(np.abs(one.reshape(4, 7, N)), axis=2, keepdims=True)
    assert np.max(np.abs(two - one).reshape(4, 7, N)/scale) < 2e-13


def test_python_exception_flags_survive_the_cached_stepper():
    """A negative absolute temperature at one node of one reactor (theta = -1.5): log(T) in the equilibrium constants -
    Python's ValueError, RMT_N2_FLAG_DOMAIN on that reactor and on no other (stage 1 is the tested, full evaluation)."""
    mech, rows, IV, _ = _sweep(3)
    IV = IV.copy()
    IV[1].reshape(7, N)[6, 300] = -1.5
    _, flags, (defs, _) = _run(mech, rows, IV, 2)
    assert defs.get("RMT_KCACHE") == "1"
    assert flags[1] & FLAG_DOMAIN and not flags[0] and not flags[2]


# ----------------------------------------------------------------------------- the chained stepper (reactors beyond 1024 nodes)
NC = 2500            # three chunks of 1024 nodes, the last one ragged


def _chain_sweep(E, hot=0.0):
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named, inputs = [], [], []
    for e in range(E):
        mi = INP.dme_notebook_input()
        mi["operating-conditions"]["temperature"] = 513.0 + 30.0*e/max(E - 1, 1) + hot*(e % 2)
        nm, row = plan.member_constants(mi, mech, NC)
        rows.append(row), named.append(nm), inputs.append(mi)
    return mech, np.array(rows), np.array([plan.initial_state(nm, mech, NC) for nm in named]), inputs


def _run_chain(mech, rows, IV, steps, **kw):
    dev = N2Device(mech, rows, NC, block=512, npt=2, **kw)
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, steps)
    out, flags, info = y.cpu().numpy(), dev.status().copy(), (dict(dev.defines), dev.lds_state, dev.last_geometry())
    _run_chain.fallbacks = dev.fallbacks()
    dev.close()
    return out, flags, info


def test_chained_cached_stepper_matches_plain_chain_and_oracle():
    E = 5
    mech, rows, IV, inputs = _chain_sweep(E)
    got, flags, (defs, lds, geo) = _run_chain(mech, rows, IV, 120)
    assert defs.get("RMT_KCACHE_CHAIN") == "1" and lds == 1 and geo[0] == 3
    assert not flags.any()
    plain, pflags, (pdefs, _, _) = _run_chain(mech, rows, IV, 120, defines={"RMT_KCACHE_CHAIN": "0"})
    assert pdefs["RMT_KCACHE_CHAIN"] == "0" and not pflags.any()
    scale = np.max(np.abs(plain.reshape(E, 7, NC)), axis=2, keepdims=True)
    assert np.max(np.abs(got - plain).reshape(E, 7, NC)/scale) < 2e-13
    pr = O.setup_n2(inputs[E - 1], NC)
    want = O.rk4(0.0, 120*2e-6, 120, pr["IV"], O.make_rhs_vec(pr), keep=False)
    sc = np.max(np.abs(want.reshape(7, NC)), axis=1, keepdims=True)
    assert np.max(np.abs(got[E - 1].reshape(7, NC) - want.reshape(7, NC))/sc) < 1e-11


@pytest.mark.parametrize("thr", ["1e-12", "3e-7"])
def test_chained_reactors_that_leave_the_cache_range_are_integrated_again(thr):
    """More reactors than teams, so a team meets voided and clean reactors in one launch; the voided ones are integrated
    again from the saved input by rmt_n2_rk4_chain_redo (1e-12: all of them = the plain chained build bit for bit)."""
    E = 200                                # 85 teams of three chunks on 256 CUs: two or three reactors per team
    mech, rows, IV, _ = _chain_sweep(E, hot=20.0)
    kw = dict(defines={"RMT_KCACHE_CHAIN": "1", "RMT_KCACHE_GEN": "0", "RMT_KCACHE_THR": thr}, lds_state=1)
    got, flags, (_, _, geo) = _run_chain(mech, rows, IV, 40, **kw)
    assert geo[0] == 3 and geo[1] < E
    assert not flags.any(), flags[flags != 0][:4]
    assert _run_chain.fallbacks == E if thr == "1e-12" else 0 < _run_chain.fallbacks <= E
    plain, pflags, _ = _run_chain(mech, rows, IV, 40, defines={"RMT_KCACHE_CHAIN": "0"}, lds_state=1)
    assert not pflags.any()
    if thr == "1e-12":
        np.testing.assert_array_equal(got, plain)
    else:
        scale = np.max(np.abs(plain.reshape(E, 7, NC)), axis=2, keepdims=True)
        assert np.max(np.abs(got - plain).reshape(E, 7, NC)/scale) < 2e-13


def test_chained_cached_stepper_reports_python_exceptions_of_clean_and_redone_reactors():
    """Status bits travel through the side word of the cached kernel (merged by the redo kernel): a negative absolute
    temperature in reactor 1 (clean path) and in reactor 2 of a build whose range voids everything."""
    mech, rows, IV, _ = _chain_sweep(4)
    IV = IV.copy()
    IV[1].reshape(7, NC)[6, 1800] = -1.5
    _, flags, _ = _run_chain(mech, rows, IV, 2)
    assert flags[1] & FLAG_DOMAIN and not flags[0] and not flags[2] and not flags[3]
    _, flags, _ = _run_chain(mech, rows, IV, 2, defines={"RMT_KCACHE_CHAIN": "1", "RMT_KCACHE_GEN": "0",
                                                          "RMT_KCACHE_THR": "1e-12"}, lds_state=1)
    assert flags[1] & FLAG_DOMAIN and not flags[0] and not flags[2] and not flags[3]


def test_a_launch_that_loses_its_reactors_switches_the_next_ones_to_the_plain_stepper():
    """rmt_n2_rk4's host policy for caching code objects: the fallback counter comes back behind every cached launch; after
    a launch that lost at least half of its reactors the next 8 calls run the plain stepper alone (no wasted cached pass),
    then the cached one is tried again.  Shrunken range = every cached launch loses all of them; synchronising between the
    calls makes the sequence deterministic: cached, 8 x plain, cached, 2 x plain."""
    import torch
    E = 6
    mech, rows, IV, _ = _sweep(E)
    dev = N2Device(mech, rows, N, block=512, npt=2, lds_state=1,
                   defines={"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "0", "RMT_KCACHE_THR": "1e-12"})
    y = dev.to_device(IV)
    counts = []
    for _ in range(12):
        dev.rk4(y, 2e-6, 10)
        torch.cuda.synchronize()
        counts.append(dev.fallbacks())
    assert counts == [E]*9 + [2*E]*3, counts
    assert not dev.status().any()
    got = y.cpu().numpy()
    dev.close()
    plain, pflags, _ = _run(mech, rows, IV, 120, defines={"RMT_KCACHE": "0"}, lds_state=1)
    np.testing.assert_array_equal(got, plain)


def test_refresh_period_follows_the_step_size():
    """n2.kc_period = the kernel's rule: at most RMT_KC_REFRESH steps and no reference point older than 13 us of model
    time; a large step (one refresh per step) and the default one give the plain stepper's result."""
    from rmt_app_amd.n2 import kc_period
    d = {"RMT_KC_REFRESH": "8"}
    assert [kc_period(d, dt) for dt in (1e-6, 2e-6, 2.5e-6, 5e-6, 1e-5, 1e-4)] == [8, 6, 5, 2, 1, 1] and kc_period({}, 2e-6) == 1
    mech, rows, IV, _ = _sweep(3)
    for dt, steps in ((1e-6, 200), (4e-6, 60)):
        got, flags, _ = _run(mech, rows, IV, steps, dt=dt)
        plain, pflags, _ = _run(mech, rows, IV, steps, dt=dt, defines={"RMT_KCACHE": "0"})
        assert not flags.any() and not pflags.any() and _run.fallbacks == 0
        scale = np.max(np.abs(plain.reshape(3, 7, N)), axis=2, keepdims=True)
        assert np.max(np.abs(got - plain).reshape(3, 7, N)/scale) < 2e-13


@pytest.mark.parametrize("composition_exp", [True, False])
def test_mechanism_with_every_kind_of_exponent_cached_vs_plain_and_oracle(composition_exp):
    """inputs.ch4_arrhenius_input: an Arrhenius constant written relative to a reference temperature, an exponent that is a
    polynomial in T, 1/T and log T, one that does not decompose, and (optionally) an exp of the composition - the small
    one-workgroup geometry (64 x 1, 20 nodes: the reference's own mesh) with whatever n2.kcache_choice picks, and the
    equilibrium-type constant forced into the cache where the mechanism allows the small exp table."""
    E, n = 5, 20
    rows, named, inputs = [], [], []
    mech = plan.Mechanism(INP.ch4_arrhenius_input(composition_exp=composition_exp))
    for e in range(E):
        mi = INP.ch4_arrhenius_input(composition_exp=composition_exp)
        mi["operating-conditions"]["temperature"] = 940.0 + 15.0*e
        nm, row = plan.member_constants(mi, mech, n)
        rows.append(row), named.append(nm), inputs.append(mi)
    rows, IV = np.array(rows), np.array([plan.initial_state(nm, mech, n) for nm in named])

    def run(**kw):
        dev = N2Device(mech, rows, n, **kw)
        y = dev.to_device(IV)
        dev.rk4(y, 2e-5, 400)
        out = (y.cpu().numpy(), dev.status().copy(), dict(dev.defines), dev.fallbacks())
        dev.close()
        return out
    got, flags, defs, fb = run()
    assert defs.get("RMT_KCACHE") == "1" and defs.get("RMT_KCACHE_GEN") == "0" and not flags.any() and fb == 0
    plain, pflags, pdefs, _ = run(defines={"RMT_KCACHE": "0"})
    assert pdefs["RMT_KCACHE"] == "0" and not pflags.any()
    V = mech.V
    scale = np.max(np.abs(plain.reshape(E, V, n)), axis=2, keepdims=True)
    assert np.max(np.abs(got - plain).reshape(E, V, n)/scale) < 2e-13
    pr = O.setup_n2(inputs[E - 1], n)
    want = O.rk4(0.0, 400*2e-5, 400, pr["IV"], O.make_rhs_vec(pr), keep=False)
    sc = np.max(np.abs(want.reshape(V, n)), axis=1, keepdims=True)
    assert np.max(np.abs(got[E - 1].reshape(V, n) - want.reshape(V, n))/sc) < 1e-11
    # the decomposable exponent in the cache too (two-slot form: this mechanism cannot keep the small exp table)
    two, tflags, tdefs, tfb = run(defines={"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "1", "RMT_KC_REFRESH": "8"})
    assert tdefs["RMT_KCACHE_GEN"] == "1" and not tflags.any() and tfb == 0
    assert np.max(np.abs(two - plain).reshape(E, V, n)/scale) < 2e-13


def test_chained_launch_that_loses_its_reactors_switches_the_next_ones_to_the_plain_stepper():
    """The same host policy for the chained caching stepper: cached, 8 x the plain chained stepper alone (input copied to
    the backup buffer, every redo word forced, not counted as fallbacks), cached again, ..."""
    import torch
    E = 4
    mech, rows, IV, _ = _chain_sweep(E)
    dev = N2Device(mech, rows, NC, block=512, npt=2, lds_state=1,
                   defines={"RMT_KCACHE_CHAIN": "1", "RMT_KCACHE_GEN": "0", "RMT_KCACHE_THR": "1e-12"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    counts = []
    for _ in range(12):
        dev.rk4(y, 2e-6, 5)
        torch.cuda.synchronize()
        counts.append(dev.fallbacks())
    assert counts == [E]*9 + [2*E]*3, counts
    assert not dev.status().any()
    got = y.cpu().numpy()
    dev.close()
    plain, pflags, _ = _run_chain(mech, rows, IV, 60, defines={"RMT_KCACHE_CHAIN": "0"}, lds_state=1)
    assert not pflags.any()
    np.testing.assert_array_equal(got, plain)


def test_model_m2_cached_stepper_matches_plain_and_oracle():
    """The dimensional dynamic model (pbReactor.py:845-1165) through the same caching on-chip stepper: against its plain
    twin and against the M2 oracle's RK4."""
    from oracle import m2_oracle as OM
    E, n = 4, 1024
    mech = plan.Mechanism(INP.m2_dme_input())
    rows, IVs, inputs = [], [], []
    for e in range(E):
        mi = INP.m2_dme_input()
        mi["operating-conditions"]["temperature"] = 510.0 + 8.0*e
        nm, row = plan.member_constants_m2(mi, mech, n)
        rows.append(row), IVs.append(plan.initial_state_m2(nm, mech, n)), inputs.append(mi)
    rows, IV = np.array(rows), np.array(IVs)

    def run(**kw):
        dev = N2Device(mech, rows, n, block=512, npt=2, **kw)
        y = dev.to_device(IV)
        dev.rk4(y, 2e-6, 120)
        out = (y.cpu().numpy(), dev.status().copy(), dict(dev.defines), dev.fallbacks())
        dev.close()
        return out
    got, flags, defs, fb = run()
    assert defs.get("RMT_KCACHE") == "1" and defs.get("RMT_KCACHE_GEN") == "0" and not flags.any() and fb == 0
    plain, pflags, pdefs, _ = run(defines={"RMT_KCACHE": "0"})
    assert pdefs["RMT_KCACHE"] == "0" and not pflags.any()
    V = mech.V
    scale = np.max(np.abs(plain.reshape(E, V, n)), axis=2, keepdims=True)
    assert np.max(np.abs(got - plain).reshape(E, V, n)/scale) < 2e-13
    pr = OM.setup_m2(inputs[E - 1], n)
    want = O.rk4(0.0, 120*2e-6, 120, pr["IV"], OM.make_rhs_vec(pr), keep=False)
    sc = np.max(np.abs(want.reshape(V, n)), axis=1, keepdims=True)
    assert np.max(np.abs(got[E - 1].reshape(V, n) - want.reshape(V, n))/sc) < 1e-10
