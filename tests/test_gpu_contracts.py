"""GPU tests of the failure contracts of the C ABI (include/rmt_n2.h): the chained stepper's
time-out / poison path, the per-stage Python-exception flags, the resume form of the adaptive
steppers (h0 < 0) and the device binding of a handle."""
import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import plan
from rmt_app_amd.lowering import FLAG_NONFINITE, FLAG_OVERFLOW, FLAG_STEP
from rmt_app_amd.n2 import N2Device

pytestmark = pytest.mark.gpu


def _sweep(N, E):
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named = [], []
    for e in range(E):
        mi = INP.dme_notebook_input()
        mi["operating-conditions"]["temperature"] = 523 + (e % 5)
        nm, row = plan.member_constants(mi, mech, N)
        rows.append(row), named.append(nm)
    return mech, np.array(rows), np.array([plan.initial_state(nm, mech, N) for nm in named])


def test_chain_stuck_producer_ends_with_step_flag_on_every_member():
    """One chunk of every chain stops publishing its boundary records (debug define): the consumers
    time out, poison the chain in both directions, every workgroup drains and the launch ENDS; every
    reactor of an affected team - the one in flight and the ones never started - carries
    RMT_FLAG_STEP, so no member reads as successfully advanced."""
    N, E = 1000, 300                     # 4 chunks of 256 nodes; 64 teams on 256 CUs -> 4-5 reactors per team
    mech, rows, IV = _sweep(N, E)
    dev = N2Device(mech, rows, N, block=128, npt=2, specialize=False,
                   defines={"RMT_CHAIN_SPINS": "4096", "RMT_CHAIN_TEST_STALL_CHUNK": "1",
                            "RMT_CHAIN_TEST_STALL_FROM": "6"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 50)
    flags = dev.status()                 # returns: the launch did not hang
    assert np.all(flags & FLAG_STEP), flags[:8]
    with pytest.raises(RuntimeError, match="step"):
        dev.rk4(y, 2e-6, 50)
        dev.raise_on_flags()
    dev.close()
    # the same geometry without the stall is healthy (the sync words are reset per launch)
    dev = N2Device(mech, rows, N, block=128, npt=2, specialize=False, defines={"RMT_CHAIN_SPINS": "4194304"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 5)
    assert not dev.status().any()
    dev.set_mode("mem")
    y2 = dev.to_device(IV)
    dev.rk4(y2, 2e-6, 5)
    assert np.max(np.abs(y.cpu().numpy() - y2.cpu().numpy())) < 1e-12
    dev.close()


def _overflow_in_trial_stage_input(N=20, dt=1e-3):
    """CH4 case with a rate law whose exp() overflows as soon as the methane mole fraction has fallen by
    what half an RK4 step removes: fine at y_0, OverflowError in Python at every trial stage and later
    state (in IEEE arithmetic 1/(1+inf) = 0 would keep the rate itself finite).  The steepness K is chosen from
    the oracle's K_1 so that the exponent is ~1000 at stage 2."""
    import math

    def build(K):
        mi = INP.ch4_input()
        c = np.asarray(mi["feed"]["concentration"], dtype=float)
        mi["reaction-rates"]["VARS"]["y_feed"] = float(c[0]/c.sum())
        mi["reaction-rates"]["RATES"] = {
            "r1": (lambda K: lambda x: x['k0']*(x['C_CH4']**2)*(1.0 + 1.0/(1.0 + math.exp(K*(x['y_feed'] - x['y_CH4'])))))(K)}
        return mi
    pr = O.setup_n2(build(0.0), N)
    y2 = (pr["IV"] + 0.5*dt*O.make_rhs_vec(pr)(0.0, pr["IV"])).reshape(4, N)
    iv = pr["IV"].reshape(4, N)
    drop = float(np.max(iv[0, 0]/iv[:3, 0].sum() - y2[0]/y2[:3].sum(0)))   # fall of y_CH4 over half a step
    return build(1000.0/drop)


def test_exception_flags_first_stage_by_default_every_stage_when_strict():
    """include/rmt_n2.h contract: the Python-exception bits are tested on the first stage of a step;
    RMT_CHECK_ALL_STAGES (solver-config 'strict-flags') tests all of them."""
    N = 20
    mi = _overflow_in_trial_stage_input(N)
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    IV = plan.initial_state(nm, mech, N)
    # the reference's path raises OverflowError at the stage-2 state of the first step
    pr = O.setup_n2(mi, N)
    k1 = O.make_rhs_vec(pr)(0.0, pr["IV"])
    O.rhs_loop(0.0, pr["IV"], pr)
    with pytest.raises(OverflowError):
        O.rhs_loop(0.0, pr["IV"] + 0.5e-3*k1, pr)
    for mode in ("reg", "mem"):
        dev = N2Device(mech, row, N)
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.rk4(y, 1e-3, 1)              # stage 1 at y_0 is clean; stages 2-4 overflow inside the lambda
        f = int(dev.status()[0])
        # default contract: not reported as the Python exception (OVERFLOW) - but not lost either: the lean
        # fp64 division turns 1/(1+inf) into NaN (Newton step on rcp(inf) = 0), so the state is poisoned and
        # the launch ends with NONFINITE (FloatingPointError on the host)
        assert not (f & FLAG_OVERFLOW) and (f & FLAG_NONFINITE), (mode, f)
        assert not np.all(np.isfinite(y.cpu().numpy()))
        dev.close()
        strict = N2Device(mech, row, N, defines={"RMT_CHECK_ALL_STAGES": "1"})
        strict.set_mode(mode)
        y = strict.to_device(IV)
        strict.rk4(y, 1e-3, 1)
        assert strict.status()[0] & FLAG_OVERFLOW, mode
        strict.close()
    from rmt_app_amd import rmtExe
    mi["solver-config"].update({"ivp": "hip-rk4", "dt": 1e-3, "quiet": True, "strict-flags": True})
    with pytest.raises(OverflowError):
        rmtExe(mi)


@pytest.mark.parametrize("stepper", ["rk45", "ros4"])
def test_adaptive_resume_keeps_per_reactor_step_and_unclipped_h_last(stepper):
    """h0 < 0 resumes every reactor from its own h_last; a last step clipped to the interval end does not
    shrink h_last; two half intervals give the same end state as one whole interval to within the
    tolerance and do not pay a restart transient (accepted steps within +3 of the single launch)."""
    N = 64
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named = [], []
    for T in (503, 543):
        mi = INP.dme_notebook_input()
        mi["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(mi, mech, N)
        rows.append(row), named.append(nm)
    IV = np.array([plan.initial_state(nm, mech, N) for nm in named])
    kw = dict(block=64, npt=1, features=("ros4",)) if stepper == "ros4" else {}
    dev = N2Device(mech, np.array(rows), N, **kw)
    run = getattr(dev, stepper)
    rtol, atol, h0, t1 = 1e-6, 1e-9, 1e-6, (2e-2 if stepper == "ros4" else 2e-3)
    y = dev.to_device(IV)
    run(y, 0.0, t1, rtol, atol, h0, 10**7)
    one = dev.rk45_stats()
    whole = y.cpu().numpy()
    y = dev.to_device(IV)
    # an interval end that certainly clips the last step
    tm = 0.5*t1*(1 + 1e-3)
    run(y, 0.0, tm, rtol, atol, h0, 10**7)
    a = dev.rk45_stats()
    run(y, tm, t1, rtol, atol, -h0, 10**7)
    b = dev.rk45_stats()
    assert not dev.status().any()
    assert np.all(b["t_end"] == t1)
    # h_last of the first half is a full-size proposal, not the clipped remainder
    assert np.all(a["h_last"] > 0.2*one["h_last"].min())
    assert a["h_last"][0] != a["h_last"][1]                     # per-reactor values
    acc2 = a["accepted"] + b["accepted"]
    assert np.all(acc2 <= one["accepted"] + 3), (acc2, one["accepted"])
    two = y.cpu().numpy()
    scale = np.max(np.abs(whole.reshape(2, 7, N)), axis=2, keepdims=True)
    assert np.max(np.abs(two.reshape(2, 7, N) - whole.reshape(2, 7, N))/scale) < 100*rtol
    dev.close()


def test_handle_is_bound_to_its_device_and_restores_the_callers():
    """rmt_n2.h: entry points run on the handle's device whatever the current one is.  With one GPU
    visible this can only check that the current device is untouched by the calls."""
    import torch
    mech, rows, IV = _sweep(64, 2)
    dev = N2Device(mech, rows, 64)
    before = torch.cuda.current_device()
    y = dev.to_device(IV)
    dev.rk4(y, 2e-6, 3)
    dev.rhs(y)
    assert not dev.status().any()
    assert torch.cuda.current_device() == before
    if torch.cuda.device_count() > 1:
        with torch.cuda.device(1):
            dev.rk4(y, 2e-6, 3)          # launched on device 0's stream and module regardless
            assert not dev.status().any()
            assert torch.cuda.current_device() == 1
    dev.close()
