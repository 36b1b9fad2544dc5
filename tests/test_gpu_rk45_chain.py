"""GPU tests of rmt_n2_rk45_chain: the adaptive on-chip stepper with one reactor cut into chunks on several
CUs (tagged-word links per RK stage, error norm down the chain, decision slot).  Checked against the oracle's
controller (oracle/n2_oracle.py rk45) at sizes it finishes in seconds, against the memory-resident kernel
of the same library at the sizes the chain is for, and for its failure contract (a stuck producer)."""
import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import plan
from rmt_app_amd.lowering import FLAG_STEP
from rmt_app_amd.n2 import N2Device, rk45_geometry

pytestmark = pytest.mark.gpu


def _members(name, N, temps):
    mech = plan.Mechanism(INP.ALL_N2_INPUTS[name]())
    rows, ivs, mis = [], [], []
    for T in temps:
        mi = INP.ALL_N2_INPUTS[name]()
        mi["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(mi, mech, N)
        rows.append(row), ivs.append(plan.initial_state(nm, mech, N)), mis.append(mi)
    return mech, np.array(rows), np.array(ivs), mis


def _rel(a, b, E, V, N):
    a, b = a.reshape(E, V, N), b.reshape(E, V, N)
    scale = np.max(np.abs(b), axis=2, keepdims=True)
    scale[scale == 0] = 1.0
    return float(np.max(np.abs(a - b)/scale))


def test_chain_vs_oracle_controller_small_chunks():
    """4 chunks of 64 nodes (N = 200: the last chunk is ragged), two reactors with their own step sequences:
    accept / reject history of the oracle's controller, end state within 50 rtol."""
    N, t1, rtol, atol, h0 = 200, 4e-3, 1e-6, 1e-9, 1e-6
    mech, rows, IV, mis = _members("dme_nb", N, (523, 538))
    dev = N2Device(mech, rows, N, block=64, npt=1, defines={"RMT_RK45_LDS": "2"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.rk45(y, 0.0, t1, rtol, atol, h0, 10**7)
    assert not dev.status().any()
    st = dev.rk45_stats()
    got = y.cpu().numpy()
    for e, mi in enumerate(mis):
        pr = O.setup_n2(mi, N)
        want, ost = O.rk45(O.make_rhs_vec(pr), 0.0, t1, pr["IV"], rtol, atol, h0)
        assert st["t_end"][e] == t1
        assert abs(int(st["accepted"][e]) - ost["accepted"]) <= max(2, 0.02*ost["accepted"]), (e, st, ost)
        assert abs(int(st["rejected"][e]) - ost["rejected"]) <= max(3, 0.05*ost["accepted"]), (e, st, ost)
        assert _rel(got[e], want, 1, mech.V, N) < 50*rtol
    dev.close()


@pytest.mark.parametrize("name,N,E,t1", [("dme_nb", 2500, 5, 3e-3), ("syn12", 1024, 6, 3e-2), ("dme_nb", 2048, 300, 1e-3)])
def test_chain_matches_memory_resident_kernel(name, N, E, t1):
    """The geometry rmtExe picks for reactors beyond one workgroup (rk45_geometry): same step sequences and the
    same end state (rounding level) as rmt_n2_rk45_mem; the third case has more reactors than teams
    (2 chunks -> 128 teams for 300 reactors), so every team integrates several reactors one after the other."""
    mech, rows, IV, _ = _members(name, N, [523 + (e % 9) if name == "dme_nb" else 600 + 3*(e % 9) for e in range(E)])
    block, npt, defs = rk45_geometry(mech.V, N)
    assert block*npt < N                         # the chained geometry
    out, stats = {}, {}
    for mode in ("mem", "chain"):
        dev = N2Device(mech, rows, N, block=block, npt=npt, defines=defs)
        dev.set_mode(mode if mode == "chain" else "mem")
        y = dev.to_device(IV)
        dev.rk45(y, 0.0, 0.4*t1, 1e-6, 1e-9, 1e-6, 10**7)
        dev.rk45(y, 0.4*t1, t1, 1e-6, 1e-9, -1e-6, 10**7)      # resumed: every reactor from its own h_last
        assert not dev.status().any(), mode
        out[mode], stats[mode] = y.cpu().numpy(), dev.rk45_stats()
        dev.close()
    assert np.all(stats["chain"]["t_end"] == t1)
    assert np.array_equal(stats["chain"]["accepted"], stats["mem"]["accepted"])
    assert np.array_equal(stats["chain"]["rejected"], stats["mem"]["rejected"])
    assert _rel(out["chain"], out["mem"], E, mech.V, N) < 1e-11


def test_chain_fp32_build_agrees_with_memory_resident_kernel():
    """`dtype: fp32` code objects (state and kinetics in float, pressure scan in fp64) have the chained stepper
    too: same number of steps within a few, end state within single-precision noise of the memory-resident one."""
    N, E, t1 = 2500, 4, 2e-3
    mech, rows, IV, _ = _members("dme_nb", N, (523, 528, 533, 538))
    block, npt, defs = rk45_geometry(mech.V, N, fp32=True)
    assert block*npt < N
    out, stats = {}, {}
    for mode in ("mem", "chain"):
        dev = N2Device(mech, rows, N, fp32=True, block=block, npt=npt, defines=defs)
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.rk45(y, 0.0, t1, 1e-4, 1e-7, 1e-6, 10**7)
        assert not dev.status().any(), mode
        out[mode], stats[mode] = y.cpu().numpy().astype(np.float64), dev.rk45_stats()
        dev.close()
    assert np.all(np.abs(stats["chain"]["accepted"] - stats["mem"]["accepted"]) <= 3)
    assert _rel(out["chain"], out["mem"], E, mech.V, N) < 2e-3


def test_auto_mode_chains_long_reactors():
    """mode 0 (what rmtExe uses): a code object with the on-chip stepper chains a reactor that does not fit one
    workgroup; the result is the chained kernel's bit for bit."""
    N, E, t1 = 2100, 3, 2e-3
    mech, rows, IV, _ = _members("dme_nb", N, (523, 530, 541))
    block, npt, defs = rk45_geometry(mech.V, N)
    res = {}
    for mode in ("auto", "chain"):
        dev = N2Device(mech, rows, N, block=block, npt=npt, defines=defs)
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.rk45(y, 0.0, t1, 1e-6, 1e-9, 1e-6, 10**7)
        assert not dev.status().any()
        res[mode] = y.cpu().numpy()
        dev.close()
    assert np.array_equal(res["auto"], res["chain"])


def test_chain_stuck_producer_ends_with_step_flag_on_every_member():
    """One chunk stops sending its stage records (debug define): the consumer times out, sets the team's abort
    word, every workgroup drains and the launch ENDS; the reactor in flight and the ones the team never
    started carry RMT_FLAG_STEP."""
    N, E = 1000, 300                     # 4 chunks of 256 nodes; 64 teams -> 4-5 reactors per team
    mech, rows, IV, _ = _members("dme_nb", N, [523 + (e % 5) for e in range(E)])
    dev = N2Device(mech, rows, N, block=128, npt=2, specialize=False,
                   defines={"RMT_RK45_LDS": "2", "RMT_CHAIN_SPINS": "4096", "RMT_CHAIN_TEST_STALL_CHUNK": "1",
                            "RMT_CHAIN_TEST_STALL_FROM": "20"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.rk45(y, 0.0, 1e-3, 1e-6, 1e-9, 1e-6, 10**7)
    flags = dev.status()                 # returns: the launch did not hang
    assert np.all(flags & FLAG_STEP), flags[:8]
    assert np.all(np.isfinite(y.cpu().numpy()))      # the last accepted states, not garbage
    dev.close()


def test_rmtexe_rk45_on_a_mesh_beyond_one_workgroup():
    """rmtExe with ivp 'RK45' at zNo = 1100: the host picks the on-chip geometry and the library chains two chunks
    per reactor; the five output times agree with the fixed-step device RK4 (dt = 2e-6 s, itself 5e-11 from the
    reference's tight LSODA run at zNo = 20) to the tolerance asked for."""
    from rmt_app_amd import rmtExe
    packs = {}
    for ivp, extra in (("hip-rk4", {"dt": 2e-6}), ("RK45", {"rtol": 1e-8, "atol": 1e-11})):
        mi = INP.dme_notebook_input(ivp=ivp)
        mi["solver-config"].update({"quiet": True, "zNo": 1100, **extra})
        res = rmtExe(mi)
        packs[ivp] = res["resModel"]["dataPack"]
        if ivp == "RK45":
            st = res["resModel"]["device-stats"]
            assert st["steps"] > 1000 and st["rejected"] is not None
    worst = 0.0
    for k in range(5):
        a, b = packs["RK45"][k]["dataYs"], packs["hip-rk4"][k]["dataYs"]
        assert a.shape == b.shape == (7, 1100)
        worst = max(worst, float(np.max(np.abs(a - b)/np.maximum(np.abs(b), 1e-300))))
    assert worst < 1e-6, worst


@pytest.mark.parametrize("name,N,E", [("dme_nb", 1024, 40), ("syn12", 512, 30), ("dme_nb", 1000, 1)])
def test_small_ensembles_are_cut_finer_and_agree_with_one_workgroup_per_reactor(name, N, E):
    """rk45_geometry with a known ensemble size: an ensemble that leaves CUs idle is cut into finer chunks (one
    node per lane); same step sequences and the same end states (rounding) as one workgroup per reactor."""
    mech, rows, IV, _ = _members(name, N, [523 + (e % 9) if name == "dme_nb" else 600 + 3*(e % 9) for e in range(E)])
    coarse = rk45_geometry(mech.V, N)
    fine = rk45_geometry(mech.V, N, E=E)
    assert coarse[0]*coarse[1] >= N and fine[0]*fine[1] < N and fine[1] == 1
    out, stats = {}, {}
    for tag, (block, npt, defs) in (("coarse", coarse), ("fine", fine)):
        dev = N2Device(mech, rows, N, block=block, npt=npt, defines=defs)
        y = dev.to_device(IV)
        dev.rk45(y, 0.0, 2e-3 if name == "dme_nb" else 3e-2, 1e-6, 1e-9, 1e-6, 10**7)     # auto mode
        assert not dev.status().any(), tag
        out[tag], stats[tag] = y.cpu().numpy(), dev.rk45_stats()
        dev.close()
    assert np.array_equal(stats["fine"]["accepted"], stats["coarse"]["accepted"])
    assert np.array_equal(stats["fine"]["rejected"], stats["coarse"]["rejected"])
    assert _rel(out["fine"], out["coarse"], E, mech.V, N) < 1e-11


def test_forced_chain_mode_on_a_reactor_that_fits_one_workgroup_is_refused():
    """mode 3 (chained) needs at least two chunks: the library refuses with a message instead of launching."""
    from rmt_app_amd.hipbind import RmtN2Error
    mech, rows, IV, _ = _members("dme_nb", 200, (523,))
    dev = N2Device(mech, rows, 200, block=256, npt=1, defines={"RMT_RK45_LDS": "2"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    with pytest.raises(RmtN2Error, match="chained rk45 needs"):
        dev.rk45(y, 0.0, 1e-4, 1e-6, 1e-9, 1e-6, 10**6)
    dev.set_mode("auto")
    dev.rk45(y, 0.0, 1e-4, 1e-6, 1e-9, 1e-6, 10**6)          # the handle is still usable
    assert not dev.status().any()
    dev.close()
