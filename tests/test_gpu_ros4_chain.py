"""GPU tests of the chained stiff stepper (rmt_n2_ros4_chain): ONE reactor cut into chunks that run on
different CUs, coupled through tagged-word links (include/rmt_n2.h; csrc/kernels/30_lanes_links.inc rmt_link_send).
Checked against the single-workgroup kernel (an independent path for everything that crosses a chunk
boundary: pressure / upstream records, Jacobi boundary iterates, error norm, step decision), against the CPU
oracle's Rosenbrock controller at BASELINE's 4096-node target shape, and for its time-out path."""
import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import plan
from rmt_app_amd.lowering import FLAG_STEP
from rmt_app_amd.n2 import N2Device

pytestmark = pytest.mark.gpu


def _members(N, temps):
    mech = plan.Mechanism(INP.dme_notebook_input())
    rows, named, mis = [], [], []
    for T in temps:
        mi = INP.dme_notebook_input()
        mi["operating-conditions"]["temperature"] = T
        nm, row = plan.member_constants(mi, mech, N)
        rows.append(row), named.append(nm), mis.append(mi)
    return mech, np.array(rows), np.array([plan.initial_state(nm, mech, N) for nm in named]), mis


@pytest.mark.parametrize("N,temps,block", [(1024, (513, 523, 538), 256), (1000, (523, 533), 128),
                                           (4096, tuple(503 + e for e in range(40)), 256)])
def test_chained_stiff_stepper_agrees_with_single_workgroup(N, temps, block):
    """4 chunks x 3 teams; ragged last chunk (1000 nodes in 8 blocks of 128); 40 reactors x 4096 nodes =
    6 chunks of 3 node blocks each.  Same accept/reject history, states equal to the accuracy of the linear
    solves (the single-workgroup kernel hands block boundaries over exactly, the chain iterates on them)."""
    mech, rows, IV, _ = _members(N, temps)
    E = len(temps)
    dev = N2Device(mech, rows, N, block=block, npt=1, features=("ros4",))
    rtol, atol, h0, t1 = 1e-6, 1e-9, 1e-5, 0.02
    out, st = {}, {}
    for mode in ("mem", "chain"):
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.ros4(y, 0.0, t1, rtol, atol, h0, 10**6)
        assert not dev.status().any(), mode
        out[mode], st[mode] = y.cpu().numpy().reshape(E, 7, N), dev.rk45_stats()
    assert np.all(st["chain"]["t_end"] == t1)
    np.testing.assert_array_equal(st["chain"]["accepted"], st["mem"]["accepted"])
    np.testing.assert_array_equal(st["chain"]["rejected"], st["mem"]["rejected"])
    scale = np.max(np.abs(out["mem"]), axis=2, keepdims=True)
    assert np.max(np.abs(out["chain"] - out["mem"])/scale) < 1e-7
    assert len(set(st["chain"]["accepted"].tolist())) > 1 or E == 1      # per-reactor step control survives
    dev.close()


@pytest.mark.parametrize("chunks", [None, "2"])
def test_chained_stiff_stepper_70_reactors_rounds_of_teams_and_multi_block_chunks(chunks, monkeypatch):
    """70 reactors x 1024 nodes.  The host's estimate picks 4 one-block chunks per reactor: 64 teams, so six teams
    integrate a second reactor after their first (rounds).  RMT_N2_ROS4_CHUNKS=2 forces 2 chunks of 2 node blocks:
    the carry between the blocks of a chunk and the link between the chunks in one launch."""
    N, E = 1024, 70
    mech, rows, IV, _ = _members(N, tuple(503 + 0.5*e for e in range(E)))
    dev = N2Device(mech, rows, N, block=256, npt=1, features=("ros4",))
    res = {}
    for mode in ("mem", "chain"):
        if chunks and mode == "chain":
            monkeypatch.setenv("RMT_N2_ROS4_CHUNKS", chunks)
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.ros4(y, 0.0, 5e-3, 1e-6, 1e-9, 1e-5, 10**6)
        monkeypatch.delenv("RMT_N2_ROS4_CHUNKS", raising=False)
        assert not dev.status().any(), mode
        res[mode] = (y.cpu().numpy().reshape(E, 7, N), dev.rk45_stats()["accepted"].copy())
    np.testing.assert_array_equal(res["chain"][1], res["mem"][1])
    scale = np.max(np.abs(res["mem"][0]), axis=2, keepdims=True)
    assert np.max(np.abs(res["chain"][0] - res["mem"][0])/scale) < 1e-7
    dev.close()


def test_single_reactor_4096_nodes_vs_oracle_controller():
    """BASELINE's target shape: ONE 6-species / 3-reaction reactor on 4096 nodes, stiff stepper chained over
    16 CUs (auto mode), against the oracle's RODAS4 controller (exact bidiagonal solves, finite-difference
    Jacobian): same step history, end state within 20 rtol."""
    N = 4096
    mech, rows, IV, mis = _members(N, (523,))
    dev = N2Device(mech, rows, N, block=256, npt=1, features=("ros4",))
    y = dev.to_device(IV)
    rtol, atol, h0, t1 = 1e-6, 1e-9, 1e-5, 0.01
    dev.ros4(y, 0.0, t1, rtol, atol, h0, 10**6)
    assert not dev.status().any()
    st = dev.rk45_stats()
    pr = O.setup_n2(mis[0], N)
    want, ost = O.ros4(pr, pr["IV"], 0.0, t1, rtol, atol, h0, scheme="rodas4")
    assert st["t_end"][0] == t1
    assert abs(int(st["accepted"][0]) - ost["accepted"]) <= max(3, 0.03*ost["accepted"]), (st, ost)
    got = y.cpu().numpy()[0].reshape(7, N)
    scale = np.max(np.abs(want.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(got - want.reshape(7, N))/scale) < 20*rtol
    dev.close()


def test_chained_stiff_stepper_stuck_producer_ends_with_step_flag():
    """Chunk 1 of every chain stops sending its stage records (debug define): its consumer times out, raises
    the team's abort word, every wait of the team returns, the launch ENDS and every reactor is flagged."""
    N, E = 1024, 5                       # 4 chunks of one node block, 5 teams
    mech, rows, IV, _ = _members(N, tuple(503 + 5*e for e in range(E)))
    dev = N2Device(mech, rows, N, block=256, npt=1, features=("ros4",), specialize=False,
                   defines={"RMT_CHAIN_SPINS": "8192", "RMT_CHAIN_TEST_STALL_CHUNK": "1",
                            "RMT_CHAIN_TEST_STALL_FROM": "40"})
    dev.set_mode("chain")
    y = dev.to_device(IV)
    dev.ros4(y, 0.0, 5e-3, 1e-6, 1e-9, 1e-5, 10**6)
    flags = dev.status()
    assert np.all(flags & FLAG_STEP), flags[:10]
    with pytest.raises(RuntimeError, match="step"):
        dev.ros4(y, 0.0, 5e-3, 1e-6, 1e-9, 1e-5, 10**6)
        dev.raise_on_flags()
    dev.close()


def test_chained_stiff_stepper_fp32():
    """dtype fp32 (state and kinetics in float, pressure scan and the links in double): the chained and the
    one-workgroup kernel take the same steps and agree to single precision."""
    N, temps = 1024, (513, 533)
    mech, rows, IV, _ = _members(N, temps)
    dev = N2Device(mech, rows, N, fp32=True, block=256, npt=1, features=("ros4",))
    out, acc = {}, {}
    for mode in ("mem", "chain"):
        dev.set_mode(mode)
        y = dev.to_device(IV)
        dev.ros4(y, 0.0, 0.01, 1e-4, 1e-7, 1e-5, 10**6)
        assert not dev.status().any(), mode
        out[mode], acc[mode] = y.cpu().numpy().astype(np.float64).reshape(2, 7, N), dev.rk45_stats()["accepted"].copy()
    assert np.all(np.abs(acc["chain"] - acc["mem"]) <= 2), acc
    scale = np.max(np.abs(out["mem"]), axis=2, keepdims=True)
    assert np.max(np.abs(out["chain"] - out["mem"])/scale) < 1e-4
    dev.close()
