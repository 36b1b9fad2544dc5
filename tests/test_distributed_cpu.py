"""Multi-process (gloo, world_size 2) tests of the ensemble sharding path on the CPU: partition,
broadcast of the rank-0 code object, gather of outlet rows.  The device integrator is replaced by
the host emulation of the generated source (test infrastructure) so that no GPU is needed; the
communication code under test is exactly what bench.py and rmt_app_amd.ensemble use with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs as INP
from rmt_app_amd import ensemble as ENS
from rmt_app_amd import hipbind, plan
from rmt_app_amd.n2 import compile_mechanism


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _members(n, model="N2"):
    base = INP.dme_notebook_input() if model == "N2" else INP.m2_dme_input()
    return ENS.expand_members(base, {"temperature": np.linspace(513, 533, n), "pressure": [5.0e6]})


def _worker(rank, world, port, n_members, N, out_path, model="N2"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.hostemu import HostEmu
        members = _members(n_members, model)
        mech = plan.Mechanism(members[0])
        de = ENS.DistributedEnsemble(mech, members, N,
                                     compile_fn=lambda mdef: compile_mechanism(mech, N, defines=mdef))
        assert "RMT_MC_P0" in de.member_defines and "RMT_MC_TF" not in de.member_defines   # T sweep
        assert de.code[:4] == b"\x7fELF"                       # every rank got rank 0's code object
        assert sum(de.counts) == n_members and de.hi - de.lo == de.counts[rank]
        emu = HostEmu(mech.source(hipbind.kernel_template(), defines=de.member_defines), tag="dist",
                      openmp=False)
        y, flags = emu.rk4(de.IV, de.rows, N, 2e-6, 25)
        assert not flags.any()
        outlet = de.gather_outlet(torch.from_numpy(y))
        if rank == 0:
            np.save(out_path, outlet.numpy())
        else:
            assert outlet is None
    finally:
        dist.destroy_process_group()


def test_shard_partition_is_contiguous_and_balanced():
    for n in (1, 2, 7, 256, 2048):
        for w in (1, 2, 3, 8):
            blocks = [ENS.shard(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_expand_members_sweep_and_overrides():
    base = INP.dme_notebook_input()
    ms = ENS.expand_members(base, {"temperature": [503.0, 543.0], "pressure": [3e6, 5e6, 7e6]})
    assert len(ms) == 6
    assert ms[4]["operating-conditions"]["temperature"] == 543.0
    assert ms[4]["operating-conditions"]["pressure"] == 5e6
    c = np.asarray(ms[4]["feed"]["concentration"])
    assert abs(c.sum() - 5e6/(8.314472*543.0)) < 1e-9*c.sum()
    assert base["operating-conditions"]["temperature"] == 523       # base untouched
    ms = ENS.expand_members(base, [{"external-heat": {"MeTe": 500}}, {}])
    assert ms[0]["external-heat"]["MeTe"] == 500 and ms[0]["external-heat"]["OvHeTrCo"] == 50
    assert ms[1]["external-heat"]["MeTe"] == 523


@pytest.mark.timeout(600)
@pytest.mark.parametrize("model", ["N2", "M2"])
def test_two_rank_ensemble_matches_single_process(tmp_path, model):
    n_members, N, world = 5, 48, 2
    out = str(tmp_path / "outlet.npy")
    mp.start_processes(_worker, args=(world, _free_port(), n_members, N, out, model), nprocs=world,
                       join=True, start_method="spawn")
    got = np.load(out)
    # single-process reference of the same ensemble
    from oracle.hostemu import HostEmu
    members = _members(n_members, model)
    mech = plan.Mechanism(members[0])
    pack, init = ((plan.member_constants, plan.initial_state) if model == "N2"
                  else (plan.member_constants_m2, plan.initial_state_m2))
    pairs = [pack(mi, mech, N) for mi in members]
    rows = np.array([r for _, r in pairs])
    IV = np.array([init(nm, mech, N) for nm, _ in pairs])
    emu = HostEmu(mech.source(hipbind.kernel_template()), tag="dist", openmp=False)
    y, _ = emu.rk4(IV, rows, N, 2e-6, 25)
    want = y.reshape(n_members, mech.V, N)[:, :, -1]
    assert got.shape == (n_members, mech.V)
    np.testing.assert_array_equal(got, want)
    assert np.ptp(got[:, 6]) > 0            # members really differ (inlet T sweep)
