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


# ----------------------------------------------------------------------------- rmtExe over the ranks
def _exe_input(model):
    if model == "N2":
        mi = INP.dme_notebook_input(ivp="hip-rk4", period=2e-4)
        mi["solver-config"].update({"quiet": True, "dt": 2e-6, "zNo": 48, "tNo": 2})
    else:
        mi = INP.m2_dme_input(ivp="hip-rk4", period=2e-4)
        mi["solver-config"].update({"quiet": True, "dt": 2e-6, "zNo": 48, "tNo": 2, "display-result": "False"})
    mi["solver-config"]["ensemble"] = {"temperature": list(np.linspace(513.0, 533.0, 5)), "pressure": [5.0e6]}
    return mi


def _run_exe(model, fail_member=None):
    """rmtExe with the device replaced by the host-emulation stand-in (tests/emu_device.py)."""
    import emu_device
    from rmt_app_amd import n2, rmtExe
    real_device, n2.N2Device = n2.N2Device, emu_device.EmuDevice
    try:
        mi = _exe_input(model)
        if fail_member is not None:          # one member far outside the stable step: its rank raises
            spec = mi["solver-config"]["ensemble"]
            mi["solver-config"]["ensemble"] = [
                {"operating-conditions": {"temperature": float(T)}} for T in spec["temperature"]]
            mi["solver-config"]["ensemble"][fail_member]["feed"] = {"volumetric-flowrate": 1e3}
        return rmtExe(mi)["resModel"], emu_device.CREATED
    finally:
        n2.N2Device = real_device        # the stand-in must not leak into other tests of this process


def _exe_worker(rank, world, port, out_path, model, fail_member):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if fail_member is not None:
            try:
                _run_exe(model, fail_member)
                raised = "none"
            except (FloatingPointError, RuntimeError) as e:
                raised = type(e).__name__
            with open("%s.%d" % (out_path, rank), "w") as f:
                f.write(raised)
            return
        res, created = _run_exe(model)
        assert res["ensemble-shard"] == {"rank": rank, "world": world, "members": [0, 3] if rank == 0 else [3, 5]}
        E, magic, defs = created[-1]
        assert E == (3 if rank == 0 else 2) and magic == b"\x7fELF"       # rank 0's code object arrived
        assert "RMT_MC_P0" in defs and "RMT_MC_TF" not in defs                # literals agreed over the ranks
        if rank == 0:
            assert len(res["ensemble"]) == 5 and res["device-stats"]["ranks"] == 2
            np.savez(out_path, **{"m%d_t%d" % (e, k): m["dataPack"][k]["dataYs"]
                                  for e, m in enumerate(res["ensemble"]) for k in range(2)})
        else:
            assert res["ensemble"] is None and res["dataPack"] == []
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model", ["N2", "M2"])
def test_rmtexe_shards_the_ensemble_over_ranks(tmp_path, model):
    """rmtExe itself under a world-size-2 process group: every member's dataPack on rank 0 equals the
    single-process run bit for bit (same kernel source, same arithmetic, different partition)."""
    out = str(tmp_path / "exe.npz")
    mp.start_processes(_exe_worker, args=(2, _free_port(), out, model, None), nprocs=2, join=True,
                       start_method="spawn")
    got = np.load(out)
    res, _ = _run_exe(model)                      # this process: no process group -> single-process path
    assert "ensemble-shard" not in res and len(res["ensemble"]) == 5
    for e, m in enumerate(res["ensemble"]):
        for k in range(2):
            np.testing.assert_array_equal(got["m%d_t%d" % (e, k)], m["dataPack"][k]["dataYs"])
    assert np.ptp([m["dataPack"][1]["dataYs"][6, -1] for m in res["ensemble"]]) > 0


@pytest.mark.timeout(900)
def test_rmtexe_failure_on_one_rank_raises_on_all(tmp_path):
    """A member that blows up on rank 1 must not leave rank 0 waiting in the gather: both ranks raise."""
    out = str(tmp_path / "fail")
    mp.start_processes(_exe_worker, args=(2, _free_port(), out, "N2", 4), nprocs=2, join=True,
                       start_method="spawn")
    r0, r1 = open(out + ".0").read(), open(out + ".1").read()
    assert r1 == "FloatingPointError" and r0 == "RuntimeError", (r0, r1)


# ----------------------------------------------------------------------------- rank-local failures outside the launches
def _construct_fail_worker(rank, world, port, out_path, where):
    """One rank's device stand-in fails at CONSTRUCTION (module load / allocation on a real device) or when the
    state is moved to the device: every rank must raise before the next collective, none may sit in it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        import emu_device
        from rmt_app_amd import n2, rmtExe

        class Failing(emu_device.EmuDevice):
            def __init__(self, *a, **k):
                if where == "create" and rank == 1:
                    raise MemoryError("hipMalloc failed (simulated) on rank 1")
                super().__init__(*a, **k)

            def to_device(self, y):
                if where == "to_device" and rank == 1:
                    raise MemoryError("H2D copy failed (simulated) on rank 1")
                return super().to_device(y)
        real_device, n2.N2Device = n2.N2Device, Failing
        try:
            rmtExe(_exe_input("N2"))
            raised = "none"
        except (MemoryError, RuntimeError) as e:
            raised = "%s: %s" % (type(e).__name__, e)
        finally:
            n2.N2Device = real_device
        with open("%s.%d" % (out_path, rank), "w") as f:
            f.write(raised)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("where", ["create", "to_device"])
def test_rank_local_failure_before_the_first_launch_raises_on_all_ranks(tmp_path, where):
    out = str(tmp_path / "cfail")
    mp.start_processes(_construct_fail_worker, args=(2, _free_port(), out, where), nprocs=2, join=True,
                       start_method="spawn")
    r0, r1 = open(out + ".0").read(), open(out + ".1").read()
    assert r1.startswith("MemoryError") and "rank 1" in r1, r1
    assert r0.startswith("RuntimeError") and "failed on rank 1" in r0, r0
