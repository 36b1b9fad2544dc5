#!/usr/bin/env python3
"""bench.py - mesh-node-steps/s of the N2 hot path on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE configs[1]'s reactor - the TEST2.ipynb DME case (6 species,
3 reactions, fp64) on 1024 axial nodes, classic RK4 with dt = 2e-6 s (the stable step for this case) - replicated as the
per-GPU shard of configs[3]'s ensemble: 256 independent reactors per GPU with the inlet-T /
pressure sweep of SURVEY.md section 8(d).4 (2048 members at 8 GPUs).  One bench "step" = one
output interval of the reference's time loop (pbHomoReactor.py:3589-3690) = ONE device launch of
1000 RK4 time steps (4 RHS evaluations each) of every node of every member on this rank; the metric
counts RK4 steps: value = ranks x members x nodes x steps x 1000 / time.  Ranks are independent
(weak scaling, no data-path collective): rank 0 broadcasts the packed constants once (RCCL),
outlet rows are gathered at the end.

Prints ONE JSON line on rank 0 (see the task contract): value = total mesh-node-steps/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TOPS = 39.3      # 78.6 TFLOP/s fp64 vector spec = 39.3e12 fp64 lane-instructions/s (an FMA counts 2 flops)
N_NODES = 1024
MEMBERS_PER_GPU = 256
DT = 2e-6
RK4_PER_STEP = 1000          # one bench "step" = one output interval = ONE launch of 1000 RK4 steps (2 ms of reactor time)


def tracked_traffic(digest, E, n_nodes):
    """HBM bytes per launch of the kernel with this machine-code digest (isa.kernel_stats), from the tracked PMC record
    profiles/traffic.json (written by tools/record_traffic.py from separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same command).  None when the record is of another code object or shape -
    a kernel change invalidates the number instead of leaving a stale one in the line."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, None
    ent = rec.get(digest)
    if not ent or ent.get("members") != E or ent.get("nodes") != n_nodes:
        return None, None
    return float(ent["bytes_per_launch"]), ent.get("source")


def reference_cpu_rate(n_nodes):
    """RK4-equivalent node-steps/s of the reference's own Python RHS (profiles/reference_cpu.json, measured in
    the build container by tools/time_reference.py: the reference cannot travel to the GPU box)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu.json")))
        case = rec["cases"].get(str(n_nodes)) or rec["cases"]["1024"]
        return {"value": case["rk4_equiv_node_steps_per_s"], "unit": "mesh-node-steps/s (RK4-equivalent: N/(4 s_per_rhs))",
                "cores": 1, "where": "build container, %s" % rec["machine"]["cpu"],
                "source": "profiles/reference_cpu.json (tools/time_reference.py, %s)" % rec["date"]}
    except (OSError, ValueError, KeyError):
        return None


def sweep_member_inputs(first, count, total=2048):
    """Members first..first+count-1 of the 64 (T) x 32 (P) sweep:
    member = iT*32 + iP, T in linspace(503,543,64), P in linspace(3e6,7e6,32); feed concentrations
    y0*P/(R*T) with the config-2 mole fractions."""
    import inputs as INP
    base = INP.dme_notebook_input()
    c0 = np.array(base["feed"]["concentration"], dtype=float)
    y0 = c0/c0.sum()
    Ts, Ps = np.linspace(503.0, 543.0, 64), np.linspace(3.0e6, 7.0e6, 32)
    out = []
    for mem in range(first, first + count):
        m = mem % total
        T, P = float(Ts[m // 32]), float(Ps[m % 32])
        mi = INP.dme_notebook_input()
        mi["operating-conditions"]["temperature"] = T
        mi["operating-conditions"]["pressure"] = P
        mi["feed"]["concentration"] = y0*P/(INP.R_CONST*T)
        out.append(mi)
    return out


def host_cores():
    """CPUs this job may actually use: cgroup quota if set (the GPU box gives 16 of its 256 logical
    CPUs to a 1-GPU job), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota)/int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(mech, rows, IV, N_NODES, seconds=12.0):
    """The oracle side, timed on this box's host cores: host emulation of the same generated
    source (oracle/hostemu_driver.cpp, OpenMP over members) running the identical RK4."""
    from oracle.hostemu import HostEmu
    from rmt_app_amd import hipbind
    emu = HostEmu(mech.source(hipbind.kernel_template()), tag="bench")
    cores = emu.set_threads(host_cores())
    E = min(len(rows), max(2*cores, 8))
    y = IV[:E].copy()
    emu.rk4(y, rows[:E], N_NODES, DT, 2)          # warm-up
    steps, t_used = 0, 0.0
    n = 20
    while t_used < seconds:
        t0 = time.perf_counter()
        y, _ = emu.rk4(y, rows[:E], N_NODES, DT, n)
        t_used += time.perf_counter() - t0
        steps += n
    return {"value": E*N_NODES*steps/t_used, "unit": "mesh-node-steps/s", "cores": cores,
            "kind": "port",
            "sample": "%d members x %d nodes x %d RK4 steps (%.1f s), host emulation of the "
                      "generated kernel source, g++ -O2 -fopenmp" % (E, N_NODES, steps, t_used)}


def accuracy_vs_scipy_reference():
    """Second half of BASELINE.json's metric ("max |dMoFri| vs SciPy ref"): the reference's own
    test input (tests/test_rmt_N1_DME.py, its default 20-node mesh, 0.5 s) integrated on the device
    by the explicit and by the stiff stepper, against the committed golden of the REFERENCE run
    under LSODA rtol=1e-10/atol=1e-12 (tests/golden/g4_tight_dme_script_lsoda.npz, G4); max over the
    5 output times of the relative difference of outlet mole fractions and temperature."""
    import inputs as INP
    from rmt_app_amd import rmtExe
    g = np.load(os.path.join(ROOT, "tests", "golden", "g4_tight_dme_script_lsoda.npz"))
    out = {"reference": "PyREMOT RHS under SciPy LSODA rtol=1e-10 (golden G4), zNo=20, t=0.1..0.5 s"}
    for ivp, extra in (("hip-rk4", {"dt": 2.5e-6}), ("hip-ros4", {"rtol": 1e-6, "atol": 1e-9})):
        mi = INP.dme_script_input(ivp=ivp)
        mi["solver-config"].update(dict(extra, quiet=True))
        t0 = time.perf_counter()
        dp = rmtExe(mi)["resModel"]["dataPack"]
        wall = time.perf_counter() - t0
        worst = wabs = 0.0
        for k in range(5):
            a, b = dp[k]["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
            worst = max(worst, float(np.max(np.abs(a - b)/np.abs(b))))
            wabs = max(wabs, float(np.max(np.abs(dp[k]["dataYs"][:-1] - g["dataYs_%d" % k][:-1]))))
        out[ivp] = {"max_rel_outlet_MoFri_T": worst, "max_abs_dMoFri_all_nodes": wabs,
                    "wall_s": round(wall, 3), **extra}
    return out


def single_reactor_4096(mech, inputs):
    """BASELINE's target case: ONE 6-species / 3-reaction dynamic reactor on 4096 axial nodes, fp64 -
    explicit RK4 on the device (the reactor chained over 32 workgroups) and the whole 0.5 s job with
    the stiff stepper, next to the same generated source on ONE host core.  The reference's own
    per-node Python RHS: profiles/reference_cpu.json (tools/time_reference.py, measured in the build
    container; the reference cannot be run on the GPU box)."""
    import torch
    from oracle.hostemu import HostEmu
    from rmt_app_amd import hipbind, plan
    from rmt_app_amd.n2 import N2Device
    N = 4096
    nm, row = plan.member_constants(inputs[0], mech, N)
    IV = plan.initial_state(nm, mech, N)
    dev = N2Device(mech, row, N)
    y = dev.to_device(IV)
    dev.rk4(y, DT, 200)
    dev.rk4(y, DT, 2000)
    ms = dev.last_kernel_ms()
    ok = not dev.status().any()
    dev.close()
    devr = N2Device(mech, row, N, block=256, npt=1, features=("ros4",))
    from rmt_app_amd.settings import DEVICE_DEFAULTS as D
    walls = {}
    for mode in ("mem", "auto"):          # one workgroup (one CU) vs chained over 16 CUs (what auto picks for E = 1)
        devr.set_mode(mode)
        y = devr.to_device(IV)
        devr.ros4(y, 0.0, 1e-4, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)       # warm-up
        y = devr.to_device(IV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        devr.ros4(y, 0.0, 0.5, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)
        torch.cuda.synchronize()
        walls[mode] = time.perf_counter() - t0
        ok = ok and not devr.status().any()
        geo = devr.last_geometry()
    wall = walls["auto"]
    st = devr.rk45_stats()
    devr.close()
    emu = HostEmu(mech.source(hipbind.kernel_template()), tag="bench")
    emu.set_threads(1)
    yc = IV.copy()
    t0 = time.perf_counter()
    yc, _ = emu.rk4(yc, row, N, DT, 200)
    cpu = N*200/(time.perf_counter() - t0)
    ref = reference_cpu_rate(1024)
    return {"nodes": N, "rk4_node_steps_per_s": N*2000/(ms*1e-3), "rk4_us_per_step": ms/2.0,
            "ros4_whole_0.5s_job_wall_s": round(wall, 4), "ros4_one_workgroup_wall_s": round(walls["mem"], 4),
            "ros4_kernel": "rmt_n2_ros4_%s, %d chunks x %d teams (chosen by the library)" % (
                "chain" if geo[0] > 1 else "mem", geo[0], geo[1]), "ros4_steps": int(st["accepted"][0] + st["rejected"][0]),
            "flags_ok": bool(ok), "cpu_port_1core_node_steps_per_s": cpu,
            "reference_python_rk4_equiv_node_steps_per_s": ref["value"] if ref else None}


def adaptive_rk45(mech, rows, IV, n_nodes):
    """BASELINE configs[4]: adaptive Dormand-Prince RK45 with per-reactor step control on the device
    (rmt_n2_rk45_reg, on chip) - accepted node-steps/s of this rank's sweep over 8 ms of reactor time after
    a warm-up interval, and the 12-species / 8-reaction mechanism on 64 x 512 nodes (rmt_n2_rk45_chain: the
    ensemble alone would leave 3/4 of the CUs idle, so every reactor is cut into 4 chunks) and on this rank's
    share of configs[4], 256 x 1024 nodes (two chunks per reactor)."""
    import inputs as INP
    from rmt_app_amd import plan
    from rmt_app_amd.n2 import N2Device, rk45_geometry
    out = {}
    for tag, mech_, rows_, IV_, N, t1 in (("dme_%dx%d" % (len(rows), n_nodes), mech, rows, IV, n_nodes, 8e-3),
                                          ("syn12_64x512", None, 64, None, 512, 0.1),
                                          ("syn12_256x1024", None, 256, None, 1024, 0.05)):
        if mech_ is None:
            mi = INP.syn12_input()
            mech_ = plan.Mechanism(mi)
            nm, row = plan.member_constants(mi, mech_, N)
            rows_, IV_ = np.tile(row, (rows_, 1)), np.tile(plan.initial_state(nm, mech_, N), (rows_, 1))
        block, npt, defs = rk45_geometry(mech_.V, N, E=len(rows_))      # (what rmtExe picks for this ensemble)
        dev = N2Device(mech_, rows_, N, block=block, npt=npt, defines=defs)
        y = dev.to_device(IV_)
        dev.rk45(y, 0.0, 1e-5, 1e-6, 1e-9, 1e-6, 10**8)
        dev.rk45(y, 1e-5, t1, 1e-6, 1e-9, -1e-6, 10**8)
        ms = dev.last_kernel_ms()
        st = dev.rk45_stats()
        ok = not dev.status().any()
        dev.close()
        out[tag] = {"accepted_node_steps_per_s": N*float(st["accepted"].sum())/(ms*1e-3), "kernel_ms": ms,
                    "accepted_per_reactor_median": int(np.median(st["accepted"])),
                    "rejected_max": int(st["rejected"].max()), "rtol": 1e-6, "atol": 1e-9,
                    "kernel": "rmt_n2_rk45_%s block=%d npt=%d" % (
                        ("reg" if block*npt >= N else "chain[%d]" % -(-N//(block*npt))) if defs else "mem", block, npt),
                    "flags_ok": ok}
    return out


def time_to_solution(mech, rows, IV, n_nodes, t_end=0.5):
    """The user-visible job behind the throughput number: every member of this rank's sweep
    integrated over the full 0.5 s transient with the stiff Rosenbrock stepper (default
    tolerances); the explicit RK4 time is steps*ms_per_step = 250000 steps at dt = 2e-6 s."""
    import torch
    from rmt_app_amd.n2 import N2Device
    from rmt_app_amd.settings import DEVICE_DEFAULTS as D
    dev = N2Device(mech, rows, n_nodes, block=256, npt=1, features=("ros4",))
    y = dev.to_device(IV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.ros4(y, 0.0, t_end, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    flags = dev.status()
    st = dev.rk45_stats()
    dev.close()
    tot = st["accepted"] + st["rejected"]
    return {"integrator": "hip-ros4", "t_end_s": t_end, "wall_s": round(wall, 4), "flags_ok": not bool(flags.any()),
            "steps_per_reactor_min_median_max": [int(tot.min()), int(np.median(tot)), int(tot.max())],
            "rtol": D["ros4-rtol"], "atol": D["ros4-atol"]}


MESH_SWEEP = ((256, 4096), (64, 16384), (1, 16384))       # BASELINE configs[2]: (reactors, nodes)
STREAM_E, STREAM_N = 16384, 1024                           # rhs_stream: 16384 x 1024 x 7 fp64 = 0.94 GB in + 0.94 GB out


def mesh_sweep(mech, inputs):
    """BASELINE configs[2] ("same dynamic model at 4096 and 16384 axial nodes - HBM-roofline sweep"): RK4 with the
    reactors cut into chunks on several CUs (rmt_n2_rk4_chain) for 256 x 4096, 64 x 16384 and ONE 16384-node
    reactor; node-steps/s, the contract's 128 B/node-step figure against 8 TB/s and the fp64 VALU fraction."""
    from rmt_app_amd import isa, plan
    from rmt_app_amd.n2 import N2Device
    out = {}
    for E, N in MESH_SWEEP:
        nm, row = plan.member_constants(inputs[0], mech, N)
        dev = N2Device(mech, np.tile(row, (E, 1)), N)
        y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
        steps = 1000 if E > 1 else 2000
        dev.rk4(y, DT, steps//10)
        dev.rk4(y, DT, steps)
        ms = dev.last_kernel_ms()
        ok = not dev.status().any()
        W = dev.block*dev.npt
        kname = "rmt_n2_rk4_%s" % ("reg" if N <= W else "chain")
        loop = isa.kernel_stats(bytes(dev._code.raw), kname)
        loop = loop.get("step_loop") or loop["whole"]
        rate = E*N*steps/(ms*1e-3)
        out["%dx%d" % (E, N)] = {
            "node_steps_per_s": rate, "kernel_ms": ms, "rk4_steps": steps,
            "kernel": "%s block=%d npt=%d chunks=%d" % (kname, dev.block, dev.npt, -(-N//W)),
            "hbm_contract_frac": rate*2*(mech.S + 2)*8/1e9/HBM_PEAK_GBS,
            "valu_fp64_frac": rate*loop["valu_f64"]/float(dev.npt)/(FP64_PEAK_TOPS*1e12), "flags_ok": bool(ok)}
        dev.close()
    return out


def rhs_stream(mech, inputs):
    """The one regime of this path where the HBM roofline is the real bound: the bare RHS kernel (rmt_n2_rhs: y in,
    dy/dt out) on a state far beyond the 256 MB Infinity Cache - 16384 reactors x 1024 nodes, 0.94 GB read + 0.94 GB
    written per launch.  ALGORITHMIC bytes (2 V 8 B per node) / HIP-event kernel time against 8 TB/s; the PMC
    FETCH_SIZE / WRITE_SIZE of the same launch are in profiles/ (tools/record_traffic.py)."""
    import torch
    from rmt_app_amd import plan
    from rmt_app_amd.n2 import N2Device
    E, N = STREAM_E, STREAM_N
    nm, row = plan.member_constants(inputs[0], mech, N)
    dev = N2Device(mech, np.tile(row, (E, 1)), N, block=STREAM_BLOCK, npt=1, specialize=False)
    y = dev.to_device(np.tile(plan.initial_state(nm, mech, N), (E, 1)))
    out = dev.rhs(y)
    ms = []
    for _ in range(10):
        out = dev.rhs(y)
        ms.append(dev.last_kernel_ms())
    ok = not dev.status().any()
    t = float(np.median(ms))
    by = 2*mech.V*8*E*N
    dev.close()
    del y, out
    torch.cuda.empty_cache()
    return {"reactors": E, "nodes": N, "state_GB_in_plus_out": by/1e9, "kernel": "rmt_n2_rhs block=%d" % STREAM_BLOCK,
            "kernel_ms": t, "node_rhs_per_s": E*N/(t*1e-3), "achieved_GBs": by/(t*1e-3)/1e9, "peak_GBs": HBM_PEAK_GBS,
            "frac": by/(t*1e-3)/1e9/HBM_PEAK_GBS, "achieved_is": "algorithmic bytes (2 V 8 B per node) / kernel time",
            "flags_ok": bool(ok)}


STREAM_BLOCK = 256


def rmtexe_ensemble_wall(n_nodes, members=MEMBERS_PER_GPU):
    """What a user of the sweep sees: rmtExe on this rank's 256-member inlet-T / pressure sweep, whole 0.5 s transient,
    5 output times, stiff stepper - wall time including the member packing, the launches, the device-to-host copies
    of every output time and the dataPack construction; once with every member's full profile returned and once with
    solver-config "ensemble-output": "outlet"."""
    import inputs as INP
    from rmt_app_amd import rmtExe
    out = {"members": members, "nodes": n_nodes, "output_times": 5, "integrator": "hip-ros4"}
    nT = members//32
    for mode in ("profile", "outlet"):
        mi = INP.dme_notebook_input(ivp="hip-ros4")
        mi["solver-config"].update({"quiet": True, "zNo": n_nodes, "tNo": 5, "ensemble-output": mode,
                                    "ensemble": {"temperature": list(np.linspace(503.0, 543.0, 64)[:nT]),
                                                 "pressure": list(np.linspace(3.0e6, 7.0e6, 32))}})
        t0 = time.perf_counter()
        res = rmtExe(mi)["resModel"]
        out["%s_wall_s" % mode] = round(time.perf_counter() - t0, 4)
        assert len(res["ensemble"]) == members and len(res["ensemble"][-1]["dataPack"]) == 5
    out["outlet_T_first_last_member_K"] = [float(res["ensemble"][0]["dataPack"][-1]["dataYs"][6, -1]),
                                           float(res["ensemble"][-1]["dataPack"][-1]["dataYs"][6, -1])]
    return out


def stiff_wide_mechanism():
    """The stiff stepper on the 12-species / 8-reaction mechanism (13 x 13 node Jacobians), 64 reactors x 512 nodes
    over 2 s of reactor time at the default tolerances: the one-node-on-four-lanes layout (kernels/61_ros4_quad.inc)."""
    import torch
    import inputs as INP
    from rmt_app_amd import plan
    from rmt_app_amd.n2 import N2Device, ros4_block
    from rmt_app_amd.settings import DEVICE_DEFAULTS as D
    mi = INP.syn12_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 512)
    dev = N2Device(mech, np.tile(row, (64, 1)), 512, block=ros4_block(mech.V, 512), npt=1, features=("ros4",))
    IV = np.tile(plan.initial_state(nm, mech, 512), (64, 1))
    y = dev.to_device(IV)
    dev.ros4(y, 0.0, 1e-3, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)        # warm-up
    y = dev.to_device(IV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.ros4(y, 0.0, 2.0, D["ros4-rtol"], D["ros4-atol"], D["ros4-h0"], 10**7)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st, ok, geo = dev.rk45_stats(), not dev.status().any(), dev.last_geometry()
    dev.close()
    return {"mechanism": "12 species / 8 reactions (V = 13)", "reactors": 64, "nodes": 512, "t_end_s": 2.0,
            "wall_s": round(wall, 4), "steps": int(st["accepted"][0] + st["rejected"][0]),
            "kernel": "rmt_n2_ros4_%s, quad layout (one node on four lanes), %d chunks x %d teams" % (
                "chain" if geo[0] > 1 else "mem", geo[0], geo[1]), "flags_ok": bool(ok)}


def executed_step_mix(loop, kname, dev_defines, compile_variant):
    """Instructions EXECUTED per time step of the bench kernel.  A stepper that moves its cache's reference point only every
    K-th step (RMT_KC_REFRESH, csrc/kernels/50_rk4.inc) holds two versions of stage 1 in its step loop, one of which runs
    per step: the static count of the loop is not what runs.  Two counting builds of the same source give the parts -
    A: every step refreshes (K = 1), C: no step does (RMT_TIMING_KC_NEVER, results wrong, never launched) - and a step
    executes A/K + C(1 - 1/K).  -> (mix dict, note)"""
    from rmt_app_amd import isa
    from rmt_app_amd.n2 import kc_period
    K = kc_period(dev_defines, DT)
    if not (kname == "rmt_n2_rk4_reg" and str(dev_defines.get("RMT_KCACHE", "0")) == "1" and K > 1):
        return loop, ""
    A = isa.kernel_stats(compile_variant({"RMT_KC_REFRESH": "1"}), kname)["step_loop"]
    Cn = isa.kernel_stats(compile_variant({"RMT_TIMING_KC_NEVER": "1"}), kname)["step_loop"]
    mix = {k: A[k]/K + Cn[k]*(1.0 - 1.0/K) for k in ("valu", "valu_f64", "lane_moves")}
    mix["scratch"] = loop["scratch"]            # (spills are a property of the build that runs)
    note = ("; executed per step = (refresh step: %d of %d VALU)/%d + (other steps: %d of %d)*%d/%d, the loop holds both "
            "versions of stage 1" % (A["valu_f64"], A["valu"], K, Cn["valu_f64"], Cn["valu"], K - 1, K))
    return mix, note


def prebuild(members=MEMBERS_PER_GPU, n_nodes=N_NODES):
    """Cross-compile (hipRTC; no GPU) every code object the default `python bench.py` run loads, into the in-tree
    cache that travels with the repository - called by __graft_entry__.build(), so the bench on a fresh GPU box
    spends its time measuring, not JIT-compiling.  Mirrors the device constructions below one by one."""
    import inputs as INP
    from rmt_app_amd import plan
    from rmt_app_amd.ensemble import DistributedEnsemble
    from rmt_app_amd.n2 import compile_mechanism, precompile, rk45_geometry, ros4_block
    inputs = sweep_member_inputs(0, members, total=max(2048, members))
    mech = plan.Mechanism(inputs[0])
    ens = DistributedEnsemble(mech, inputs, n_nodes,
                              compile_fn=lambda mdef: compile_mechanism(mech, n_nodes, defines=mdef, E=members))
    rows = ens.rows
    keys = ["main sweep kernel (%d bytes)" % len(ens.code)]
    for extra in ({"RMT_KC_REFRESH": "1"}, {"RMT_TIMING_KC_NEVER": "1"}):      # executed_step_mix's counting builds
        keys.append("counting build (%d bytes)" % len(compile_mechanism(
            mech, n_nodes, defines={**ens.member_defines, **extra}, E=members)))
    # accuracy_vs_scipy_reference: rmtExe on the reference's test input, zNo = 20
    mi = INP.dme_script_input()
    m20 = plan.Mechanism(mi)
    _, r20 = plan.member_constants(mi, m20, 20)
    keys.append(precompile(m20, r20, 20))
    keys.append(precompile(m20, r20, 20, block=ros4_block(m20.V, 20), npt=1, features=("ros4",)))
    # single_reactor_4096
    _, r4096 = plan.member_constants(inputs[0], mech, 4096)
    keys.append(precompile(mech, r4096, 4096))
    keys.append(precompile(mech, r4096, 4096, block=256, npt=1, features=("ros4",)))
    # adaptive_rk45: the sweep and the 12-species mechanism
    block, npt, defs = rk45_geometry(mech.V, n_nodes, E=members)
    keys.append(precompile(mech, rows, n_nodes, block=block, npt=npt, defines=defs))
    ms = plan.Mechanism(INP.syn12_input())
    _, rs = plan.member_constants(INP.syn12_input(), ms, 512)
    block, npt, defs = rk45_geometry(ms.V, 512, E=64)
    keys.append(precompile(ms, np.tile(rs, (64, 1)), 512, block=block, npt=npt, defines=defs))
    _, rs = plan.member_constants(INP.syn12_input(), ms, 1024)
    block, npt, defs = rk45_geometry(ms.V, 1024, E=256)
    keys.append(precompile(ms, np.tile(rs, (256, 1)), 1024, block=block, npt=npt, defines=defs))
    # time_to_solution / rmtexe_ensemble_wall
    keys.append(precompile(mech, rows, n_nodes, block=256, npt=1, features=("ros4",)))
    # mesh_sweep, rhs_stream
    for E, N in MESH_SWEEP:
        _, r = plan.member_constants(inputs[0], mech, N)
        keys.append(precompile(mech, np.tile(r, (E, 1)), N))
    _, r = plan.member_constants(inputs[0], mech, STREAM_N)
    keys.append(precompile(mech, np.tile(r, (2, 1)), STREAM_N, block=STREAM_BLOCK, npt=1, specialize=False))
    # stiff_wide_mechanism
    _, rs = plan.member_constants(INP.syn12_input(), ms, 512)
    keys.append(precompile(ms, np.tile(rs, (64, 1)), 512, block=ros4_block(ms.V, 512), npt=1, features=("ros4",)))
    return keys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10,
                    help="timed bench steps; one step = one launch of %d RK4 steps of every reactor" % RK4_PER_STEP)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--members", type=int, default=MEMBERS_PER_GPU, help="reactors per GPU")
    ap.add_argument("--nodes", type=int, default=N_NODES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="auto", choices=["auto", "reg", "mem"])
    ap.add_argument("--block", type=int, default=None)
    ap.add_argument("--npt", type=int, default=None)
    ap.add_argument("--lds", type=int, default=None, help="RK4 vectors kept in LDS (0,1,2)")
    ap.add_argument("--no-specialize", action="store_true", help="keep all member fields run-time")
    ap.add_argument("--copt", default="", help="extra hipRTC compiler options (tuning experiments)")
    ap.add_argument("--define", action="append", default=[], help="kernel tuning macro NAME=VALUE")
    args = ap.parse_args()

    # `python bench.py --gpus N` starts its N ranks itself (one process per GPU under torch.distributed.run,
    # rendezvous on 127.0.0.1) BEFORE this process touches the GPU, and exits with their status; under the
    # driver's own torchrun line the ranks arrive here with RANK / WORLD_SIZE set and simply run.
    from rmt_app_amd import launch
    if args.gpus > 1 and not launch.is_rank():
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started as one of %d ranks: launch it with "
                         "--nproc-per-node %d (or plainly, and it starts the ranks itself)"
                         % (args.gpus, world, args.gpus))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit("rank %d (local %d) has no GPU: %d visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # under torchrun, even alone
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from rmt_app_amd import plan
    from rmt_app_amd.ensemble import DistributedEnsemble
    from rmt_app_amd.n2 import N2Device, compile_mechanism
    n_nodes = args.nodes
    E = args.members
    total = world*E
    # every rank describes the whole sweep cheaply (dict literals); DistributedEnsemble packs only
    # the rank's own contiguous block of members and receives rank 0's code object over RCCL
    inputs = sweep_member_inputs(0, total, total=max(2048, total))
    mech = plan.Mechanism(inputs[0])
    defines = dict(d.split('=', 1) for d in args.define)
    ens = DistributedEnsemble(
        mech, inputs, n_nodes, device=torch.device("cuda", local),
        compile_fn=lambda mdef: compile_mechanism(mech, n_nodes, block=args.block, npt=args.npt,
                                                  lds_state=args.lds, defines={**defines, **mdef}, E=E,
                                                  extra_opts=args.copt))
    rows, IV = ens.rows, ens.IV
    assert rows.shape[0] == E
    if args.no_specialize:
        ens.member_defines.clear()
    dev = N2Device(mech, rows, n_nodes, block=args.block, npt=args.npt, lds_state=args.lds,
                   defines={**defines, **ens.member_defines}, specialize=False,
                   code=None if args.no_specialize else ens.code, extra_opts=args.copt)
    dev.set_mode(args.mode)
    y = dev.to_device(IV)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dev.rk4(y, DT, RK4_PER_STEP)
    barrier()
    # HIP events on the stream the library launches on (N2Device binds torch's current stream)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):               # back-to-back launches on one stream, no host sync in between
        dev.rk4(y, DT, RK4_PER_STEP)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)/args.steps        # average launch duration over the timed region
    flags = dev.status()
    if flags.any():
        raise SystemExit("device flags set during the bench: %s" % flags[flags != 0][:4])
    # reactor-launches the caching stepper handed to its plain twin (warm-up + timed region): 0 = every timed launch was
    # the cached kernel alone, nothing was integrated twice
    fallbacks = dev.fallbacks()
    tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())
    # what the driver needs to see that RCCL really had N ranks: the group's size and every rank's own kernel time
    kms = torch.tensor([kernel_ms], device="cuda", dtype=torch.float64)
    if distributed:
        parts = [torch.zeros_like(kms) for _ in range(world)]
        dist.all_gather(parts, kms)
        kernel_ms_ranks = [float(p.item()) for p in parts]
        rccl_ranks, backend = dist.get_world_size(), dist.get_backend()
    else:
        kernel_ms_ranks, rccl_ranks, backend = [float(kernel_ms)], 1, None
    outlet = ens.gather_outlet(y)          # [world*E][V] on rank 0: the sweep's result table
    if rank == 0:
        assert outlet.shape == (total, mech.V) and bool(torch.isfinite(outlet).all())

    code_blob = bytes(dev._code.raw)
    dev.close()
    if rank == 0:
        from rmt_app_amd import isa
        node_steps = world*E*n_nodes*args.steps*RK4_PER_STEP
        value = node_steps/tmax
        bytes_per_node_step = 2*(mech.S + 2)*8
        # kernel_ms = average launch duration in the timed region (HIP events on the launch stream)
        achieved = (E*n_nodes*RK4_PER_STEP*bytes_per_node_step/1e9)/(kernel_ms/1e3)
        # fp64 VALU instructions per node-step: counted in the step loop of the code object that was
        # launched (llvm-objdump), not a constant; HBM traffic: the tracked PMC record of THIS code object
        kname = "rmt_n2_rk4_%s" % ("mem" if args.mode == "mem" else
                                   ("reg" if n_nodes <= dev.block*dev.npt else "chain"))
        ist = isa.kernel_stats(code_blob, kname)
        loop, mix_note = executed_step_mix(
            ist.get("step_loop") or ist["whole"], kname, dev.defines,
            lambda extra: compile_mechanism(mech, n_nodes, block=args.block, npt=args.npt, lds_state=args.lds,
                                            defines={**defines, **ens.member_defines, **extra}, E=E,
                                            extra_opts=args.copt))
        f64_ops = loop["valu_f64"]/float(dev.npt)
        traffic, traffic_src = tracked_traffic(ist["kernel_digest"], E, n_nodes)
        valu_rate = E*n_nodes*RK4_PER_STEP*f64_ops/(kernel_ms/1e3)
        line = {
            "metric": "mesh-node-steps/s (6-sp DME dynamic model); max |\u0394MoFri| vs SciPy ref",
            "value": value, "unit": "mesh-node-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3*tmax/args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rccl_ranks": rccl_ranks, "dist_backend": backend, "kernel_ms_per_rank": kernel_ms_ranks,
            "config": {"workload": "DME N2 (TEST2.ipynb reactor), %d nodes, RK4 dt=2e-6 s, %d "
                                   "reactors/GPU of the 64x32 inlet-T/P sweep; one step = one output interval "
                                   "= one launch of %d RK4 steps" % (n_nodes, E, RK4_PER_STEP),
                       "rk4_steps_per_step": RK4_PER_STEP,
                       "members_per_gpu": E, "nodes": n_nodes, "integrator": "rk4",
                       "parallelism": "ensemble-dp%d" % world,
                       "extras": ("full (cpu_baseline, accuracy, single reactor, mesh sweep, adaptive, time to "
                                  "solution)" if (world == 1 and not args.no_cpu_baseline) else
                                  "throughput line only (the extra measurements run at N = 1)"),
                       "kernel": "%s block=%d npt=%d lds_state=%d" % (kname, dev.block, dev.npt, dev.lds_state),
                       "kernel_digest": ist["kernel_digest"], "code_object_digest": ist["digest"],
                       "cache_fallbacks_rank0": fallbacks},
            # contract form: ALGORITHMIC bytes (2(S+2)8 B per node-step) / kernel time against the
            # HBM peak.  The state stays on chip for all steps of a launch, so this is NOT the kernel's HBM
            # use (that is `measured_hbm_GBs` = PMC traffic / kernel time, ~1000x smaller) and 1/frac is not
            # bandwidth headroom: the limiter is fp64 VALU issue, priced in `valu_fp64`.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved/HBM_PEAK_GBS, "achieved_is": "algorithmic bytes / kernel time",
                         "limiter": "valu_fp64", "limiter_frac": valu_rate/(FP64_PEAK_TOPS*1e12),
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE+WRITE_SIZE)",
                         "traffic_source": traffic_src,
                         "measured_hbm_GBs": (traffic/1e9)/(kernel_ms/1e3) if traffic else None,
                         "kernel_ms": kernel_ms, "bytes_per_node_step": bytes_per_node_step},
            # the real ceiling of this kernel: fp64 vector issue; ops/node-step = v_*_f64 instructions in
            # the step loop of the launched code object / nodes per lane (rmt_app_amd/isa.py)
            "valu_fp64": {"ops_per_node_step": f64_ops, "ops_source": "llvm-objdump of code object %s, step loop "
                          "of %s: %d v_*_f64 of %d VALU, %d scratch, %d lane moves%s" % (
                              ist["digest"], kname, loop["valu_f64"], loop["valu"], loop["scratch"], loop["lane_moves"],
                              mix_note),
                          "achieved_Tops": valu_rate/1e12, "peak_Tops": FP64_PEAK_TOPS,
                          "frac": valu_rate/(FP64_PEAK_TOPS*1e12),
                          # what this instruction mix could reach with no stall at all, from the measured issue costs
                          # of profiles/round2_issue_model.md (2 waves/SIMD: 4.8 cycles per fp64 VALU instruction,
                          # ~3 per other VALU instruction, against the 4 cycles `peak_Tops` assumes)
                          "mix_ceiling_frac": 4.0*loop["valu_f64"]/(4.8*loop["valu_f64"]
                                                                    + 3.0*(loop["valu"] - loop["valu_f64"])),
                          "mix_ceiling_source": "profiles/round2_issue_model.md (tools/microbench/issue.hip)"},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(mech, rows, IV, n_nodes)
            line["cpu_baseline"]["reference_python"] = reference_cpu_rate(n_nodes)
            line["accuracy"] = accuracy_vs_scipy_reference()
            line["single_reactor_4096"] = single_reactor_4096(mech, inputs)
            line["mesh_sweep"] = mesh_sweep(mech, inputs)
            line["rhs_stream"] = rhs_stream(mech, inputs)
            line["adaptive_rk45"] = adaptive_rk45(mech, rows, IV, n_nodes)
            line["time_to_solution"] = time_to_solution(mech, rows, IV, n_nodes)
            line["time_to_solution"]["rk4_equivalent_wall_s"] = round(250000*tmax/(args.steps*RK4_PER_STEP), 3)
            line["rmtexe_ensemble_wall_s"] = rmtexe_ensemble_wall(n_nodes, E)
            line["stiff_wide_mechanism"] = stiff_wide_mechanism()
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
