"""Model "M2" on the device: the reference's DIMENSIONAL dynamic packed-bed model
(PackedBedReactorClass.runM2 / modelEquationM2, PyREMOT/docs/pbReactor.py:552-842, 845-1165;
dispatched by rmtCore.M2Init, PyREMOT/docs/rmtCore.py:239-249).  SURVEY.md section 8(f) rank 3.

Same kernel generator, same steppers and C ABI as N2 (``RMT_MODEL 2`` selects the M2 node
functions in csrc/kernels/21_node_m2.inc): concentrations in kmol/m^3, EOS gas velocity - hence a
nonlinear Ergun march, solved on the device by Newton sweeps over the affine scan - and the
catalyst's thermal mass in the energy balance.

The reference returns only plot lists from runM2 (pbReactor.py:835-840: the temperature series
of every output time); ``run_m2`` returns exactly those two keys plus ``dataPack`` (the per-interval
records the reference builds internally, :745-753), ``computation-time`` and ``device-stats``.
"""
from timeit import default_timer as timer

import numpy as np

from . import plan
from .n2 import (ROUND_FUN_ACCURACY, integrate_intervals, mechanism_for, open_auto, open_members, resolve_ivp,
                 rk45_geometry, ros4_block)
from .settings import solverSetting


def pack_interval(Yflat, mech, zNo, t_end):
    """One entry of runM2's dataPack (pbReactor.py:729-753)."""
    Y = np.reshape(Yflat, (mech.V, zNo))
    C = Y[:mech.S].copy()
    T = np.array([Y[mech.S]])
    return {"successStatus": True, "dataTime": t_end, "dataYCons": C, "dataYTemp": T,
            "dataYs": np.concatenate((C/np.sum(C, axis=0), T), axis=0)}


def result_lists(packs, ReLe, zNo, opTSpan):
    """The dict runM2 returns (pbReactor.py:806-840): after its loop over the variables the names
    XYList/dataList hold the LAST variable's (temperature's) series, one per output time, built by
    plots2DSetXYList / plots2DSetDataList (PyREMOT/library/plot.py:85-115)."""
    dataXs = np.linspace(0, ReLe, zNo)
    series = np.array([p["dataYs"][-1] for p in packs])
    XYList = [[dataXs, row] for row in series]
    dataList = [{"x": XYList[t][0], "y": XYList[t][1], "leg": "Temperature at t=" + str(opTSpan[t + 1])}
                for t in range(len(packs))]
    return {"XYList": XYList, "dataList": dataList}


def run_m2(modelInput, members_inputs=None):
    start = timer()
    cfg = modelInput['solver-config']
    ivp = resolve_ivp(cfg['ivp'])
    zNo = int(cfg.get('zNo', solverSetting['S2']['zNo']))           # pbReactor.py:625
    tNo = int(cfg.get('tNo', solverSetting['S2']['tNo']))           # :694
    quiet = bool(cfg.get('quiet', False))
    if cfg.get('dtype', 'fp64') not in ('fp64', 'float64'):
        raise ValueError("model M2 is built in fp64 only")
    opT = modelInput['operating-conditions']['period']
    inputs = list(members_inputs) if members_inputs else [modelInput]
    mech = mechanism_for(modelInput, inputs, cfg)
    from .ensemble import active_ranks, guarded
    sync = active_ranks(len(inputs)) if members_inputs else None       # one rank of a torchrun job?
    block, npt = cfg.get('block'), cfg.get('nodes-per-thread')
    if ivp == "hip-ros4" and block is None:
        block, npt = ros4_block(mech.V, zNo, quad=False), 1
    defines = {}
    if ivp == "hip-rk45" and block is None:
        block, npt, defines = rk45_geometry(mech.V, zNo, E=len(inputs) if sync is None else max(sync.counts))
    if ivp == "hip-auto":               # the reference's LSODA: automatic stiff / non-stiff choice (n2.AutoStepper)
        dev, named_local, IV = open_auto(mech, inputs, zNo, plan.member_constants_m2, plan.initial_state_m2, sync,
                                         False, None, block, npt)
    else:
        dev, named_local, IV = open_members(mech, inputs, zNo, plan.member_constants_m2, plan.initial_state_m2, sync,
                                            block=block, npt=npt, defines=defines,
                                            features=("ros4",) if ivp == "hip-ros4" else ())
    packer = sync is None or sync.rank == 0          # rank 0 (or the only process) packs every member
    n_pack = len(inputs) if packer else 0
    opTSpan = np.linspace(0, opT, tNo + 1)                          # :695
    try:
        y = guarded(sync, dev.to_device, IV)
        packs = [[] for _ in range(n_pack)]

        def on_interval(i, t1, Yh):
            Yg = Yh if sync is None else sync.gather(Yh)
            if Yg is not None:
                for e in range(n_pack):
                    packs[e].append(pack_interval(Yg[e], mech, zNo, t1))
        stats = integrate_intervals(dev, y, cfg, ivp, opTSpan, len(named_local), zNo, quiet or not packer,
                                    on_interval, sync)
    finally:
        dev.close()
    ReLe = modelInput['reactor']['ReLe']
    res = result_lists(packs[0] if packs else [], ReLe, zNo, opTSpan)
    res["dataPack"] = packs[0] if packs else []
    res["computation-time"] = np.round(timer() - start, ROUND_FUN_ACCURACY)
    res["device-stats"] = stats
    if members_inputs:
        res["ensemble"] = [dict(result_lists(p, mi['reactor']['ReLe'], zNo, opTSpan), dataPack=p)
                           for p, mi in zip(packs, inputs)] if packer else None
    if sync is not None:
        res["ensemble-shard"] = {"rank": sync.rank, "world": sync.world, "members": [sync.lo, sync.hi]}
    return res
