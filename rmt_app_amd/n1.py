"""Steady-state model N1 on the device: the replacement for PackedBedHomoReactorClass.runN1
(PyREMOT/docs/pbHomoReactor.py:2694-3015).  The reference integrates the S+2 unknowns
[c_i, P*, theta] along the dimensionless length with solve_ivp(LSODA) and samples
t_eval = linspace(0, 1, zNo+1) (:2931); here one launch integrates every member of an ensemble
(one reactor per lane) with the RODAS4 scheme of the N2 stiff stepper."""
from timeit import default_timer as timer

import numpy as np

from . import plan
from . import n2 as _n2
from .settings import DEVICE_DEFAULTS, MODEL_SETTING, ROUND_FUN_ACCURACY, solverSetting


def pack_profile(U, named, mech, modelId, elapsed):
    """dataPack entry of runN1 (pbHomoReactor.py:2947-3008; sortResult4, solResultAnalysis.py:191-249).
    U: (nout, S+2) dimensionless states."""
    S = mech.S
    Y = np.asarray(U, dtype=np.float64).T                     # (S+2, nout) like sol.y
    nout = Y.shape[1]
    conc_dl = Y[0:S, :]
    temp_dl = Y[S + 1, :] if not mech.iso else np.repeat(0, nout).reshape(nout)
    # sortResult4 (solResultAnalysis.py:226-231): species i times Cif[i] when MODEL_SETTING['GaMaCoTe0'] != "MAX"
    scale = named.get("SpCoi0_Set", named["Cmax"])
    conc = conc_dl*(np.reshape(scale, (-1, 1)) if np.ndim(scale) else scale)
    Preal = (Y[S]*named["Pf"]).reshape((1, nout))
    mofr = conc/np.sum(conc, axis=0)
    labelList = list(mech.compList) + ["Pressure"]
    parts = [mofr, Preal]
    Treal = None
    if not mech.iso:
        labelList.append("Temperature")
        Treal = (Y[S + 1]*named["Tf"] + named["Tf"]).reshape((1, nout))
        parts.append(Treal)
    dataXs = np.linspace(0, 1, nout)
    return {
        "modelId": modelId, "processType": mech.processType, "successStatus": True,
        "computation-time": elapsed, "dataShape": np.array(dataXs).shape, "labelList": labelList,
        "indexList": [S, S, S + 1], "dataTime": [], "dataXs": dataXs,
        "dataYCons1": conc_dl, "dataYCons2": conc, "dataYTemp1": temp_dl,
        "dataYTemp2": Treal if Treal is not None else np.zeros((1, nout)),
        "dataYs": np.concatenate(parts, axis=0),
    }


def run_n1(modelInput, members_inputs=None):
    """runN1 on the device; returns the reference's list with one dataPack dict (or, for an
    ensemble, a list with one dict per member)."""
    start = timer()
    cfg = modelInput['solver-config']
    displayResult = cfg['display-result'] == "True"
    zNo = int(cfg.get('zNo', solverSetting['N1']['zNo']))
    nout = zNo + 1
    all_inputs = list(members_inputs) if members_inputs else [modelInput]
    mech = _n2.mechanism_for(modelInput, all_inputs, cfg)
    # as one rank of a torch.distributed job: integrate this rank's contiguous block of profiles.  Every
    # rank-local phase (packing, device creation, the launch + status read) runs under ensemble.guarded: a failure
    # on one rank is raised on every rank before the next collective.
    from .ensemble import active_ranks, guarded
    sync = active_ranks(len(all_inputs)) if members_inputs else None
    inputs = all_inputs if sync is None else all_inputs[sync.lo:sync.hi]

    def pack_and_open():
        pairs = [plan.member_constants_n1(mi, mech) for mi in inputs]
        rows1 = np.ascontiguousarray(np.array([r for _, r in pairs]))
        # the handle is an N2 handle (same generated module); its N2 member rows are not used here
        dummy = np.array([plan.member_constants(mi, mech, 64)[1] for mi in inputs])
        # MODEL_SETTING['GaMaCoTe0'] != "MAX": model N1 runs with per-species scaling in the reference (:2819, 3159)
        defs = {"RMT_N1_SCALE_FIX": "1"} if MODEL_SETTING['GaMaCoTe0'] != "MAX" else None
        return pairs, rows1, _n2.device_cls()(mech, dummy, 64, block=64, npt=1, specialize=False, features=("n1",),
                                              defines=defs)
    pairs, rows1, dev = guarded(sync, pack_and_open)
    try:
        def launch():
            out = dev.n1_profile(rows1, nout, float(cfg.get('rtol', DEVICE_DEFAULTS['n1-rtol'])),
                                 float(cfg.get('atol', DEVICE_DEFAULTS['n1-atol'])), float(cfg.get('h0', 1e-6)),
                                 int(cfg.get('max-steps', 10**7)))
            dev.raise_on_flags()
            return dev.rk45_stats(), out
        stats, U = guarded(sync, launch)
    finally:
        dev.close()
    if sync is not None:                    # rank 0 returns every member's profile, the other ranks None
        U = sync.gather(U)
        stats = {k: sync.gather(stats[k]) for k in ("accepted", "rejected")}
        if U is None:
            return None
        inputs = all_inputs
        pairs = [plan.member_constants_n1(mi, mech) for mi in inputs]
    elapsed = np.round(timer() - start, ROUND_FUN_ACCURACY)
    packs = [pack_profile(U[e], pairs[e][0], mech, modelInput['model'], elapsed) for e in range(len(inputs))]
    for p, acc, rej in zip(packs, stats["accepted"], stats["rejected"]):
        p["device-stats"] = {"accepted": int(acc), "rejected": int(rej)}
    if displayResult:
        from .plotting import plotResultsSteadyState
        plotResultsSteadyState([packs[0]])                       # pbHomoReactor.py:3011-3013
    return packs if members_inputs else [packs[0]]
