"""Public API of the device build: the same two functions the reference exports
(PyREMOT/__init__.py:1-3, PyREMOT/rmt.py:21-92) with the same ``modelInput`` schema and the same
result shape ``{"resModel": ..., "comTime": ...}``.

Only the hot path named in BASELINE.json is implemented: ``model == "N2"`` (dynamic homogeneous
packed-bed reactor), plus the "next" rows of SURVEY.md section 8(f): ``"N1"`` (steady state) and ``"M2"``
(the dimensional dynamic model).  ``solver-config.ivp`` selects the device integrator: ``"hip-ros4"`` (stiff
Rosenbrock), ``"hip-rk4"``, ``"hip-rk45"``, ``"AM"`` (the reference's PreCorr3); ``"default"`` - LSODA
in the reference, pbHomoReactor.py:3576, i.e. automatic stiffness detection - maps to ``"hip-auto"`` (explicit
pair while the problem is not stiff, Rosenbrock when it is); SciPy's implicit method names map to ``"hip-ros4"``.  Any other model id raises - the reference silently returns None there
(rmtCore.py:90-127), which is not a behaviour worth mirroring for unsupported models.
"""
import timeit

from . import compdb
from .plan import build_component_list


def rmtExe(modelInput):
    """Check the model input, then run it (PyREMOT/rmt.py:21-80)."""
    try:
        tic = timeit.default_timer()
        modelType = modelInput['model']
        FeCom = modelInput['feed']['components']
        compList = build_component_list(FeCom)
        for c in compList:
            if c not in compdb.componentSymbolList:
                raise Exception("Component database is not up to date!")
        if modelType == "N2":
            from .n2 import run_n2
            ensemble = modelInput['solver-config'].get('ensemble')
            if ensemble is not None:
                from .ensemble import expand_members
                ensemble = expand_members(modelInput, ensemble)
            resModel = run_n2(modelInput, ensemble)
        elif modelType == "N1":
            from .n1 import run_n1
            ensemble = modelInput['solver-config'].get('ensemble')
            if ensemble is not None:
                from .ensemble import expand_members
                ensemble = expand_members(modelInput, ensemble)
            resModel = run_n1(modelInput, ensemble)
        elif modelType == "M2":
            from .m2 import run_m2
            ensemble = modelInput['solver-config'].get('ensemble')
            if ensemble is not None:
                from .ensemble import expand_members
                ensemble = expand_members(modelInput, ensemble)
            resModel = run_m2(modelInput, ensemble)
        else:
            raise NotImplementedError(
                "model %r is outside the MI355X hot path (only 'N2', its steady sibling 'N1' and the "
                "dimensional dynamic model 'M2' are built; SURVEY.md section 8)" % (modelType,))
        tac = timeit.default_timer()
        # the reference's comTime is (timeit.timeit()-timeit.timeit())*1000, i.e. noise
        # (rmt.py:28,67,70); here it is the real wall time in ms.
        return {"resModel": resModel, "comTime": (tac - tic)*1000}
    except Exception as e:
        print(e)
        raise


def rmtCom():
    """Components available in the database (PyREMOT/rmt.py:83-92)."""
    return ','.join(compdb.componentSymbolList)
