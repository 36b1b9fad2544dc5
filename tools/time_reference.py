#!/usr/bin/env python3
"""Time the REFERENCE's own right-hand side (PyREMOT modelEquationN2, docs/pbHomoReactor.py:3706) on this
machine's CPU and write profiles/reference_cpu.json.

Build container only (the reference does not travel to the GPU box; bench.py reads the JSON):

    PYTHONPATH=/root/reference MPLBACKEND=Agg python3 tools/time_reference.py

For N in (20, 100, 1024): the DME TEST2.ipynb case is set up by the reference's rmtExe up to its first
solve_ivp call (tools/make_golden.py capture), then modelEquationN2 is called >= 8 times on the initial
state and timed.  "rk4_equiv_node_steps_per_s" = N / (4 * seconds per RHS call): what an explicit RK4
built on that RHS would deliver, the figure bench.py quotes beside its own."""
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import inputs as INP  # noqa: E402
import make_golden as MG  # noqa: E402  (imports the reference)


def main():
    out = {"what": "PyREMOT PackedBedHomoReactorClass.modelEquationN2 (docs/pbHomoReactor.py:3706), DME "
                   "TEST2.ipynb case, one call = one RHS evaluation of all N nodes",
           "machine": {"cpu": platform.processor() or platform.machine(), "cores_used": 1,
                       "python": platform.python_version(), "numpy": np.__version__},
           "date": time.strftime("%Y-%m-%d"), "cases": {}}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                out["machine"]["cpu"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    for N, calls in ((20, 40), (100, 16), (1024, 8)):
        IV, params = MG.capture(INP.dme_notebook_input(), N)
        MG.rhs(params, IV)                                   # warm-up (imports, caches)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            MG.rhs(params, IV)
            ts.append(time.perf_counter() - t0)
        per = float(np.median(ts))
        out["cases"][str(N)] = {"calls": calls, "s_per_rhs_median": per, "s_per_rhs_min": float(min(ts)),
                                "ms_per_node_per_rhs": 1e3*per/N,
                                "node_rhs_per_s": N/per, "rk4_equiv_node_steps_per_s": N/(4*per)}
        print("N=%5d: %.4f s per RHS call (%.3f ms/node) -> %.1f RK4-equivalent node-steps/s"
              % (N, per, 1e3*per/N, N/(4*per)), flush=True)
    with open(os.path.join(ROOT, "profiles", "reference_cpu.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
