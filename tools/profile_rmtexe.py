#!/usr/bin/env python3
"""cProfile of the user-visible ensemble call (bench.py rmtexe_ensemble_wall): where do the ~0.2 s go beside the 0.04 s of
device time?  usage: profile_rmtexe.py [profile|outlet]"""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP                       # noqa: E402
from rmt_app_amd import rmtExe             # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "profile"


def call():
    mi = INP.dme_notebook_input(ivp="hip-ros4")
    mi["solver-config"].update({"quiet": True, "zNo": 1024, "tNo": 5, "ensemble-output": mode,
                                "ensemble": {"temperature": list(np.linspace(503.0, 543.0, 64)[:8]),
                                             "pressure": list(np.linspace(3.0e6, 7.0e6, 32))}})
    return rmtExe(mi)


call()                                      # warm: code objects cached, library loaded
pr = cProfile.Profile()
pr.enable()
call()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
