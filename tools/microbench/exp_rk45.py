#!/usr/bin/env python3
"""rk45_reg variants on the DME 256 x 1024 workload (warm launch, like tools/rk45_bench.py)."""
import os, sys
sys.argv = [sys.argv[0], "quick"]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("rk45_bench_mod", os.path.join(ROOT, "tools", "rk45_bench.py"))
src = open(os.path.join(ROOT, "tools", "rk45_bench.py")).read().split('print("| mechanism')[0]
ns = {"__name__": "rk45_bench_mod", "__file__": os.path.join(ROOT, "tools", "rk45_bench.py")}
exec(compile(src, "rk45_bench.py", "exec"), ns)
run = ns["run"]
t1 = 8e-3
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2)
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2, {"RMT_RK45_TWO_COPIES": "1"})
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2, {"RMT_RK45_LDS": "1"})
run("dme_nb", 1024, 256, t1, 1e-6, "reg", 512, 2, {"RMT_EXP_BITS": "6"})
run("syn12", 512, 64, 0.1, 1e-6, "reg", 256, 2)
run("syn12", 512, 64, 0.1, 1e-6, "reg", 256, 2, {"RMT_RK45_TWO_COPIES": "1"})
