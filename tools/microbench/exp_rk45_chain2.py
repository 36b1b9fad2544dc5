#!/usr/bin/env python3
"""rmt_n2_rk45_chain vs rmt_n2_rk45_mem over the shapes of BASELINE configs 3 and 5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "tools", "microbench", "exp_rk45_chain.py")).read().split('print("| mechanism')[0]
exec(compile(src, "exp_rk45_chain.py", "exec"))
from rmt_app_amd.n2 import rk45_geometry, rk45_block
print("| mechanism | N | E | kernel | steps | ms | accepted node-steps/s | vs mem | flags |")
print("|---|---|---|---|---|---|---|---|---|")
for name, N, E, t1 in (("dme_nb", 4096, 1, 4e-3), ("dme_nb", 4096, 64, 4e-3), ("dme_nb", 4096, 256, 4e-3), ("dme_nb", 16384, 1, 2e-3),
                       ("dme_nb", 16384, 16, 2e-3), ("dme_nb", 16384, 64, 2e-3), ("syn12", 1024, 64, 0.05), ("syn12", 1024, 256, 0.05), ("syn12", 4096, 64, 0.05)):
    V = 7 if name == "dme_nb" else 13
    ref = run(name, N, E, t1, "mem", rk45_block(V, N), 1)
    b, n, d = rk45_geometry(V, N)
    run(name, N, E, t1, "chain", b, n, d, ref)
