#!/usr/bin/env python3
"""Small ensembles under RK45: one workgroup per reactor vs the reactor cut into 2 / 4 chunks (more CUs busy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "tools", "microbench", "exp_rk45_chain.py")).read().split('print("| mechanism')[0]
exec(compile(src, "exp_rk45_chain.py", "exec"))
print("| mechanism | N | E | kernel | steps | ms | accepted node-steps/s | vs first | flags |")
print("|---|---|---|---|---|---|---|---|---|")
for name, N, E, t1, geos in (("syn12", 512, 64, 0.1, ((256, 2, 2, "reg"), (128, 2, 2, "chain"), (64, 2, 2, "chain"), (128, 1, 2, "chain"))),
                             ("syn12", 512, 16, 0.1, ((256, 2, 2, "reg"), (64, 2, 2, "chain"), (64, 1, 2, "chain"))),
                             ("dme_nb", 1024, 64, 8e-3, ((512, 2, 2, "reg"), (256, 2, 2, "chain"), (128, 2, 2, "chain"), (256, 1, 2, "chain"))),
                             ("dme_nb", 1024, 16, 8e-3, ((512, 2, 2, "reg"), (128, 2, 2, "chain"), (64, 2, 2, "chain"), (64, 1, 2, "chain"))),
                             ("dme_nb", 1024, 128, 8e-3, ((512, 2, 2, "reg"), (256, 2, 2, "chain")))):
    ref = None
    for blk, npt, lds, mode in geos:
        out = run(name, N, E, t1, mode, blk, npt, {"RMT_RK45_LDS": str(lds)}, ref)
        if ref is None: ref = out
