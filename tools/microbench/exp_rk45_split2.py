#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "tools", "microbench", "exp_rk45_chain.py")).read().split('print("| mechanism')[0]
exec(compile(src, "exp_rk45_chain.py", "exec"))
print("| mechanism | N | E | kernel | steps | ms | accepted node-steps/s | vs first | flags |")
print("|---|---|---|---|---|---|---|---|---|")
for name, N, E, t1, geos in (("dme_nb", 1024, 128, 8e-3, ((512, 2, 2, "reg"), (512, 1, 2, "chain"), (256, 2, 2, "chain"))),
                             ("dme_nb", 1024, 1, 8e-3, ((512, 2, 2, "reg"), (64, 1, 2, "chain"), (128, 1, 2, "chain"), (256, 1, 2, "chain"))),
                             ("dme_nb", 4096, 8, 4e-3, ((512, 2, 2, "chain"), (256, 1, 2, "chain"), (128, 1, 2, "chain"))),
                             ("dme_nb", 4096, 32, 4e-3, ((512, 2, 2, "chain"), (512, 1, 2, "chain"), (256, 2, 2, "chain"))),
                             ("syn12", 1024, 64, 0.05, ((256, 2, 2, "chain"), (256, 1, 2, "chain"), (128, 2, 2, "chain"))),
                             ("syn12", 512, 128, 0.1, ((256, 2, 2, "reg"), (256, 1, 2, "chain")))):
    ref = None
    for blk, npt, lds, mode in geos:
        out = run(name, N, E, t1, mode, blk, npt, {"RMT_RK45_LDS": str(lds)}, ref)
        if ref is None: ref = out
