#!/usr/bin/env python3
"""Instruction mix of a kernel in the hipRTC-built code object (what actually runs).
usage: isa_stats.py [mechanism] [block] [npt] [lds] [kernel] [NAME=VAL ...]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP
from rmt_app_amd import plan, hipbind
name = sys.argv[1] if len(sys.argv) > 1 else "dme_nb"
block, npt = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 512), (3, 2)))
lds = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].lstrip("-").isdigit() else None
kern = sys.argv[5] if len(sys.argv) > 5 else "rmt_n2_rk4_reg"
defines = dict(a.split("=", 1) for a in sys.argv[6:])
copt = defines.pop("COPT", "")
mech = plan.Mechanism(INP.m2_dme_input() if name == "m2" else INP.ALL_N2_INPUTS[name]())
blob, _ = hipbind.compile_source(mech.source(hipbind.kernel_template(), False, block, npt, lds, defines), "gfx950", copt)
f = tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False); f.write(blob); f.close()
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout
i = out.index("<%s>:" % kern)
m = re.search(r"\n[0-9a-f]+ <[A-Za-z0-9_]+>:", out[i + 20:])
body = out[i:i + 20 + m.start()] if m else out[i:]
ins = [l.split("\t")[1].split()[0] for l in body.split("\n") if "\t" in l and len(l.split("\t")) > 1 and l.split("\t")[1].strip()]
c = collections.Counter(ins)
f64 = sum(v for k, v in c.items() if "f64" in k)
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print("%s %s %dx%d lds%s %s: total %d valu %d f64 %d (per node-step %.0f) readlane+writelane %d cndmask %d s_mov %d ds %d scratch %d" % (
    name, kern, block, npt, lds, defines, len(ins), valu, f64, f64/npt, c["v_readlane_b32"] + c["v_writelane_b32"],
    c["v_cndmask_b32_e64"] + c["v_cndmask_b32_e32"], c["s_mov_b32"], sum(v for k, v in c.items() if k.startswith("ds_")),
    sum(v for k, v in c.items() if k.startswith("scratch"))))
meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
blk = meta[meta.index(".name:           %s" % kern) - 400: meta.index(".name:           %s" % kern) + 400]
print("  ", " ".join(x.strip() for x in re.findall(r"\.(?:vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+\d+", blk)))
os.unlink(f.name)
