#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<run>/...) into the small files kept under profiles/.
usage: summarize_prof.py <run_dir> <out_prefix>"""
import csv
import glob
import json
import os
import sys

run, out = sys.argv[1], sys.argv[2]
lines = []
for f in sorted(glob.glob(os.path.join(run, "stats", "*", "*kernel_stats.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith("rmt_")]
    lines.append("## rocprofv3 --kernel-trace --stats  (%s)\n" % os.path.relpath(f, run))
    lines.append("| kernel | calls | total ns | average ns | % of GPU time | min ns | max ns |")
    lines.append("|---|---|---|---|---|---|---|")
    for r in rows:
        lines.append("| %s | %s | %s | %s | %s | %s | %s |" % (
            r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
    lines.append("")
for f in sorted(glob.glob(os.path.join(run, "stats", "*", "*kernel_trace.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("rmt_")]
    lines.append("## per-dispatch (kernel trace)\n")
    lines.append("| kernel | grid | workgroup | LDS B | scratch B | VGPR | AGPR | SGPR | duration ns |")
    lines.append("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        lines.append("| %s | %s | %s | %s | %s | %s | %s | %s | %d |" % (
            r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")),
            r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""), r.get("VGPR_Count", ""),
            r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    lines.append("")
pm = {}
for f in sorted(glob.glob(os.path.join(run, "pmc_*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rmt_"):
            pm.setdefault((r["Kernel_Name"], r["Dispatch_Id"], os.path.basename(os.path.dirname(os.path.dirname(f)))), {})[
                r["Counter_Name"]] = float(r["Counter_Value"])
if pm:
    lines.append("## PMC counters (separate --pmc passes; one row per dispatch)\n")
    for (k, d, p), c in sorted(pm.items()):
        lines.append("* `%s` dispatch %s (%s): %s" % (k, d, p, json.dumps(c, sort_keys=True)))
    lines.append("")
for name in ("bench_default.json", "bench_prof.json"):
    p = os.path.join(run, name)
    if os.path.exists(p):
        txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if txt:
            lines.append("## %s\n\n```json\n%s\n```\n" % (name, txt[-1]))
open(out + ".md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:6000])
