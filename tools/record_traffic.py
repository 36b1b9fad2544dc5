#!/usr/bin/env python3
"""Write the digest-keyed HBM traffic record bench.py reads (profiles/traffic.json) from the rocprofv3
PMC passes of tools/prof_round.sh:   record_traffic.py gpurun_out/<tag> [source-note]

bytes per launch = mean over the TIMED dispatches of the bench kernel of
    FETCH_SIZE * 64 B * corr + WRITE_SIZE * 64 B   (MI355X_MICROARCH.md, HBM/rocprofv3 section:
    FETCH_SIZE / WRITE_SIZE are in KiB-like 1 KB units on this rocprofv3 - see the guide's unit note);
the key is the code-object digest printed in bench_prof.json, so a rebuilt kernel has no entry."""
import csv
import glob
import json
import os
import sys

run = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else run
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = [l for l in open(os.path.join(run, "bench_prof.json")).read().splitlines() if l.startswith("{")][-1]
b = json.loads(line)
kname = b["config"]["kernel"].split()[0]
digest = b["config"]["kernel_digest"]


def mean_counter(sub, name, skip):
    vals = []
    for f in sorted(glob.glob(os.path.join(run, sub, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == kname and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    vals = vals[skip:] if len(vals) > skip else vals
    return sum(vals)/len(vals) if vals else None


fetch = mean_counter("pmc_fetch", "FETCH_SIZE", b["warmup"])
write = mean_counter("pmc_write", "WRITE_SIZE", b["warmup"])
if fetch is None or write is None:
    raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for %s under %s" % (kname, run))
# guide (MI355X_MICROARCH.md, rocprofv3 HBM section): both counters are reported in KiB; on gfx950 FETCH_SIZE
# under-counts 128-B requests by a factor 2 for streaming loads (corrected x2)
bytes_per_launch = 2.0*fetch*1024.0 + write*1024.0
path = os.path.join(ROOT, "profiles", "traffic.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec[digest] = {"kernel": b["config"]["kernel"], "members": b["config"]["members_per_gpu"], "nodes": b["config"]["nodes"],
               "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "bytes_per_launch": bytes_per_launch,
               "formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction of the guide)", "source": note}
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec[digest]))
