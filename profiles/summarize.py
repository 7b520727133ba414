#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/prof/ into the tracked summaries in profiles/.

Usage (after the gpurun call in profiles/README.md):  python profiles/summarize.py r01
Writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc.json and profiles/traffic.json.
HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are reported in KiB,
collected in separate --pmc passes, and on gfx950 FETCH_SIZE counts 64 B per 128-B request,
so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    files = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                      # newest run only
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


stats = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
summary = {}
for sub in ("fetch", "write", "sq", "sq2"):
    for k, v in counters(sub).items():
        for c, vals in v.items():
            summary.setdefault(k, {})[c] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
for k, v in summary.items():
    if k.startswith("void k_cb_pass<3, 0, false") and "FETCH_SIZE" in v and "WRITE_SIZE" in v and "SQ_WAVES" in v:
        fetch_kib, write_kib = v["FETCH_SIZE"]["mean"], v["WRITE_SIZE"]["mean"]
        traffic = {
            "kernel": k, "round": tag, "lattice": 100,
            "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 1 --no-cpu`, builder's lease)",
            "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
            "k_cb_pass_hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
            "moves_per_launch": 1000000,
            "SQ_INSTS_VALU_per_launch": v.get("SQ_INSTS_VALU", {}).get("mean"),
            "note": "one launch = one fused sweep of the N = 10^6 bench system (all 8 colours); "
                    "2 x FETCH_SIZE + WRITE_SIZE; the factor 2 is calibrated for this access width "
                    "(8 B per lane, profiles/calib/RESULT.md); ~14x below the algorithmic bytes",
        }
        json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
        print(traffic)
for src, dst in (("bench_full.json", f"{tag}_bench.json"), ("bench_equilibrated.json", f"{tag}_bench_equilibrated.json")):
    bj = os.path.join(SRC, src)
    if os.path.exists(bj):
        line = [l for l in open(bj) if l.startswith('{"metric"')]
        if line:
            json.dump(json.loads(line[0]), open(os.path.join(ROOT, "profiles", dst), "w"), indent=1)
print("wrote", tag)
