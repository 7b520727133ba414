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


TIMED = 4      # the PMC passes run `bench.py --steps 4`: the last 4 launches of the sweep kernel are the timed region


def counters(sub):
    """kernel -> counter -> values per launch; for the sweep kernel only the launches of the timed region (the last TIMED
    dispatches: the run starts from the lattice and equilibrates first, see bench.py)"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    files = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                      # newest run only
        rows = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in rows.items():
            for c, vals in v.items():
                vals.sort()
                out[k][c] = [x for _, x in (vals[-TIMED:] if "k_cb_pass" in k else vals)]
    return out


stats = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
# the same run's per-dispatch trace, averaged over the timed region only (the last 20 launches of `bench.py --steps 20`)
traces = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
if traces:
    d = sorted((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
               for r in csv.DictReader(open(traces[-1])) if "k_cb_pass" in r["Kernel_Name"])
    timed = [x for _, x in d[-20:]]
    json.dump({"kernel": "k_cb_pass", "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu",
               "launches_in_run": len(d), "all_launches_avg_ms": sum(x for _, x in d) / max(1, len(d)),
               "timed_region_launches": len(timed), "timed_region_avg_ms": sum(timed) / max(1, len(timed)),
               "note": "the --stats summary averages all launches (lattice start + 1000 equilibration sweeps + timed steps); the timed "
                       "region is the last 20 dispatches of the same trace"},
              open(os.path.join(ROOT, "profiles", f"{tag}_timed_region.json"), "w"), indent=1)
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
            "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 1 --equilibrate 600 --no-cpu`, "
                      "the 4 timed launches, builder's lease)",
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
for src, dst in (("bench_full.json", f"{tag}_bench.json"), ("bench_equilibrated.json", f"{tag}_bench_equilibrated.json"),
                 ("bench_lattice.json", f"{tag}_bench_lattice.json")):
    bj = os.path.join(SRC, src)
    if os.path.exists(bj):
        line = [l for l in open(bj) if l.startswith('{"metric"')]
        if line:
            json.dump(json.loads(line[0]), open(os.path.join(ROOT, "profiles", dst), "w"), indent=1)
print("wrote", tag)
