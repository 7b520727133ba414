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
SEQ_ONLY = "--seq-only" in sys.argv
OUT = os.environ.get("PMC_PROFILES_OUT") or os.path.join(ROOT, "profiles")


TIMED = 4      # the PMC passes run `bench.py --steps 16 --rounds 4 --only-headline` (m = 1 pass: `--steps 4 --rounds 1`): the last 4
               # launches of the sweep kernel are the timed region
ROUNDS = 4     # schedule parameter m of the headline: one launch = one pass = ROUNDS sweeps


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


if not SEQ_ONLY:
    stats = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    # the same run's per-dispatch trace, averaged over the timed region only: the last 200 / m launches of
    # `bench.py --steps 200 --warmup 20 --only-headline` (m = 4, the headline) and of the same with `--rounds 1`
    region = {}
    for sub, m in (("trace", ROUNDS), ("trace_m1", 1)):
        traces = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
        if traces:
            d = sorted((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
                       for r in csv.DictReader(open(traces[-1])) if "k_cb_pass" in r["Kernel_Name"])
            timed = [x for _, x in d[-(200 // m):]]
            region[f"rounds_{m}"] = {
                "kernel": "k_cb_pass", "command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --rounds {m} --only-headline --no-cpu",
                "launches_in_run": len(d), "all_launches_avg_ms": sum(x for _, x in d) / max(1, len(d)),
                "timed_region_launches": len(timed), "timed_region_avg_ms": sum(timed) / max(1, len(timed)),
                "sweeps_per_launch": m, "timed_region_ms_per_sweep": sum(timed) / max(1, len(timed)) / m,
                "note": "the --stats summary averages all launches (1000 equilibration sweeps at m = 1 + warm-up + timed steps); the timed "
                        "region is the last 200 / m dispatches of the same trace"}
        if sub == "trace_m1":
            st1 = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
            if st1:
                shutil.copy(st1[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_m1.csv"))
    if region:
        json.dump(region, open(os.path.join(ROOT, "profiles", f"{tag}_timed_region.json"), "w"), indent=1)
    summary = {}
    for sub in ("sq_m1",):
        for k, v in counters(sub).items():
            for c, vals in v.items():
                summary.setdefault(k + " [m=1]", {})[c] = {"launches": len(vals), "mean": sum(vals) / len(vals), "sweeps_per_launch": 1}
    for sub in ("fetch", "write", "sq", "sq2"):
        for k, v in counters(sub).items():
            for c, vals in v.items():
                summary.setdefault(k, {})[c] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
    json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    for k, v in summary.items():
        if k.startswith("void k_cb_pass<3, 0, false") and "[m=1]" not in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v and "SQ_WAVES" in v:
            fetch_kib, write_kib = v["FETCH_SIZE"]["mean"], v["WRITE_SIZE"]["mean"]
            traffic = {
                "kernel": k, "round": tag, "lattice": 100,
                "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc passes of `bench.py --steps 16 --warmup 4 --equilibrate 600 --only-headline --no-cpu`, "
                          "the 4 timed launches of 4 sweeps each, builder's lease)",
                "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                "k_cb_pass_hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
                "moves_per_launch": 1000000 * ROUNDS,
                "SQ_INSTS_VALU_per_launch": v.get("SQ_INSTS_VALU", {}).get("mean"),
                "note": "one launch = one pass of the N = 10^6 bench system (all 8 colours, 4 sweeps = 4 * 10^6 attempted moves); "
                        "2 x FETCH_SIZE + WRITE_SIZE; the factor 2 is calibrated for this access width "
                        "(8 B per lane, profiles/calib/RESULT.md)",
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
    me = os.path.join(ROOT, "gpurun_out", "move_efficiency.json")
    if os.path.exists(me):
        shutil.copy(me, os.path.join(ROOT, "profiles", "move_efficiency.json"))
    for src, dst in (("throughput_table.jsonl", f"{tag}_throughput_table.jsonl"), ("throughput_table_m4.jsonl", f"{tag}_throughput_table_m4.jsonl")):
        tj = os.path.join(ROOT, "gpurun_out", src)
        if os.path.exists(tj):
            with open(os.path.join(ROOT, "profiles", dst), "w") as f:
                f.writelines(l for l in open(tj) if l.startswith("{"))
    print("wrote", tag)


# ---- round 4: the reference-order chain (collect.sh "round 4") -------------------------------------------------------------------
def _seq_summary():
    out = {}
    for sub, name in (("seq_trace", "engine2_N1e6"), ("seq_small_trace", "engine2_N4096_one_launch_per_round"), ("seq_many_trace", "engine3_1000xN216")):
        st = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if not st:
            continue
        shutil.copy(st[-1], os.path.join(OUT, f"{tag}_seq_{name}_kernel_stats.csv"))
        rows = [r for r in csv.DictReader(open(st[-1])) if "k_sq" in r["Name"]]
        out[name] = {"kernels": {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                            "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in rows}}
        log = os.path.join(SRC, {"seq_trace": "seq_rate.log", "seq_small_trace": "seq_small.log"}.get(sub, "seq_many.log"))
        if os.path.exists(log):
            lines = [json.loads(l) for l in open(log) if l.startswith("{")]
            if lines:
                out[name]["run"] = lines[-1]
    pmc = {}
    for sub in ("seq_sq", "seq_many_sq", "seq_fetch", "seq_write"):
        for k, v in counters(sub).items():
            if "k_sq" not in k:
                continue
            kk = sub + ":" + k.split("(")[0]
            pmc.setdefault(kk, {})
            for c, vals in v.items():
                pmc[kk][c] = {"mean_per_launch": sum(vals) / max(1, len(vals)), "launches": len(vals), "sum": sum(vals)}
    if pmc:
        out["pmc"] = pmc
    # the team-of-8 instance of k_cb_pass on BASELINE config 2 (N = 32768): the last 40 launches are the timed sweeps
    teams = {}
    for k, v in counters("teams_sq").items():
        if "k_cb_pass" in k:
            teams["kernel"] = k
            for c, vals in v.items():
                vals = vals[-40:]
                teams[c] = {"mean_per_launch": sum(vals) / max(1, len(vals)), "launches": len(vals)}
    st = sorted(glob.glob(os.path.join(SRC, "teams_trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        for r in csv.DictReader(open(st[-1])):
            if "k_cb_pass" in r["Name"]:
                teams["avg_launch_us_all_launches"] = float(r["AverageNs"]) / 1e3
    if teams:
        if "SQ_INSTS_VALU" in teams:
            teams["valu_wave_instructions_per_move"] = teams["SQ_INSTS_VALU"]["mean_per_launch"] / 32768.0
        json.dump(teams, open(os.path.join(OUT, f"{tag}_pmc_teams.json"), "w"), indent=1)
    # round 5: the big-neighbourhood instance of k_cb_pass (published mixture, N = 8000) and the single-cell box (N = 1000 as published):
    # the last 40 launches are the timed sweeps
    for w, nmoves in (("big", 8000.0), ("single", 1000.0)):
        rec = {}
        for k, v in counters(w + "_sq").items():
            if "k_cb_pass" in k:
                rec["kernel"] = k
                for c, vals in v.items():
                    vals = vals[-40:]
                    rec[c] = {"mean_per_launch": sum(vals) / max(1, len(vals)), "launches": len(vals)}
        st = sorted(glob.glob(os.path.join(SRC, w + "_trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if st:
            shutil.copy(st[-1], os.path.join(OUT, f"{tag}_{w}_kernel_stats.csv"))
            for r in csv.DictReader(open(st[-1])):
                if "k_cb_pass" in r["Name"]:
                    rec["avg_launch_us_all_launches"] = float(r["AverageNs"]) / 1e3
        log = os.path.join(SRC, w + "_run.log")
        if os.path.exists(log):
            rec["run"] = [l.strip() for l in open(log) if l.startswith("{")][-1:]
        if rec:
            if "SQ_INSTS_VALU" in rec:
                rec["valu_wave_instructions_per_move"] = rec["SQ_INSTS_VALU"]["mean_per_launch"] / nmoves
            json.dump(rec, open(os.path.join(OUT, f"{tag}_pmc_{w}.json"), "w"), indent=1)
    # round 6: engine 4 (VerletList chains in LDS, k_sq_verlet) on the reference's CI system: one chain, and 256 chains in one launch
    ver = {}
    for sub, name in (("verlet_trace", "one_chain"), ("verlet_many_trace", "256_chains")):
        st = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if not st:
            continue
        shutil.copy(st[-1], os.path.join(OUT, f"{tag}_verlet_{name}_kernel_stats.csv"))
        rows = [r for r in csv.DictReader(open(st[-1])) if "k_sq_verlet" in r["Name"] or "k_ref" in r["Name"]]
        ver[name] = {"kernels": {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                             "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in rows}}
        log = os.path.join(SRC, sub.replace("_trace", ".log"))
        if os.path.exists(log):
            ver[name]["run"] = [json.loads(l) for l in open(log) if l.startswith("{")]
    for k, v in counters("verlet_sq").items():
        if "k_sq_verlet" in k:
            ver.setdefault("pmc_one_chain", {})["kernel"] = k
            for c, vals in v.items():
                ver["pmc_one_chain"][c] = {"mean_per_launch": sum(vals) / max(1, len(vals)), "launches": len(vals)}
    if ver:
        json.dump(ver, open(os.path.join(OUT, f"{tag}_verlet.json"), "w"), indent=1)
        print(f"wrote profiles/{tag}_verlet.json")
    src = os.path.join(ROOT, "gpurun_out", "verlet_rate_table.jsonl")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(OUT, f"{tag}_verlet_rate_table.jsonl"))
    # round 6: engine 5 (small EmptyList chains in LDS, k_sq_empty) on the reference's published benchmark system
    emp = {}
    for sub, name in (("empty_trace", "one_chain"), ("empty_many_trace", "256_chains")):
        st = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if not st:
            continue
        shutil.copy(st[-1], os.path.join(OUT, f"{tag}_empty_{name}_kernel_stats.csv"))
        rows = [r for r in csv.DictReader(open(st[-1])) if "k_sq_empty" in r["Name"]]
        emp[name] = {"kernels": {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                             "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in rows}}
        log = os.path.join(SRC, sub.replace("_trace", ".log"))
        if os.path.exists(log):
            emp[name]["run"] = [json.loads(l) for l in open(log) if l.startswith("{")]
    for k, v in counters("empty_sq").items():
        if "k_sq_empty" in k:
            emp.setdefault("pmc_one_chain", {})["kernel"] = k
            for c, vals in v.items():
                emp["pmc_one_chain"][c] = {"mean_per_launch": sum(vals) / max(1, len(vals)), "launches": len(vals)}
    if emp:
        json.dump(emp, open(os.path.join(OUT, f"{tag}_empty.json"), "w"), indent=1)
        print(f"wrote profiles/{tag}_empty.json")
    src = os.path.join(ROOT, "gpurun_out", "empty_rate_table.jsonl")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(OUT, f"{tag}_empty_rate_table.jsonl"))
    if out:
        json.dump(out, open(os.path.join(OUT, f"{tag}_seq.json"), "w"), indent=1)
        print(f"wrote profiles/{tag}_seq.json")
    for f, dst in (("seq_rate_table.jsonl", f"{tag}_seq_rate_table.jsonl"), ("seq_many_table.jsonl", f"{tag}_seq_many_table.jsonl")):
        src = os.path.join(ROOT, "gpurun_out", f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(OUT, dst))


_seq_summary()
