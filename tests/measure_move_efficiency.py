#!/usr/bin/env python3
"""The long form of tests/test_gpu_efficiency.py (same code, more schedules): what an attempted move of the checkerboard
schedule with m = 1, 2, 4, 8 sweeps per grid shift is worth in attempted moves of the reference's sequential chain.
Writes gpurun_out/move_efficiency.json; profiles/summarize.py copies it to profiles/move_efficiency.json, which bench.py
replays (with its `source`) beside the MSD ratio it measures itself.

  python tests/measure_move_efficiency.py [chains] [sweeps]       (defaults: 64 chains of 1500 sweeps per leg, ~2 minutes)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import particlesmc_amd as pmc
    from oracle import efficiency as E
    from inputs import lj_lattice
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nsw = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=4242)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
    pmc.mc_sweep(g, pool, nsweeps=3000)
    p0 = g.position.copy()
    seq_series, seq_msd, seq_acc = E.sequential_leg(p0, sp, rho, 1.0, 0.05, nch, nsw, stride=1)
    out = {"system": "N=4096 LJ liquid rho=1 T=1 sigma=0.05, 6 cells of width 2.667 per side",
           "chains_per_leg": nch, "sweeps_per_chain": nsw, "acceptance_sequential": seq_acc, "by_rounds": {},
           "msd_ratio_by_rounds": {}, "tau_ratio_by_rounds": {},
           "source": "profiles/move_efficiency.json (tests/measure_move_efficiency.py on one MI355X + its host cores, builder's lease)"}
    for m in (1, 2, 4, 8):
        cb_series, cb_msd, cb_acc, cb_null = E.gpu_cb_leg(pmc, p0, sp, rho, 1.0, 0.05, nch, nsw - nsw % m, stride=1, rounds=m)
        ss = [s[m - 1::m][: len(cb_series[0])] for s in seq_series]
        r = E.efficiency(ss, seq_msd, cb_series, cb_msd)
        r.update({"acceptance_checkerboard": cb_acc, "null_fraction": cb_null})
        out["by_rounds"][str(m)] = r
        out["msd_ratio_by_rounds"][str(m)] = {"value": r["msd_ratio"], "sigma": r["msd_ratio_sigma"]}
        out["tau_ratio_by_rounds"][str(m)] = {"value": r["tau_ratio_seq_over_cb"], "sigma": r["tau_ratio_sigma"]}
        print(m, r["msd_ratio"], r["msd_ratio_sigma"], r["tau_ratio_seq_over_cb"], r["tau_ratio_sigma"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "move_efficiency.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
