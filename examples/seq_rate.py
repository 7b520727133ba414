"""Rate of the reference-order chain (PMC_SEQUENTIAL): the literal one-wavefront loop (engine 1) against the speculative
windows (engine 2, pmc_seq.h).  Usage: python examples/seq_rate.py [n ...] (lattice side; N = n^3)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice


def main():
    sides = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16, 32, 100]
    windows = [int(w) for w in os.environ.get("WINDOWS", "0").split(",")]
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
    for n in sides:
        pos, sp, rho = lj_lattice(n, jitter=0.05, seed=42)
        N = len(pos)
        s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
        pmc.mc_sweep(s, pool, nsweeps=20 if N < 200000 else 5, mode="checkerboard")     # melt the lattice first
        for w in windows:
            s.set_sequential_engine(2, w)
            nsw = max(1, int(float(os.environ.get("STEPS", 2e6)) // N))
            pmc.mc_sweep(s, pool, nsweeps=1, mode="sequential")
            t0 = time.perf_counter()
            st = pmc.mc_sweep(s, pool, nsweeps=nsw, mode="sequential")
            dt = time.perf_counter() - t0
            rounds, evals, win, folded = s.sequential_info()
            print(json.dumps({"N": N, "engine": 2, "window": win, "sweeps": nsw, "moves_per_s": nsw * N / dt, "kernel_ms": st.kernel_ms,
                              "rounds": rounds, "steps_per_round": nsw * N / max(rounds, 1), "evals_per_step": evals / (nsw * N),
                              "us_per_round": 1e3 * st.kernel_ms / max(rounds, 1), "acceptance": st.accepted[0] / st.attempted[0]}), flush=True)
        if N <= 40000 and not os.environ.get("NO_ENGINE1"):
            s.set_sequential_engine(1, 0)
            nst = max(1, 20000 // N)
            t0 = time.perf_counter()
            st = pmc.mc_sweep(s, pool, nsweeps=nst, mode="sequential")
            dt = time.perf_counter() - t0
            print(json.dumps({"N": N, "engine": 1, "moves_per_s": nst * N / dt}), flush=True)


if __name__ == "__main__":
    main()
