"""The reference's published benchmark system as published (benchmark/particles_benchmarks.jl:10-28: N = 216 3-D 50:50 BHHP soft
spheres n = 12, rho = 0.5342, T = 0.2, sigma_move = 0.1, EmptyList) on the reference-order chain: engine 5 (state in LDS, one
resolver and six evaluators, evaluations repaired term by term), engine 3 (state in LDS, one step after the other) and the
literal loop (engine 1); one chain and the chains of a parallel = true run.  One JSON line per measurement.

    python examples/empty_rate.py [sweeps [chains ...]]        (defaults: 1000 sweeps; 1 16 256 chains; ENGINES=5,3,1)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import particlesmc_amd as pmc                                       # noqa: E402
from particlesmc_amd.workloads import random_gas                    # noqa: E402


def main():
    nsw = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    counts = [int(a) for a in sys.argv[2:]] or [1, 16, 256]
    pos, _ = random_gas(216, 0.5342, 3, seed=42, min_dist=0.85)
    sp = np.ones(216, dtype=np.int32)
    sp[np.random.default_rng(42).permutation(216)[:108]] = 2
    N = 216
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.1}, 1.0)]

    def chain(k, engine):
        g = pmc.System(pos, sp, 0.5342, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=42, chain=k)
        g.set_sequential_engine(engine, 0)
        return g
    for engine in [int(e) for e in os.environ.get("ENGINES", "5,3,1").split(",")]:
        for n in counts:
            sweeps = nsw if engine == 5 else max(1, nsw // (5 if engine == 3 else 50))
            if n == 1:
                g = chain(0, engine)
                pmc.mc_sweep(g, pool, nsweeps=5, mode="sequential")
                t0 = time.perf_counter()
                st = pmc.mc_sweep(g, pool, nsweeps=sweeps, mode="sequential")
                dt = time.perf_counter() - t0
                row = {"engine": engine, "chains": 1, "sweeps": sweeps, "moves_per_s": sweeps * N / dt, "us_per_step": 1e6 * dt / (sweeps * N),
                       "kernel_us_per_step": 1e3 * st.sweep_kernel_ms / (sweeps * N), "acceptance": float(st.accepted[0]) / max(1, st.attempted[0])}
                if engine == 5:
                    polls, again, patched, steps = g.verlet_info()
                    row.update({"steps_repaired": patched / max(1, steps), "steps_evaluated_again": again / max(1, steps), "resolver_polls": polls})
            else:
                if engine == 1 and n > 16:
                    continue
                gs = [chain(k, engine) for k in range(n)]
                pmc.mc_sweep_many(gs, pool, nsweeps=2, mode="sequential")
                t0 = time.perf_counter()
                sts = pmc.mc_sweep_many(gs, pool, nsweeps=sweeps, mode="sequential")
                dt = time.perf_counter() - t0
                row = {"engine": engine, "chains": n, "sweeps": sweeps, "moves_per_s": n * sweeps * N / dt,
                       "kernel_moves_per_s": n * sweeps * N / (sts[0].sweep_kernel_ms * 1e-3)}
                del gs
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
