"""The reference's CI system (test/runtests.jl:22-91: config_0, N = 1290 2-D ternary JBB, VerletList dr = 0.2, Displacement
sigma = 0.05) on the reference-order chain: engine 4 (state in LDS, one resolver and twelve evaluators per chain) and the
literal loop (engine 1), one chain and the chains of a parallel = true run in one launch.  One JSON line per measurement.

    python examples/verlet_rate.py [sweeps [chains ...]]        (defaults: 100 sweeps; 1 16 256 chains)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import particlesmc_amd as pmc                                       # noqa: E402
from particlesmc_amd.io import load_configuration                   # noqa: E402


def main():
    nsw = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    counts = [int(a) for a in sys.argv[2:]] or [1, 16, 256]
    cfg = load_configuration(os.path.join(ROOT, "tests", "golden", "config_0.xyz"))
    pos, sp, box = np.asarray(cfg["position"], dtype=np.float64), np.asarray(cfg["species"], dtype=np.int32), np.asarray(cfg["box"], dtype=np.float64)
    pos = pos - np.floor(pos / box) * box
    rho = len(pos) / float(np.prod(box))
    N = len(pos)
    swaps = os.environ.get("SWAPS") is not None
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 0.2 if swaps else 1.0)]
    if swaps:
        pool.append(pmc.Move(pmc.DiscreteSwap(0, 0, (1, 2), (int((sp == 1).sum()), int((sp == 2).sum())), 0.0), pmc.DoubleUniform(), {}, 0.8))

    def chain(k, engine):
        g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=pmc.VerletList, list_parameters={"dr": 0.2}, seed=10, chain=k)
        g.set_sequential_engine(engine, 0)
        return g
    for engine in (4, 1):
        for n in counts:
            sweeps = nsw if engine == 4 else max(1, nsw // 25)
            if n == 1:
                g = chain(0, engine)
                pmc.mc_sweep(g, pool, nsweeps=1, mode="sequential")
                t0 = time.perf_counter()
                st = pmc.mc_sweep(g, pool, nsweeps=sweeps, mode="sequential")
                dt = time.perf_counter() - t0
                row = {"engine": engine, "chains": 1, "sweeps": sweeps, "moves_per_s": sweeps * N / dt,
                       "kernel_moves_per_s": sweeps * N / (st.sweep_kernel_ms * 1e-3), "us_per_step": 1e6 * dt / (sweeps * N),
                       "acceptance": float(st.accepted[0]) / max(1, st.attempted[0])}
                if engine == 4:
                    polls, alone, rebuilds, steps = g.verlet_info()
                    row.update({"steps_run_alone": alone / max(1, steps), "rebuilds_per_step": rebuilds / max(1, steps), "resolver_polls": polls})
            else:
                gs = [chain(k, engine) for k in range(n)]
                pmc.mc_sweep_many(gs, pool, nsweeps=1, mode="sequential")
                t0 = time.perf_counter()
                sts = pmc.mc_sweep_many(gs, pool, nsweeps=sweeps, mode="sequential")
                dt = time.perf_counter() - t0
                row = {"engine": engine, "chains": n, "sweeps": sweeps, "moves_per_s": n * sweeps * N / dt,
                       "kernel_moves_per_s": n * sweeps * N / (sts[0].sweep_kernel_ms * 1e-3), "launches": int(sts[0].sweep_kernel_launches)}
                del gs
            row["pool"] = "20% Displacement + 80% DiscreteSwap" if swaps else "Displacement"
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
