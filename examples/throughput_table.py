#!/usr/bin/env python3
"""Throughput of the checkerboard engine over the BASELINE configurations and other sizes
(the numbers quoted in DESIGN.md section 5).  One MI355X, synthetic inputs of particlesmc_amd/workloads.py.

  python examples/throughput_table.py            # all rows
  python examples/throughput_table.py lj ka      # selected groups: lj, ka, ss2d, many, equil, images
  PMC_TEAM=2 python examples/throughput_table.py lj   # with a fixed team width (every row carries the team in use)
  PMC_ROUNDS=4 python examples/throughput_table.py    # with m = 4 sweeps per grid shift (every row carries `rounds`)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import particlesmc_amd as pmc  # noqa: E402
from particlesmc_amd.workloads import binary_lattice, lj_lattice  # noqa: E402


def disp(sigma=0.05, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def rate(system, pool, sweeps, warm):
    pmc.mc_sweep(system, pool, nsweeps=warm)
    t0 = time.perf_counter()
    st = pmc.mc_sweep(system, pool, nsweeps=sweeps)
    dt = time.perf_counter() - t0
    _, fix = system.total_energy()
    assert fix == st.energy_fix, "running energy != recomputed energy"
    att = float(np.sum(st.attempted))
    return {"moves_per_s": att / dt, "acceptance": [float(a) / max(1.0, float(b)) for a, b in zip(st.accepted, st.attempted) if b > 0],
            "e_per_particle": st.energy / system.N}


def main():
    groups = set(sys.argv[1:]) or {"lj", "ka", "ss2d", "many", "equil", "images"}
    team = os.environ.get("PMC_TEAM", "auto")
    rounds = int(os.environ.get("PMC_ROUNDS", "1"))      # schedule parameter m (pmc_set_rounds); every row carries it
    rows = []

    def emit(row):
        row["rounds"] = rounds
        rows.append(row)
        print(json.dumps(row), flush=True)

    if "lj" in groups:
        for n, sw in ((16, 400), (24, 400), (32, 400), (48, 300), (64, 200), (100, 40), (128, 20), (200, 10)):
            pos, sp, rho = lj_lattice(n, jitter=0.05, seed=42)
            s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
            s.total_energy()
            emit({"row": f"LJ rho=1 T=1 N={len(pos)}", "team": team, **rate(s, [disp()], sw, max(2, sw // 10))})
    if "equil" in groups:
        pos, sp, rho = lj_lattice(100, jitter=0.05, seed=42)
        s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
        s.total_energy()
        emit({"row": "LJ N=1000000 after 2000 warm-up sweeps", **rate(s, [disp()], 100, 2000)})
    if "images" in groups:
        # every particle shifted by random whole boxes: positions are never wrapped (src/moves.jl:46-48), so this is what a
        # long run of a liquid looks like to the kernels -- every cell visit takes the general minimum-image path
        pos, sp, rho = lj_lattice(100, jitter=0.05, seed=42)
        L = (len(pos) / rho) ** (1.0 / 3.0)
        for shifted in (False, True):
            p = pos + (np.random.default_rng(7).integers(-3, 4, size=pos.shape) * L if shifted else 0.0)
            s = pmc.System(p, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
            s.total_energy()
            emit({"row": "LJ N=1000000 " + ("shifted by random whole boxes" if shifted else "all in the primary box"), "team": team,
                         **rate(s, [disp()], 40, 5)})
    if "ka" in groups:
        for N, sw in ((100000, 100), (1000000, 20)):
            for swaps in (True, False):
                pos, sp, rho = binary_lattice(N, frac_b=0.2, rho=1.2, jitter=0.05, seed=42)
                s = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=42)
                s.total_energy()
                pool = [disp(p=0.9), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1)] if swaps else [disp()]
                emit({"row": f"Kob-Andersen 80:20 rho=1.2 N={N} " + ("90% Displacement + 10% DiscreteSwap" if swaps else "Displacement only"),
                             **rate(s, pool, sw, max(2, sw // 10))})
    if "ss2d" in groups:
        n = 256
        N, rho = n * n, 0.8
        L = (N / rho) ** 0.5
        g = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
        pos = (g + 0.5) * (L / n) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
        sp = np.ones(N, dtype=np.int32)
        sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
        mm = [[pmc.SoftSpheres(1.0, sg, 14) for sg in row] for row in ([1.0, 1.2], [1.2, 1.4])]
        s = pmc.System(pos, sp, rho, 1.0, mm, seed=5)
        s.total_energy()
        emit({"row": "2-D soft spheres n=14 binary N=65536", **rate(s, [disp(0.065)], 400, 40)})
    if "many" in groups:
        pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
        chains = [pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42, chain=c) for c in range(8)]
        for c in chains:
            c.total_energy()
        pool = [disp()]
        pmc.mc_sweep_many(chains, pool, nsweeps=50)
        t0 = time.perf_counter()
        pmc.mc_sweep_many(chains, pool, nsweeps=500)
        dt = time.perf_counter() - t0
        emit({"row": "8 chains of N=4096 in shared launches (pmc_sweep_many)", "moves_per_s": 8 * 4096 * 500 / dt})
        t0 = time.perf_counter()
        for c in chains:
            pmc.mc_sweep(c, pool, nsweeps=500)
        dt = time.perf_counter() - t0
        emit({"row": "the same 8 chains one after the other", "moves_per_s": 8 * 4096 * 500 / dt})
        # small systems fill the chip by the number of chains, not by their size (parallel = true runs, tempering ladders)
        for side, nch, sw in ((16, 64, 200), (16, 256, 100), (32, 32, 100)):
            pos, sp, rho = lj_lattice(side, jitter=0.05, seed=42)
            chains = [pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42, chain=c) for c in range(nch)]
            for c in chains:
                c.total_energy()
            pmc.mc_sweep_many(chains, pool, nsweeps=max(10, sw // 5))
            t0 = time.perf_counter()
            pmc.mc_sweep_many(chains, pool, nsweeps=sw)
            dt = time.perf_counter() - t0
            emit({"row": f"{nch} chains of N={len(pos)} in shared launches (pmc_sweep_many)", "moves_per_s": nch * len(pos) * sw / dt})
            del chains


if __name__ == "__main__":
    main()
