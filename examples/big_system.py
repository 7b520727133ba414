#!/usr/bin/env python3
"""Systems far beyond the benchmark size on one GPU (288 GB of HBM3E): N = 2.7e7 and 6.4e7 3-D LJ from the jittered lattice,
m = 4: twelve sweeps, running energy == recomputation bit for bit.  One MI355X: 1.59 / 1.60e9 attempted moves/s."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice
for side in (300, 400):
    pos, sp, rho = lj_lattice(side, jitter=0.05, seed=42)
    N = len(pos)
    t0 = time.perf_counter()
    s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
    s.rounds = 4
    e0, f0 = s.total_energy()
    t1 = time.perf_counter()
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
    pmc.mc_sweep(s, pool, nsweeps=4)
    t2 = time.perf_counter(); st = pmc.mc_sweep(s, pool, nsweeps=8); el = time.perf_counter() - t2
    ok = s.total_energy()[1] == st.energy_fix
    out = s.position
    print(f"N={N}: create+energy {t1-t0:.1f}s, {N*8/el:.3e} moves/s, acc {st.accepted[0]/st.attempted[0]:.3f}, running==recomputed {ok}, e/N {st.energy/N:.4f}, max move {np.abs(out-pos).max():.3f}", flush=True)
    del s, pos, out
