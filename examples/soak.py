"""Soak run: N = 10^6 LJ, 20 000 fused sweeps; every 2000 sweeps the running energy must equal a recomputation exactly."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice
pos, sp, rho = lj_lattice(100, jitter=0.05, seed=42)
s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
s.total_energy()
pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
tot = 0
t0 = time.perf_counter()
for block in range(10):
    st = pmc.mc_sweep(s, pool, nsweeps=2000)
    tot += 2000
    _, fix = s.total_energy()
    print(block, tot, st.energy / len(pos), st.accepted[0] / st.attempted[0], fix == st.energy_fix, flush=True)
    assert fix == st.energy_fix
dt = time.perf_counter() - t0
print("moves", tot * len(pos), "rate incl. checks", tot * len(pos) / dt)
