#!/usr/bin/env python3
"""256 chains of N = 4096 (LJ, rho = 1, m = 4) in shared launches for 2000 sweeps, in calls of 100-102 sweeps: after every call
the running energy of four chains equals a recomputation bit for bit, and at the end three chains equal the same chains run one at
a time (their own team width, no shared launch).  On one MI355X: 1.2e9 attempted moves/s including the checks."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice
pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
nch = 256
chains = [pmc.System(pos, sp, rho, 0.9 + 0.001 * c, pmc.LennardJones(1.0, 1.0), seed=42, chain=c) for c in range(nch)]
for c in chains: c.rounds = 4
pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
t0 = time.perf_counter(); tot = 0
for call in range(20):
    st = pmc.mc_sweep_many(chains, pool, nsweeps=100 + (call % 3))
    tot += 100 + (call % 3)
    for r in (0, 17, 128, 255):
        assert chains[r].total_energy()[1] == st[r].energy_fix, (call, r)
el = time.perf_counter() - t0
print("sweeps", tot, "moves/s incl. checks %.3e" % (nch * 4096 * tot / el), "e/N", st[0].energy / 4096, st[255].energy / 4096)
# determinism: a second set of the same chains ends in the same state
ch2 = [pmc.System(pos, sp, rho, 0.9 + 0.001 * c, pmc.LennardJones(1.0, 1.0), seed=42, chain=c) for c in (0, 17, 255)]
for c in ch2: c.rounds = 4
for call in range(20):
    for c in ch2: pmc.mc_sweep(c, pool, nsweeps=100 + (call % 3))
for c, r in zip(ch2, (0, 17, 255)):
    assert np.array_equal(c.position, chains[r].position), r
print("shared launches (width 1, learnt capacity) == each chain alone (width 8) after", tot, "sweeps")
