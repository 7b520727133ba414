#!/usr/bin/env python3
"""Long runs of the throughput engine on one MI355X: after every block of sweeps the exact running energy must equal a
from-scratch recomputation bit for bit, and two runs of the same chain must end in bit-identical positions.
N = 10^6 LJ with m = 4 (50 000 sweeps = 5 * 10^10 attempted moves, ~32 s) and m = 1, the same from a configuration shifted by
random whole boxes (periodic images), Kob-Andersen N = 10^6 with swaps, and two team-path sizes.

  python examples/soak.py
"""
import os
import sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice, binary_lattice
def run(N3, m, nblocks, block, ka=False, shifted=False):
    if ka:
        pos, sp, rho = binary_lattice(N3, frac_b=0.2, rho=1.2, jitter=0.05, seed=42)
        model = pmc.KobAndersen()
    else:
        pos, sp, rho = lj_lattice(N3, jitter=0.05, seed=42)
        model = pmc.LennardJones(1.0, 1.0)
    if shifted:
        L = (len(pos) / rho) ** (1.0 / 3.0)
        pos = pos + np.random.default_rng(7).integers(-3, 4, size=pos.shape) * L
    s = pmc.System(pos, sp, rho, 1.0, model, seed=42)
    s.rounds = m
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 0.9 if ka else 1.0)]
    if ka:
        pool.append(pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1))
    t0 = time.time()
    for b in range(nblocks):
        st = pmc.mc_sweep(s, pool, nsweeps=block)
        _, fix = s.total_energy()
        assert fix == st.energy_fix, (b, fix, st.energy_fix)
    return hashlib.md5(s.position.tobytes()).hexdigest(), st.energy_fix, time.time() - t0
for args in ((100, 4, 10, 5000), (100, 1, 4, 2500), (100, 4, 4, 2500, False, True), (1000000, 4, 4, 1000, True), (32, 4, 10, 5000), (48, 2, 4, 5000)):
    a = run(*args); b = run(*args)
    print(args, a, "deterministic" if a[:2] == b[:2] else "DIFFERENT", flush=True)
    assert a[:2] == b[:2]
print("soak ok")
