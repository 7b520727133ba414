"""Not a pytest file (the oracle needs 2-3 minutes): one sweep of the reference-order chain at N = n^3 (default 200^3 = 8 * 10^6, the
maximum-size case) on the speculative windows against the oracle -- accept count, positions, system.energy, cs, head / list.
usage (MI355X box): python tests/big_seq_check.py [n]     round 4: N = 8 * 10^6 bit-equal, 0.27 s against 140 s"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import particlesmc_amd as pmc
from oracle import oracle as O
from inputs import lj_lattice
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pos, sp, rho = lj_lattice(n, jitter=0.05, seed=5, rho=1.0)
N = len(pos)
g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=11)
pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
t0 = time.time(); st = pmc.mc_sweep(g, pool, nsweeps=1, mode="sequential"); t1 = time.time()
print("gpu", N, "steps", t1 - t0, "s", N / (t1 - t0), "moves/s", g.sequential_info(), flush=True)
o = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
assert g is not None
t0 = time.time(); att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 11, 0, 0, N); t1 = time.time()
print("oracle", t1 - t0, "s", flush=True)
assert st.accepted[0] == acc[0], (st.accepted[0], acc[0])
assert np.array_equal(g.position, o.positions()), "positions"
assert g.energy[0] == o.energy, (g.energy[0], o.energy)
assert np.array_equal(g.get_cells(), o.cells())
hg, lg = g.get_linked_list(); ho, lo = o.linked_list()
assert np.array_equal(hg, ho) and np.array_equal(lg, lo)
print("N =", N, "sequential chain bit-equal to the oracle: accepted", int(acc[0]))
