"""BASELINE config 2 (N = 32768 LJ liquid, the team-of-8 instance of k_cb_pass) for the counter passes of collect_seq.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice

pos, sp, rho = lj_lattice(32, jitter=0.05, seed=42)
s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
s.rounds = int(os.environ.get("ROUNDS", "1"))
pmc.mc_sweep(s, pool, nsweeps=300)
st = pmc.mc_sweep(s, pool, nsweeps=40)
print({"moves_per_s_kernel": 32768 * 40 / (st.sweep_kernel_ms * 1e-3), "launches": st.sweep_kernel_launches})
