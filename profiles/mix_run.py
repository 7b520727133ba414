"""The published mixture (examples/lj-mixture: binary LJ, rcut = 4, 90 % Displacement + 10 % DiscreteSwap) for the profile passes of
collect_seq.sh: N = 8000 (4 x 4 x 4 cells, the big-neighbourhood instance of k_cb_pass) or, with `single`, N = 1000 as published
(a single-cell box).  The last 40 launches are the timed sweeps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import particlesmc_amd as pmc

n = 10 if "single" in sys.argv else 20
N, rho, T = n ** 3, 0.8, 1.5
L = (N / rho) ** (1.0 / 3.0)
g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
pos = (g + 0.5) * (L / n)
sp = np.ones(N, dtype=np.int32)
sp[np.random.default_rng(3).permutation(N)[: N // 2]] = 2
EPS, SIG = ((1.0, 1.1523), (1.1523, 1.3702)), ((1.0, 1.0339), (1.0339, 1.0640))
mm = [[pmc.LennardJones(EPS[i][j], SIG[i][j], rcut=4.0, shift_potential=False) for j in range(2)] for i in range(2)]
s = pmc.System(pos, sp, N / L ** 3, T, mm, seed=42)
pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 0.9),
        pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1)]
s.total_energy()
pmc.mc_sweep(s, pool, nsweeps=100)
st = pmc.mc_sweep(s, pool, nsweeps=40)
print({"N": N, "grid": [int(v) for v in s.grid()[0]], "moves_per_s_kernel": N * 40 / (st.sweep_kernel_ms * 1e-3), "launches": st.sweep_kernel_launches,
       "pair_tests_per_move": st.pair_tests / (40.0 * N), "dense_evals_per_move": st.dense_evals / (40.0 * N)})
