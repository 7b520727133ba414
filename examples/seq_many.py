"""Many small chains on the reference's sequential chain (PMC_SEQUENTIAL through pmc_sweep_many): the reference's benchmark
system (benchmark/particles_benchmarks.jl:10-28, N = 216 LJ at rho = 1) as `nchains` chains with parallel = true.
Usage: python examples/seq_many.py [side=6] [nchains=1000] [sweeps=200]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    nsw = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    pos, sp, rho = lj_lattice(side, jitter=0.05, seed=42)
    N = len(pos)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
    chains = [pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42, chain=k) for k in range(nch)]
    pmc.mc_sweep_many(chains, pool, nsweeps=5, mode="sequential")
    t0 = time.perf_counter()
    st = pmc.mc_sweep_many(chains, pool, nsweeps=nsw, mode="sequential")
    dt = time.perf_counter() - t0
    acc = sum(s.accepted[0] for s in st) / sum(s.attempted[0] for s in st)
    print(json.dumps({"N": N, "chains": nch, "sweeps": nsw, "moves_per_s": nch * nsw * N / dt, "us_per_step_per_chain": 1e6 * dt / (nsw * N),
                      "kernel_ms": st[0].kernel_ms, "acceptance": acc}), flush=True)


if __name__ == "__main__":
    main()
