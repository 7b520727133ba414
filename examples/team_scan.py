#!/usr/bin/env python3
"""Which team width (wavefronts per cell visit) is fastest for which system size: the measurement behind `team_rule` in
particlesmc_amd/csrc/pmc_hip.hip.  Every (system, width, m) runs in a child process with PMC_TEAM / PMC_ROUNDS set.

  python examples/team_scan.py lj      # 3-D LJ rho = 1, N = 32768 ... 729000
  python examples/team_scan.py ka      # Kob-Andersen 80:20 rho = 1.2, with and without 10 % DiscreteSwap
  python examples/team_scan.py ss2d    # 2-D binary r^-14 soft spheres rho = 0.8, N = 4096 ... 2^20

One JSON row per system: moves/s in units of 10^8 for widths 1 / 2 / 4 / 8 (t1 ... t8), `nact` = cells per colour."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {"lj": [(n, 0) for n in (32, 40, 48, 56, 64, 80, 90)],
         "ka": [(n, sw) for sw in (0, 1) for n in (30000, 100000, 300000, 600000)],
         "ss2d": [(n, 0) for n in (64, 128, 256, 512, 1024)]}


def child(kind, size, swaps):
    import numpy as np
    import particlesmc_amd as pmc
    from particlesmc_amd.workloads import binary_lattice, lj_lattice
    sigma = 0.05
    if kind == "lj":
        pos, sp, rho = lj_lattice(size, jitter=0.05, seed=42)
        s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
    elif kind == "ka":
        pos, sp, rho = binary_lattice(size, frac_b=0.2, rho=1.2, jitter=0.05, seed=42)
        s = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=42)
    else:
        N, rho, sigma = size * size, 0.8, 0.065
        L = (N / rho) ** 0.5
        g = np.stack(np.meshgrid(np.arange(size), np.arange(size), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
        pos = (g + 0.5) * (L / size) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
        sp = np.ones(N, dtype=np.int32)
        sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
        mm = [[pmc.SoftSpheres(1.0, sg, 14) for sg in row] for row in ([1.0, 1.2], [1.2, 1.4])]
        s = pmc.System(pos, sp, rho, 1.0, mm, seed=5)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, 0.9 if swaps else 1.0)]
    if swaps:
        pool.append(pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1))
    pmc.mc_sweep(s, pool, nsweeps=200)
    k = max(40, int(4e7 / len(pos)))
    t0 = time.perf_counter()
    pmc.mc_sweep(s, pool, nsweeps=k)
    el = time.perf_counter() - t0
    nact = 1
    for v in s.grid()[0]:
        nact *= max(1, int(v) // 2)
    print(json.dumps({"rate": len(pos) * k / el, "nact": nact, "N": len(pos)}))


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    for kind in (sys.argv[1:] or ["lj", "ka", "ss2d"]):
        for m in (1, 4):
            for size, swaps in CASES[kind]:
                row = {"system": kind, "rounds": m, "swaps": swaps}
                for team in (1, 2, 4, 8):
                    env = dict(os.environ, PMC_TEAM=str(team), PMC_ROUNDS=str(m))
                    pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, str(size), str(swaps)], env=env,
                                        capture_output=True, text=True)
                    try:
                        d = json.loads(pr.stdout.strip().split("\n")[-1])
                    except Exception:
                        row[f"t{team}"] = None
                        continue
                    row.update({f"t{team}": round(d["rate"] / 1e8, 3), "nact": d["nact"], "N": d["N"]})
                print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
