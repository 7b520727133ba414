#!/usr/bin/env python3
"""Config 5 of BASELINE.json: 2-D soft spheres (r^-14, BHHP size ratios) with parallel tempering,
one replica per GPU, the exchange step over torch.distributed (RCCL with nccl, gloo when ranks
share a GPU).

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/tempering.py [--rccl]
  python examples/tempering.py --local 8            # 8 replicas on one GPU, no process group

The reference has no replica exchange (README.md:33), so there is nothing to compare against; the
script reports attempted moves/s over all replicas, the exchange acceptance and <E>/N per temperature.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

RHO = 0.5      # test/ss142d.jl:16


def make_system(pmc, n, rho, T, seed, chain, device):
    N = n * n
    L = (N / rho) ** 0.5
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    pos = (g + 0.5) * (L / n) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, s, 14) for s in row] for row in ([1.0, 1.2], [1.2, 1.4])]   # BHHP sizes, n = 14
    s = pmc.System(pos, sp, rho, T, mm, seed=seed, chain=chain, device=device)
    s.total_energy()
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=256, help="N = side^2 particles per replica")
    ap.add_argument("--sweeps", type=int, default=400)
    ap.add_argument("--exchange-every", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=1, help="schedule parameter m (pmc_set_rounds); exchange-every should be a multiple of it")
    ap.add_argument("--local", type=int, default=0, help="run this many replicas in one process on one GPU")
    ap.add_argument("--rccl", action="store_true",
                    help="one replica per rank: exchange through the library's own RCCL entry point (pmc_replica_exchange_rccl, "
                         "the path a Julia host takes) instead of torch.distributed's gather")
    ap.add_argument("--tmin", type=float, default=1.0)
    ap.add_argument("--tmax", type=float, default=1.02,
                    help="with N = 65536 adjacent temperatures must differ by ~1/sqrt(N) for exchanges to be accepted; "
                         "BASELINE config 5's 1.0 -> 2.0 ladder gives acceptance 0 at this size")
    args = ap.parse_args()
    import particlesmc_amd as pmc
    from particlesmc_amd import parallel, tempering
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.065}, 1.0)]
    if args.local:
        n = args.local
        # (the replicas share every launch -- pmc_sweep_many -- and the launch picks its wavefronts per cell from the cells of ALL
        # of them: nothing to set here)
        ladder = np.geomspace(args.tmin, args.tmax, n)
        systems = [make_system(pmc, args.side, RHO, ladder[r], 5, r, 0) for r in range(n)]
        for s_ in systems:
            s_.rounds = args.rounds
        t0 = time.perf_counter()
        label, acc, att, traces = tempering.run_tempering_local(systems, pool, ladder, args.sweeps, args.exchange_every, seed=9)
        dt = time.perf_counter() - t0
        byT = {k: [] for k in range(n)}
        for tr in traces:
            for lab, e in tr[len(tr) // 2:]:
                byT[lab].append(e / systems[0].N)
        print(json.dumps({"replicas": n, "mode": "local", "moves_per_s": n * systems[0].N * args.sweeps / dt,
                          "exchange_acceptance": acc / max(1, att), "ladder": list(ladder),
                          "energy_per_particle": [float(np.mean(byT[k])) for k in range(n)]}))
        return
    import torch
    ndev = torch.cuda.device_count()
    rank, local_rank, world = parallel.env_ranks()
    dist, rank, local_rank, world = parallel.init("nccl" if ndev >= world else "gloo")
    dev = local_rank % max(1, ndev)
    tdev = f"cuda:{dev}" if (dist is not None and ndev >= world) else "cpu"
    ladder = np.geomspace(args.tmin, args.tmax, world)
    system = make_system(pmc, args.side, RHO, ladder[rank], 5, rank, dev)
    system.rounds = args.rounds
    comm = tempering.rccl_comm_from_env(dev) if args.rccl else None
    t0 = time.perf_counter()
    rex, trace = tempering.run_tempering(system, pool, ladder, args.sweeps, args.exchange_every, seed=9, rank=rank,
                                         world=world, device=tdev, comm=comm)
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dist, tdev)
    if rank == 0:
        print(json.dumps({"replicas": world, "mode": "one per rank", "moves_per_s": world * system.N * args.sweeps / dt,
                          "exchange_acceptance": rex.accepted / max(1, rex.attempted), "ladder": list(ladder),
                          "final_labels": [int(v) for v in rex.label]}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
