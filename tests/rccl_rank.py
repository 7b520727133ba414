"""One rank of the multi-GPU replica-exchange test (started by tests/test_gpu_multirank.py, one fresh process per GPU):
BASELINE config 5's system on device `rank`, a communicator built through the library's own entry points
(pmc_rccl_unique_id / pmc_rccl_comm_create, id through a file), `nrounds` exchange rounds through
pmc_replica_exchange_rccl.  Writes what it saw to <outdir>/rank<r>.json; the parent compares with the oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    rank, world, outdir, nrounds = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    import particlesmc_amd as pmc
    from particlesmc_amd import tempering
    side, rho = 64, 0.8
    N = side * side
    L = (N / rho) ** 0.5
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    pos = (g + 0.5) * (L / side) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, s, 14) for s in row] for row in ([1.0, 1.2], [1.2, 1.4])]
    ladder = np.geomspace(1.0, 1.12, world)
    system = pmc.System(pos, sp, rho, float(ladder[rank]), mm, seed=5, chain=rank, device=rank)   # one device per rank
    system.total_energy()
    comm = tempering.rccl_comm_from_env(rank, id_file=os.path.join(outdir, "rccl_id"))
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.065}, 1.0)]
    label = np.arange(world, dtype=np.int32)
    rounds = []
    try:
        for rnd in range(nrounds):
            st = pmc.mc_sweep(system, pool, nsweeps=5)
            label, nacc = comm.exchange(system, ladder, 9, rnd, label)
            rounds.append({"energy": st.energy, "label": [int(v) for v in label], "nacc": int(nacc), "temperature": system.temperature})
    finally:
        comm.close()
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "device": rank, "ladder": [float(v) for v in ladder], "rounds": rounds}, f)


if __name__ == "__main__":
    main()
