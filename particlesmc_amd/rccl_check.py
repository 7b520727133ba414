"""One rank of the multi-GPU replica-exchange self-check: `python -m particlesmc_amd.rccl_check RANK WORLD DEVICE OUTDIR NROUNDS`.

A fresh process per GPU builds BASELINE config 5's system (2-D binary r^-14 soft spheres, test/ss142d.jl) on its device, joins
an RCCL communicator through the library's own entry points (pmc_rccl_unique_id / pmc_rccl_comm_create, the 128-byte id handed
over through a file in OUTDIR) and runs NROUNDS rounds of five sweeps + one exchange through pmc_replica_exchange_rccl (one
ncclAllGather of one double per rank; reference: README.md:33 has no counterpart, replica sharding src/IO/IO.jl:319-327).  What
the rank saw goes to OUTDIR/rank<r>.json; the caller compares every rank's labels with the decision function on the energies
all ranks reported (tests/test_gpu_zz_multirank.py: the oracle; bench.py: pmc_pt_decide).

It is a separate process on purpose: bench.py's N > 1 run starts one per rank AFTER its measurements, so that nothing that
happens inside the collective -- a hang, a crash of a communicator that has never run on that node -- can cost the bench line."""
import json
import os
import sys

import numpy as np


def run(rank, world, device, outdir, nrounds, nsweeps=5, side=64):
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    import particlesmc_amd as pmc
    from particlesmc_amd import tempering
    rho = 0.8
    N = side * side
    L = (N / rho) ** 0.5
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    pos = (g + 0.5) * (L / side) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, s, 14) for s in row] for row in ([1.0, 1.2], [1.2, 1.4])]
    ladder = 1.016 ** np.arange(world)        # adjacent temperatures 1.6 % apart whatever the world size (8 ranks: 1.0 ... 1.12): exchanges do happen at N = 4096
    system = pmc.System(pos, sp, rho, float(ladder[rank]), mm, seed=5, chain=rank, device=device)   # one device per rank
    system.total_energy()
    comm = tempering.rccl_comm_from_env(device, id_file=os.path.join(outdir, "rccl_id"))
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.065}, 1.0)]
    label = np.arange(world, dtype=np.int32)
    rounds = []
    try:
        for rnd in range(nrounds):
            st = pmc.mc_sweep(system, pool, nsweeps=nsweeps)
            label, nacc = comm.exchange(system, ladder, 9, rnd, label)
            rounds.append({"energy": st.energy, "label": [int(v) for v in label], "nacc": int(nacc), "temperature": system.temperature})
    finally:
        comm.close()
    with open(os.path.join(outdir, f"rank{rank}.json.tmp"), "w") as f:
        json.dump({"rank": rank, "world": world, "device": device, "ladder": [float(v) for v in ladder], "rounds": rounds}, f)
    os.replace(os.path.join(outdir, f"rank{rank}.json.tmp"), os.path.join(outdir, f"rank{rank}.json"))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 5:
        print(__doc__, file=sys.stderr)
        return 2
    run(int(argv[0]), int(argv[1]), int(argv[2]), argv[3], int(argv[4]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
