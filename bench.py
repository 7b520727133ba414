#!/usr/bin/env python3
"""bench.py -- attempted Monte-Carlo moves per second of the HIP hot path on MI355X.

Workload (BASELINE.json metric): N = 10^6 3-D Lennard-Jones
particles, rho = 1.0, T = 1.0, rcut = 2.5 (shifted), Displacement moves with
sigma = 0.05: the liquid that a 100^3 simple-cubic lattice jittered by U(-0.05, 0.05) melts into
within --equilibrate (default 1000) untimed sweeps; the first sweeps from the lattice itself, which is
what rounds 1-2 timed, are measured too and reported as "from_lattice" (--equilibrate 0 makes them the value).
One "step" = one sweep = N attempted moves.  With --gpus N every rank runs its own
independent chain (seed 42 + rank) on its own GPU: weak scaling, no data-path
collective (reference: src/IO/IO.jl:319-327, src/ParticlesMC.jl:164,246).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--lattice 100] [--no-cpu]

`--gpus N` with N > 1 started plainly (no WORLD_SIZE in the environment) launches the N ranks
itself through `python -m torch.distributed.run` (one process per GPU, RCCL rendezvous on
127.0.0.1) BEFORE anything in this process touches the GPU, relays rank 0's JSON line and exits
with the ranks' status.  Started by torchrun (WORLD_SIZE set) it is one of those ranks.
Fewer visible devices than ranks is an error (exit 3), not a silent share.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz = half the 157.3 TF FP32 vector peak
FP64_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0   # FP64 wave-instructions/s: one per 4 cycles per SIMD
SIGMA_MOVE = 0.05
TEMPERATURE = 1.0
# algorithmic flop model of the pair loop: one distance test = 3 sub + 3 mul + 2 add + 1 compare,
# one Lennard-Jones pair energy = 1 div + 5 mul + 2 sub + 1 accumulate (src/models.jl:30-34,121-123)
FLOP_PER_PAIR_TEST = 9
FLOP_PER_PAIR_ENERGY = 9


def algorithmic_bytes_per_move(N, d, ncells_ref, multi_species=False):
    """B_alg = 27 n_c (8 d + s) + 2 * 8 d with n_c the mean population of the reference's
    LinkedList cells: 10 972 B at N = 10^6."""
    nbar = N / float(np.prod(ncells_ref))
    return (3 ** d) * nbar * (8 * d + (4 if multi_species else 0)) + 2 * 8 * d


def cpu_baseline(pos, sp, rho, budget_moves):
    """The oracle's sequential LinkedList path (one thread) on the same input, bounded sample."""
    from oracle import oracle as O
    t0 = time.perf_counter()
    ref = O.System(pos, sp, rho, TEMPERATURE, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    t_build = time.perf_counter() - t0
    pool = O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}])
    # calibrate on a short burst, then run the bounded sample
    t0 = time.perf_counter()
    ref.mc_steps(pool, 42, 0, 0, 20000)
    rate = 20000 / (time.perf_counter() - t0)
    n = int(min(budget_moves, max(20000, rate * 12.0)))
    t0 = time.perf_counter()
    att, acc = ref.mc_steps(pool, 42, 0, 20000, n)
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "attempted moves/s", "cores": 1, "kind": "port",
        "sample": f"{n} sequential Displacement steps (reference LinkedList path, oracle/pmc_oracle.c) on the same "
                  f"N={len(pos)} input after a {t_build:.1f} s build; acceptance {acc[0] / max(1, att[0]):.3f}",
    }


def _cpu_worker(args):
    """One independent chain of the oracle's sequential path (own process = own host core)."""
    lattice, chain, nsteps = args
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from particlesmc_amd.workloads import lj_lattice
    pos, sp, rho = lj_lattice(lattice, jitter=0.05, seed=42)
    ref = O.System(pos, sp, rho, TEMPERATURE, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    pool = O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}])
    ref.mc_steps(pool, 42 + chain, chain, 0, 2000)
    t0 = time.perf_counter()
    ref.mc_steps(pool, 42 + chain, chain, 2000, nsteps)
    return nsteps, time.perf_counter() - t0


def cpu_baseline_multicore(lattice, single_rate, max_cores=32):
    """one independent chain per host core (reference `parallel=true`),
    bounded to `max_cores` processes of ~10 s each; aggregate attempted moves/s."""
    import multiprocessing as mp
    cores = max(1, min(max_cores, os.cpu_count() or 1))
    nsteps = int(max(20000, single_rate * 2.5))    # chains slow each other down ~4x (shared caches, DRAM)
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(lattice, c, nsteps) for c in range(cores)])
    wall = time.perf_counter() - t0
    tmax = max(r[1] for r in res)
    return {"value": sum(r[0] for r in res) / tmax, "unit": "attempted moves/s", "cores": cores, "kind": "port",
            "sample": f"{cores} independent chains (one process per core), {nsteps} sequential steps each on the "
                      f"same N input; slowest chain {tmax:.1f} s, {wall:.1f} s including the builds"}


# ---- launcher: `bench.py --gpus N` started plainly ------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(ngpus, argv):
    """Start `ngpus` ranks of this script (one process per GPU) as CHILD processes and return their
    exit status.  Nothing in this (parent) process has touched the GPU: no torch import, no HIP call."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ---- test hook: the launcher / reduction logic without a GPU ----------------------------------
class _StubEngine:
    """PMC_BENCH_STUB=1 (tests/test_bench_launcher.py only): no device work at all, so the multi-rank
    launch, the barriers and the max-over-ranks reduction of this script can be exercised over gloo on a
    machine without a GPU.  The JSON line then carries "stub": true and its value is meaningless."""

    class _Stats:
        accepted = [0]
        attempted = [1]
        energy_fix = 0
        sweep_kernel_ms = 0.0
        sweep_kernel_launches = 0
        pair_tests = 0
        pairs_in_range = 0
        null_moves = 0

    def __init__(self, N):
        self.N = N

    def sweep(self, nsweeps):
        time.sleep(0.002 * nsweeps)
        return self._Stats()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lattice", type=int, default=100, help="lattice sites per side (N = lattice^3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--equilibrate", type=int, default=1000,
                    help="untimed sweeps that melt the jittered lattice into the rho = 1, T = 1 liquid before the warm-up "
                         "(0: time the first sweeps from the lattice instead, as rounds 1-2 did; with > 0 that figure is still "
                         "measured first and reported as \"from_lattice\")")
    ap.add_argument("--share-devices", action="store_true",
                    help="functional testing only: allow more ranks than visible GPUs (ranks share devices, gloo reductions)")
    return ap.parse_args(argv)


def roofline_objects(N, steps, st, avg_launch_s, moves_per_launch, ncell_ref):
    """The bench line's roofline fields.  Everything under "roofline" is measured in
    THIS run (kernel counters + HIP events) except the entries that carry a "source"."""
    b_alg = algorithmic_bytes_per_move(N, 3, ncell_ref)
    moves_timed = float(N) * steps
    tests = float(getattr(st, "pair_tests", 0))
    inrange = float(getattr(st, "pairs_in_range", 0))
    flop_per_move = (tests * FLOP_PER_PAIR_TEST + inrange * FLOP_PER_PAIR_ENERGY) / moves_timed if tests > 0 else None
    roof = {
        "bound": "fp64-valu", "unit": "TFLOP/s", "peak": FP64_VECTOR_PEAK_TFLOPS,
        "achieved": None, "frac": None, "traffic": None,
        "kernel": "k_cb_pass<3,LJ,single-species> (one launch = one fused sweep of all 8 colours)",
        "moves_per_launch": moves_per_launch, "avg_launch_ms": avg_launch_s * 1e3,
        "flop_model": {"per_pair_test": FLOP_PER_PAIR_TEST, "per_pair_energy": FLOP_PER_PAIR_ENERGY,
                       "pair_tests_per_move": tests / moves_timed if tests > 0 else None,
                       "pair_energies_per_move": inrange / moves_timed if tests > 0 else None,
                       "flop_per_move": flop_per_move,
                       "note": "pair tests and in-range pair energies are COUNTED by the kernel in this run "
                               "(cell-leaving proposals test nothing); flop = tests*9 + energies*9, FMA-free FP64 "
                               "operation counts of the reference arithmetic; peak counts an FMA as 2"},
    }
    if flop_per_move is not None:
        roof["achieved"] = flop_per_move * moves_per_launch / avg_launch_s / 1e12
        roof["frac"] = roof["achieved"] / FP64_VECTOR_PEAK_TFLOPS
    # the SURVEY 8(d) contract figure, kept as a clearly named secondary field: it is NOT a fraction of anything
    # the kernel is bound by (the neighbourhood is staged once per cell visit and reused from LDS)
    contract = b_alg * moves_per_launch / avg_launch_s / 1e9
    roof["contract_hbm_equiv"] = {"algorithmic_bytes_per_move": b_alg, "GBps_equivalent": contract,
                                  "over_hbm_peak": contract / HBM_PEAK_GBS,
                                  "reference_traversal_bytes_per_move": 2 * 27 * (N / float(np.prod(ncell_ref))) * 40.0,
                                  "note": "SURVEY 8(d) B_alg x moves / launch time; may exceed 1: not a bound"}
    # counters that need rocprofv3 passes cannot be measured inside this run: replayed from the committed profile
    # of the same command, labelled with their source
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            src = t.get("source", "profiles/traffic.json")
            mpl = float(t.get("moves_per_launch", 0)) or None
            if mpl and t.get("lattice", 100) ** 3 == N:
                hbm_bytes = t["k_cb_pass_hbm_bytes_per_launch"] * moves_per_launch / mpl
                roof["traffic"] = hbm_bytes
                roof["traffic_source"] = src
                roof["hbm_measured_frac"] = {"value": hbm_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS, "source": src,
                                             "note": "PMC bytes of the profiled launch over THIS run's launch time"}
                if t.get("SQ_INSTS_VALU_per_launch"):
                    per_move = t["SQ_INSTS_VALU_per_launch"] / mpl
                    roof["valu_issue_frac"] = {"value": per_move * moves_per_launch / avg_launch_s / FP64_ISSUE_PEAK,
                                               "wave_instructions_per_move": per_move, "source": src,
                                               "note": "VALU wave-instructions/s over 256 CUs x 4 SIMDs x 1 per 4 cycles x 2.4 GHz (the FP64 issue rate; FP32 and integer instructions issue in 2.5 cycles, so this ratio is not capped at 1)"}
        except Exception as exc:      # a malformed profile summary must not cost the bench line
            roof["traffic_source"] = f"profiles/traffic.json unreadable: {exc!r}"
    return roof


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    stub = os.environ.get("PMC_BENCH_STUB") == "1"
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it plainly (it launches its own ranks) or "
              f"with --nproc-per-node {args.gpus}", file=sys.stderr)
        return 2
    import torch
    dist = None
    backend = None
    ndev = 0 if stub else torch.cuda.device_count()
    if not stub:
        if ndev < 1:
            print("bench.py needs a HIP device; the HIP path has no CPU fallback", file=sys.stderr)
            return 3
        if ndev < world and not args.share_devices:
            print(f"bench.py: {world} ranks but only {ndev} visible GPU(s); one rank per GPU is the contract "
                  "(--share-devices overrides this for functional testing)", file=sys.stderr)
            return 3
    dev_index = local_rank % max(1, ndev)
    if world > 1:
        import torch.distributed as dist
        backend = "nccl" if (not stub and ndev >= world) else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    ngpus = world

    from particlesmc_amd.workloads import lj_lattice
    pos, sp, rho = lj_lattice(args.lattice, jitter=0.05, seed=42)
    N = len(pos)
    if stub:
        engine = _StubEngine(N)
        sweep = engine.sweep
        ncell_ref, grid = [1, 1, 1], [0, 0, 0]
        e0 = 0.0

        def total_energy():
            return 0.0, 0
    else:
        import particlesmc_amd as pmc
        if pmc.device_count() < 1:
            raise RuntimeError("bench.py needs a HIP device; the HIP path has no CPU fallback")
        system = pmc.System(pos, sp, rho, TEMPERATURE, pmc.LennardJones(1.0, 1.0), seed=42 + rank, chain=rank,
                            device=dev_index)
        pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": SIGMA_MOVE}, 1.0)]
        e0, _ = system.total_energy()
        ncell_ref = [int(v) for v in system.neighbour_list.ncells]
        grid = [int(v) for v in system.grid()[0]]
        total_energy = system.total_energy

        def sweep(nsweeps):
            return pmc.mc_sweep(system, pool, nsweeps=nsweeps)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        sync()

    from_lattice = None
    if args.equilibrate > 0:
        # the state rounds 1-2 timed: the first sweeps from the jittered lattice (same warm-up, same number of steps)
        if args.warmup > 0:
            sweep(args.warmup)
        sync()
        tl = time.perf_counter()
        stl = sweep(args.steps)
        sync()
        tl = time.perf_counter() - tl
        from_lattice = {"value": float(N) * args.steps * ngpus / tl, "ms_per_step": tl * 1e3 / args.steps,
                        "acceptance": float(stl.accepted[0]) / float(max(1, stl.attempted[0])),
                        "note": "this rank's first sweeps from the jittered lattice, before the equilibration (not max-over-ranks)"}
        sweep(args.equilibrate)
    if args.warmup > 0:
        sweep(args.warmup)
    barrier()
    t0 = time.perf_counter()
    st = sweep(args.steps)       # synchronises its stream before returning
    sync()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{dev_index}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # exactness property at full size: the running energy equals a recomputation bit for bit
    run_fix = st.energy_fix
    e1, efix = total_energy()
    energy_ok = bool(efix == run_fix)
    if dist is not None:
        ok = torch.tensor([1.0 if energy_ok else 0.0], dtype=torch.float64,
                          device=f"cuda:{dev_index}" if backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        energy_ok = bool(ok.item() == 1.0)

    if rank == 0:
        moves = float(N) * args.steps * ngpus
        value = moves / elapsed
        launches = int(st.sweep_kernel_launches)
        timed_steps = min(args.steps, 4096)
        if launches > 0 and st.sweep_kernel_ms > 0:
            avg_launch_s = st.sweep_kernel_ms * 1e-3 / launches
            moves_per_launch = N * timed_steps / launches
        else:
            launches = args.steps
            avg_launch_s = elapsed / max(1, launches)
            moves_per_launch = float(N)
        attempted = float(max(1, st.attempted[0]))
        out = {
            "metric": "attempted MC moves/sec, N=10^6 3D LJ rho=1.0 (one sweep = N attempted moves)",
            "value": value, "unit": "attempted moves/s", "n_gpus": ngpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={N} 3D LJ rho=1.0 T=1.0 rcut=2.5 shifted, Displacement sigma=0.05, "
                                   + (f"liquid equilibrated by {args.equilibrate} untimed sweeps from the jittered {args.lattice}^3 lattice"
                                      if args.equilibrate > 0 else f"jittered {args.lattice}^3 lattice")
                                   + ", one independent chain per GPU",
                       "schedule": "checkerboard", "grid": grid, "reference_cells": ncell_ref,
                       "equilibration_sweeps": args.equilibrate,
                       "backend": backend or "single process", "ranks": world},
            "sweeps_per_s": args.steps * ngpus / elapsed,
            "acceptance": float(st.accepted[0]) / attempted,
            # cell-leaving proposals are rejected without any dE work; they are part of the schedule and count
            # as attempted moves, so the fraction is printed next to the headline
            "null_move_fraction": float(getattr(st, "null_moves", 0)) / attempted,
            "energy_per_particle": e1 / N, "energy_per_particle_initial": e0 / N,
            "running_energy_equals_recomputed": energy_ok,
            "roofline": roofline_objects(N, timed_steps, st, avg_launch_s, moves_per_launch, ncell_ref),
        }
        if from_lattice is not None:
            out["from_lattice"] = from_lattice
        if stub:
            out["stub"] = True
        if ngpus == 1 and not args.no_cpu and not stub:
            out["cpu_baseline"] = cpu_baseline(pos, sp, rho, 3_000_000)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            try:
                out["cpu_baseline_multicore"] = cpu_baseline_multicore(args.lattice, out["cpu_baseline"]["value"])
            except Exception as exc:      # the single-core figure is the contract; this one is a courtesy
                out["cpu_baseline_multicore"] = {"error": repr(exc)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)        # parent: no torch, no HIP; the ranks are child processes
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
