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

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rounds M] [--lattice 100] [--no-cpu]

The headline `value` runs the schedule with M = 4 sweeps per grid shift (pmc_set_rounds); the same liquid is
timed with M = 1 (what rounds 1-2 timed) and M = 2 in the same run ("by_rounds"), and what an attempted move of each is
worth in attempted moves of the reference's sequential chain is measured beside them ("move_efficiency_vs_sequential").

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
import platform
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz = half the 157.3 TF FP32 vector peak
FP32_VECTOR_PEAK_TFLOPS = 157.3
FP64_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0   # FP64 wave-instructions/s: one per 4 cycles per SIMD
SIGMA_MOVE = 0.05
TEMPERATURE = 1.0
LATTICE_WARMUP, LATTICE_STEPS = 5, 20     # the protocol of the "from_lattice" figure: what rounds 1-2 timed
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
        "sample": f"{n} sequential Displacement steps (reference LinkedList path, oracle/pmc_oracle.c) from the state the "
                  f"GPU's timed region starts from (N={len(pos)}), after a {t_build:.1f} s build; acceptance {acc[0] / max(1, att[0]):.3f}",
    }


def _cpu_worker(args):
    """One independent chain of the oracle's sequential path (own process = own host core)."""
    state_path, rho, chain, nsteps = args
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    pos = np.load(state_path)
    sp = np.ones(len(pos), dtype=np.int32)
    ref = O.System(pos, sp, rho, TEMPERATURE, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    pool = O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}])
    ref.mc_steps(pool, 42 + chain, chain, 0, 2000)
    t0 = time.perf_counter()
    ref.mc_steps(pool, 42 + chain, chain, 2000, nsteps)
    return nsteps, time.perf_counter() - t0


def host_cpu_model():
    """`lscpu`'s model string of the host the CPU legs run on ("print nproc / lscpu model and core count")."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline_multicore(pos, rho, single_rate, max_cores=None):
    """BASELINE.md section 4 (b), SURVEY 8(d): one independent chain per host core (the reference's `parallel = true`,
    src/ParticlesMC.jl:164,246) from the same state, on ALL host cores the process may use (os.sched_getaffinity) unless
    `max_cores` / --cpu-cores caps it (the line then says so); ~5 s of steps per chain; aggregate attempted moves/s over the
    slowest chain's time."""
    import multiprocessing as mp
    import tempfile
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    # every chain holds its own copy of the state and lists (~0.2 GB at N = 10^6): never more chains than memory allows
    try:
        mem_gb = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
    except (ValueError, OSError):
        mem_gb = 64.0
    mem_cap = max(1, int(0.5 * mem_gb / (0.25 * max(1.0, len(pos) / 1e6))))
    cores = max(1, min(avail, mem_cap, max_cores or avail))
    cap_reason = None
    if cores < avail:
        cap_reason = "--cpu-cores" if (max_cores and cores == max_cores) else "host memory"
    # chains slow each other down (shared caches, DRAM, SMT siblings): ~4x at 32 chains, more on every hardware thread
    nsteps = int(max(20000, single_rate * (2.5 if cores <= 32 else 1.0)))
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "state.npy")
        np.save(path, pos)
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(path, rho, c, nsteps) for c in range(cores)])
        wall = time.perf_counter() - t0
    tmax = max(r[1] for r in res)
    return {"value": sum(r[0] for r in res) / tmax, "unit": "attempted moves/s", "cores": cores, "kind": "port",
            "host_cores_available": avail, "cores_capped_by": cap_reason, "cpu_model": host_cpu_model(),
            "sample": f"{cores} independent chains (one process per hardware thread the host offers: {avail}), {nsteps} sequential "
                      f"steps each from the same state; slowest chain {tmax:.1f} s, {wall:.1f} s including the builds"}


# ---- launcher: `bench.py --gpus N` started plainly ------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _descendants(pid):
    """Every live descendant of `pid`, from /proc/*/stat (field 4 = ppid; the command name in field 2 may contain spaces and
    parentheses, so the line is split after its last ')')."""
    parent = {}
    try:
        entries = os.listdir("/proc")
    except OSError:
        return []
    for e in entries:
        if not e.isdigit():
            continue
        try:
            with open(f"/proc/{e}/stat") as f:
                st = f.read()
            parent[int(e)] = int(st[st.rindex(")") + 2:].split()[1])
        except (OSError, ValueError, IndexError):
            continue
    out, front = [], [pid]
    while front:
        nxt = [c for c, pp in parent.items() if pp in front and c not in out]
        out.extend(nxt)
        front = nxt
    return out


def launch_ranks(ngpus, argv, timeout=None):
    """Start `ngpus` ranks of this script (one process per GPU) as CHILD processes and return their
    exit status.  Nothing in this (parent) process has touched the GPU: no torch import, no HIP call."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    if not timeout or timeout <= 0:
        return subprocess.call(cmd, env=env)
    # --launch-timeout: a rank that hangs (a rendezvous that never completes, a collective that never returns) must end the job
    # with a non-zero status, never with a re-exec: the ranks run in a process group of their own, which is killed as a whole
    import signal
    p = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        # exactly the processes this call started: the launcher, its process group, and the ranks it spawned (the elastic agent
        # gives its workers sessions of their own, so the group alone would leave them -- and the caller's pipes -- alive).
        # The descendants are read from /proc (no optional package); the launcher gets SIGTERM and a short grace first, because
        # the elastic agent tears its workers down itself on SIGTERM; whatever is left is killed
        kids = _descendants(p.pid)
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        try:
            p.wait(timeout=5.0)
        except subprocess.TimeoutExpired:
            pass
        kids = sorted(set(kids) | set(_descendants(p.pid)))
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        left = []
        for k in kids:
            try:
                os.kill(k, signal.SIGKILL)
            except ProcessLookupError:
                pass
            except Exception:
                left.append(k)
        p.wait()
        if left:
            print(f"bench.py: could not signal {len(left)} descendant process(es) of the launcher: {left}", file=sys.stderr)
        print(f"bench.py: the {ngpus} ranks did not finish within {timeout:.0f} s: killed", file=sys.stderr)
        return 5


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

    def __init__(self, N, rank=0):
        self.N = N
        self.hang = os.environ.get("PMC_BENCH_STUB_HANG") == str(rank)     # (the launcher's timeout test: this rank never returns)

    def sweep(self, nsweeps):
        if self.hang:
            time.sleep(3600)
        time.sleep(0.002 * nsweeps)
        return self._Stats()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattice", type=int, default=100, help="lattice sites per side (N = lattice^3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU legs (baseline and move efficiency)")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="cap of the multi-core CPU leg (0: every hardware thread the host offers, SURVEY 8d)")
    ap.add_argument("--equilibrate", type=int, default=1000,
                    help="untimed sweeps that melt the jittered lattice into the rho = 1, T = 1 liquid before the warm-up "
                         "(0: time the first sweeps from the lattice instead, as rounds 1-2 did; with > 0 that figure is still "
                         "measured first and reported as \"from_lattice\")")
    ap.add_argument("--rounds", type=int, default=4,
                    help="schedule parameter m of the headline figure (pmc_set_rounds: sweeps per grid shift = proposal rounds per "
                         "cell visit); m = 1 -- what rounds 1-2 timed -- and m = 2 are timed in the same run and reported beside it")
    ap.add_argument("--long-steps", type=int, default=1000, help="steps of the secondary, longer timed region (0: skip)")
    ap.add_argument("--only-headline", action="store_true",
                    help="profiling runs: skip the lattice figure, the other schedules and the long region, so that the LAST "
                         "steps / rounds launches of the sweep kernel are the timed region (profiles/collect.sh)")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the bounded block of the other BASELINE configurations and of the reference-order chain (\"other_configs\", \"reference_order_chain\")")
    ap.add_argument("--launch-timeout", type=float, default=0.0,
                    help="seconds after which the self-launched ranks are killed as a group and the exit status is 5 (0: wait)")
    ap.add_argument("--strict-rccl", action="store_true",
                    help="exit with status 4 unless the replica-exchange self-check ran on every rank and agreed with pmc_pt_decide")
    ap.add_argument("--share-devices", action="store_true",
                    help="functional testing only: allow more ranks than visible GPUs (ranks share devices, gloo reductions)")
    return ap.parse_args(argv)


def roofline_objects(N, steps, st, avg_launch_s, moves_per_launch, ncell_ref):
    """The bench line's roofline fields.  Everything under "roofline" is measured in
    THIS run (kernel counters + HIP events) except the entries that carry a "source"."""
    b_alg = algorithmic_bytes_per_move(N, 3, ncell_ref)
    moves_timed = float(N) * steps            # the counters cover every sweep of the call
    tests = float(getattr(st, "pair_tests", 0))
    inrange = float(getattr(st, "pairs_in_range", 0))
    dense = float(getattr(st, "dense_evals", 0))
    staged = float(getattr(st, "staged_candidates", 0))
    written = float(getattr(st, "written_back", 0))
    flop_per_move = (tests * FLOP_PER_PAIR_TEST + inrange * FLOP_PER_PAIR_ENERGY) / moves_timed if tests > 0 else None
    roof = {
        "bound": "fp64-valu", "unit": "TFLOP/s", "peak": FP64_VECTOR_PEAK_TFLOPS,
        "achieved": None, "frac": None, "traffic": None,
        "kernel": "k_cb_pass<3,LJ,single-species> (one launch = one pass: all 8 colours, `rounds` sweeps)",
        "moves_per_launch": moves_per_launch, "avg_launch_ms": avg_launch_s * 1e3,
        "flop_model": {"per_pair_test": FLOP_PER_PAIR_TEST, "per_pair_energy": FLOP_PER_PAIR_ENERGY,
                       "pair_tests_per_move": tests / moves_timed if tests > 0 else None,
                       "pair_energies_per_move": inrange / moves_timed if tests > 0 else None,
                       "flop_per_move": flop_per_move,
                       "note": "ALGORITHMIC flop: every (candidate, position) distance test the schedule needs after the box "
                               "filter and every in-range pair energy, COUNTED by the kernel in this run and priced as the FP64 "
                               "operations of the reference arithmetic (9 + 9, FMA-free; the peak counts an FMA as 2).  The kernel "
                               "EXECUTES most of the tests in packed FP32 (pre-filter): see `executed`"},
    }
    if flop_per_move is not None:
        roof["achieved"] = flop_per_move * moves_per_launch / avg_launch_s / 1e12
        roof["frac"] = roof["achieved"] / FP64_VECTOR_PEAK_TFLOPS
        # what the kernel executes, by precision: FP64 = distance evaluations of the pre-filter's survivors + pair energies;
        # FP32 = the pre-filter's distance tests (two per packed instruction lane)
        f64 = (dense * FLOP_PER_PAIR_TEST + inrange * FLOP_PER_PAIR_ENERGY) / moves_timed
        f32 = tests * FLOP_PER_PAIR_TEST / moves_timed
        rate = moves_per_launch / avg_launch_s / 1e12
        roof["executed"] = {"fp64_flop_per_move": f64, "fp64_TFLOPs": f64 * rate, "fp64_frac_of_peak": f64 * rate / FP64_VECTOR_PEAK_TFLOPS,
                            "fp32_flop_per_move": f32, "fp32_TFLOPs": f32 * rate, "fp32_frac_of_peak": f32 * rate / FP32_VECTOR_PEAK_TFLOPS,
                            "dense_evals_per_move": dense / moves_timed}
    # bytes the schedule moves, from counts the kernel takes in this run: every candidate a visit loads (3 doubles), every
    # proposal slot it reads (4 doubles + 2 words + 1 byte), every particle it writes back (sorted copy + position array)
    if staged > 0:
        sched = (staged * 24.0 + moves_timed * 41.0 + written * 48.0) / moves_timed * moves_per_launch
        roof["schedule_bytes"] = {"per_launch": sched, "GBps": sched / avg_launch_s / 1e9, "frac_of_hbm_peak": sched / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                                  "staged_candidates_per_move": staged / moves_timed, "written_back_per_move": written / moves_timed,
                                  "note": "counted in-run: 24 B per staged candidate + 41 B per proposal slot + 48 B per written-back particle; "
                                          "the binning kernels' traffic (about 100 B per particle and pass) is not included"}
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


def move_efficiency(pmc, rounds_list):
    """Second half of the CPU-baseline leg (rank 0, N = 1, skipped with --no-cpu; like `cpu_baseline` it runs the oracle's
    sequential chain on host cores, as the yardstick, never as the thing measured): what an attempted move of the checkerboard schedule is worth in attempted moves of the reference's sequential chain
    (src/moves.jl:57-81): mean-square displacement per attempted move, checkerboard (device) over sequential (oracle), on an
    N = 4096 liquid at the benchmark's state point and cell width (6 cells of 2.667).  A bounded sample: 16 chains of 500
    sweeps per leg (about 20 s of CPU on 16 host cores).  The integrated autocorrelation time of e/N needs ~100x the CPU
    time: it is measured by tests/test_gpu_efficiency.py / examples/move_efficiency.py and replayed with its source."""
    from oracle import efficiency as E
    from particlesmc_amd.workloads import lj_lattice
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
    g = pmc.System(pos, sp, rho, TEMPERATURE, pmc.LennardJones(1.0, 1.0), seed=4242)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": SIGMA_MOVE}, 1.0)]
    pmc.mc_sweep(g, pool, nsweeps=3000)
    p0 = g.position.copy()
    nch, nsw = 16, 500
    _, seq_msd, seq_acc = E.sequential_leg(p0, sp, rho, TEMPERATURE, SIGMA_MOVE, nch, nsw, stride=nsw, processes=nch)
    out = {"observable": f"MSD after {nsw} sweeps per attempted move, checkerboard (device) / sequential (oracle), N=4096 LJ liquid "
                         f"rho=1 T=1 sigma={SIGMA_MOVE}, {nch} chains per leg", "by_rounds": {}}
    for m in rounds_list:
        _, cb_msd, cb_acc, cb_null = E.gpu_cb_leg(pmc, p0, sp, rho, TEMPERATURE, SIGMA_MOVE, nch, nsw, stride=nsw, rounds=m)
        r = E.efficiency(None, seq_msd, None, cb_msd)
        out["by_rounds"][str(m)] = {"msd_ratio": r["msd_ratio"], "sigma": r["msd_ratio_sigma"], "null_move_fraction": cb_null,
                                    "acceptance": cb_acc}
    out["acceptance_sequential"] = seq_acc
    epath = os.path.join(ROOT, "profiles", "move_efficiency.json")
    if os.path.exists(epath):
        try:
            t = json.load(open(epath))
            out["tau_int"] = {"by_rounds": t.get("tau_ratio_by_rounds"), "source": t.get("source", "profiles/move_efficiency.json"),
                              "note": "tau_int(e/N) of the sequential chain over the checkerboard's, per attempted move"}
        except Exception as exc:
            out["tau_int"] = {"error": repr(exc)}
    return out


def rccl_exchange_check(rank, world, dev_index, dist, rounds=4, timeout=150.0):
    """The path's one collective on the ranks of this job, each rank in a CHILD process of its own
    (`python -m particlesmc_amd.rccl_check`): config 5's system on the rank's device, a communicator built by the library's own
    entry points (pmc_rccl_unique_id -> file -> pmc_rccl_comm_create), `rounds` rounds of sweeps + pmc_replica_exchange_rccl
    (ncclAllGather of one double per rank).  Rank 0 then checks every rank's labels against pmc_pt_decide on the energies all
    ranks reported.  A child, because nothing that happens inside a communicator that has never run on this node -- a hang, a
    crash -- may cost the bench line: a rank that does not finish within `timeout` seconds is killed and reported."""
    import shutil
    import tempfile
    from particlesmc_amd import tempering
    box = [tempfile.mkdtemp(prefix="pmc_bench_rccl_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    outdir = box[0]
    env = dict(os.environ)
    for k in list(env):
        if k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT") or k.startswith("TORCHELASTIC"):
            env.pop(k)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.abspath(__file__))
    err = None
    try:
        p = subprocess.Popen([sys.executable, "-m", "particlesmc_amd.rccl_check", str(rank), str(world), str(dev_index), outdir, str(rounds)],
                             cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        try:
            out = p.communicate(timeout=timeout)[0].decode(errors="replace")
            if p.returncode != 0:
                err = f"rank {rank}: exit code {p.returncode}: {out[-400:]}"
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            err = f"rank {rank}: pmc_replica_exchange_rccl did not finish within {timeout:.0f} s"
    except Exception as exc:
        err = f"rank {rank}: {exc!r}"
    errs = [None] * world
    dist.all_gather_object(errs, err)                     # also the barrier: every rank's file is written (or its error known)
    res = None
    if rank == 0:
        errs = [e for e in errs if e]
        if errs:
            res = {"n_ranks_rccl": 0, "error": "; ".join(errs)[:1500]}
        else:
            try:
                ranks = [json.load(open(os.path.join(outdir, f"rank{r}.json"))) for r in range(world)]
                ladder = ranks[0]["ladder"]
                label, ok, acc = list(range(world)), True, 0
                for rnd in range(rounds):
                    energies = [ranks[r]["rounds"][rnd]["energy"] for r in range(world)]
                    label, nacc = tempering.pt_decide(9, rnd, energies, ladder, label)
                    label = [int(v) for v in label]
                    acc += int(nacc)
                    for r in range(world):
                        got = ranks[r]["rounds"][rnd]
                        ok = ok and got["label"] == label and got["nacc"] == int(nacc) and got["temperature"] == ladder[label[r]]
                res = {"n_ranks_rccl": world, "rounds": rounds, "decisions_equal_pt_decide": bool(ok), "accepted": acc,
                       "system": "config 5: 2-D binary r^-14 soft spheres, N = 4096 per rank, 5 sweeps per round"}
            except Exception as exc:
                res = {"n_ranks_rccl": 0, "error": repr(exc)}
        shutil.rmtree(outdir, ignore_errors=True)
    return res


def _oracle_rate(make_oracle, pool_o, seed, budget_s=1.5):
    """Attempted moves/s of the oracle's sequential chain (one host core) on a system, bounded sample."""
    o = make_oracle()
    t0 = time.perf_counter()
    o.mc_steps(pool_o, seed, 0, 0, 4000)
    r = 4000 / (time.perf_counter() - t0)
    n = int(max(4000, r * budget_s))
    t0 = time.perf_counter()
    o.mc_steps(pool_o, seed, 0, 4000, n)
    return n / (time.perf_counter() - t0)


def _flop_frac(st, seconds):
    """Counted algorithmic flop (9 per distance test + 9 per pair energy, as the headline's roofline) over the FP64 peak."""
    tests, inr = float(getattr(st, "pair_tests", 0)), float(getattr(st, "pairs_in_range", 0))
    if tests <= 0 or seconds <= 0:
        return None
    return (tests * FLOP_PER_PAIR_TEST + inr * FLOP_PER_PAIR_ENERGY) / seconds / 1e12 / FP64_VECTOR_PEAK_TFLOPS


def _seq_flop_frac(N, d, ncells_ref, rate, in_range_per_eval):
    """The reference-order chain's algorithmic flop (two traversals of the 3^d neighbour cells per step, src/atoms.jl:81-88,
    plus the in-range pair energies) over the FP64 peak: this chain is a latency chain, not an arithmetic one."""
    cand = (3 ** d) * N / float(np.prod(ncells_ref))
    return (2 * cand * FLOP_PER_PAIR_TEST + 2 * in_range_per_eval * FLOP_PER_PAIR_ENERGY) * rate / 1e12 / FP64_VECTOR_PEAK_TFLOPS


def reference_order_chain(pmc, system, pool, nsteps_target=3_000_000):
    """The reference's own chain -- one particle per step, every step seeing the state all earlier steps left
    (src/moves.jl:11-20,57-81) -- on the headline's system and state: PMC_SEQUENTIAL, bit-identical to the CPU baseline's
    chain (tests/test_gpu_sequential.py), so GPU / CPU here is like for like."""
    N = system.N
    nsw = max(1, int(nsteps_target // N))
    pmc.mc_sweep(system, pool, nsweeps=1, mode="sequential")
    t0 = time.perf_counter()
    st = pmc.mc_sweep(system, pool, nsweeps=nsw, mode="sequential")
    dt = time.perf_counter() - t0
    rounds, evals, window, folded = system.sequential_info()
    return {"value": nsw * N / dt, "unit": "attempted moves/s", "steps": nsw * N, "acceptance": float(st.accepted[0]) / max(1, st.attempted[0]),
            "engine": "speculative windows (pmc_seq.h): windows of pending steps evaluated at once, committed in dependency order",
            "window": window, "rounds": rounds, "steps_per_round": nsw * N / max(1, rounds), "evaluations_per_step": evals / max(1, nsw * N),
            "us_per_round": 1e3 * st.kernel_ms / max(1, rounds),
            "note": "same chain as cpu_baseline bit for bit (positions, lists, accept counts, system.energy)"}


def other_configs(pmc, budget_cpu_s=1.2):
    """BASELINE.json's other configurations and the reference's own small-system workload, each on this GPU in a bounded
    region, with the oracle's sequential chain on one host core beside it.  About a minute in all."""
    from oracle import oracle as O
    from particlesmc_amd import tempering
    from particlesmc_amd.workloads import binary_lattice, lj_lattice
    rows = []

    def disp(sigma=SIGMA_MOVE, p=1.0):
        return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)

    def cb_leg(system, pool, sweeps, warm, m):
        system.rounds = m
        pmc.mc_sweep(system, pool, nsweeps=warm)
        t0 = time.perf_counter()
        st = pmc.mc_sweep(system, pool, nsweeps=sweeps)
        dt = time.perf_counter() - t0
        att = float(np.sum(st.attempted))
        return {"moves_per_s": att / dt, "rounds": m, "roofline_frac": _flop_frac(st, st.sweep_kernel_ms * 1e-3),
                "acceptance": [float(a) / max(1.0, float(b)) for a, b in zip(st.accepted, st.attempted) if b > 0],
                "running_energy_equals_recomputed": bool(system.total_energy()[1] == st.energy_fix)}

    def seq_leg(system, pool, steps):
        nsw = max(1, int(steps // system.N))
        pmc.mc_sweep(system, pool, nsweeps=1, mode="sequential")
        t0 = time.perf_counter()
        st = pmc.mc_sweep(system, pool, nsweeps=nsw, mode="sequential")
        dt = time.perf_counter() - t0
        return {"moves_per_s": nsw * system.N / dt, "acceptance": [float(a) / max(1.0, float(b)) for a, b in zip(st.accepted, st.attempted) if b > 0]}

    # -- config 2: N = 32768 LJ liquid
    pos, sp, rho = lj_lattice(32, jitter=0.05, seed=42)
    s = pmc.System(pos, sp, rho, TEMPERATURE, pmc.LennardJones(1.0, 1.0), seed=42)
    pool = [disp()]
    s.total_energy()
    pmc.mc_sweep(s, pool, nsweeps=300)                           # melt
    row = {"config": "C2: N=32768 3D LJ rho=1 T=1 rcut=2.5, Displacement sigma=0.05 (liquid: 300 untimed sweeps from the lattice)",
           "checkerboard_m1": cb_leg(s, pool, 300, 20, 1), "checkerboard_m4": cb_leg(s, pool, 300, 20, 4)}
    s.rounds = 1
    state = s.position.copy()
    row["reference_order_chain"] = seq_leg(s, pool, 1.2e6)
    row["reference_order_chain"]["roofline_frac"] = _seq_flop_frac(s.N, 3, s.neighbour_list.ncells, row["reference_order_chain"]["moves_per_s"], 62.0)
    row["cpu_oracle_moves_per_s"] = _oracle_rate(lambda: O.System(state, sp, rho, TEMPERATURE, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST),
                                                 O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}]), 42, budget_cpu_s)
    rows.append(row)
    # -- config 3: Kob-Andersen N = 10^5, 90 % Displacement + 10 % DiscreteSwap
    pos, sp, rho = binary_lattice(100000, frac_b=0.2, rho=1.2, jitter=0.05, seed=42)
    s = pmc.System(pos, sp, rho, TEMPERATURE, pmc.KobAndersen(), seed=42)
    pool = [disp(p=0.9), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1)]
    pool_o = O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE, "probability": 0.9}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}])
    s.total_energy()
    row = {"config": "C3: Kob-Andersen 80:20 N=100000 rho=1.2 T=1, 90% Displacement + 10% DiscreteSwap (from the lattice)",
           "checkerboard_m1": cb_leg(s, pool, 100, 10, 1), "checkerboard_m4": cb_leg(s, pool, 100, 12, 4)}
    s.rounds = 1
    state, sstate = s.position.copy(), s.species.copy()
    row["reference_order_chain"] = seq_leg(s, pool, 1.0e6)
    row["reference_order_chain"]["roofline_frac"] = _seq_flop_frac(s.N, 3, s.neighbour_list.ncells, row["reference_order_chain"]["moves_per_s"], 75.0)
    row["cpu_oracle_moves_per_s"] = _oracle_rate(lambda: O.System(state, sstate, rho, TEMPERATURE, O.kob_andersen(), O.LINKEDLIST), pool_o, 42, budget_cpu_s)
    rows.append(row)
    # -- the reference's own validation example at a liquid density (examples/lj-mixture/run-validation.py:26-34,98-115): binary LJ,
    #    rcut = 4 sigma unshifted, 90 % Displacement + 10 % DiscreteSwap; N = 8000 so that a checkerboard grid exists (4 x 4 x 4 cells of
    #    125 particles: 3375 candidates per neighbourhood -- the big-neighbourhood instances of k_cb_pass, round 5)
    nmx = 20
    Nmx, rho_mx, T_mx = nmx ** 3, 0.8, 1.5
    Lmx = (Nmx / rho_mx) ** (1.0 / 3.0)
    gmx = np.stack(np.meshgrid(*[np.arange(nmx)] * 3, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    pos = (gmx + 0.5) * (Lmx / nmx)
    sp = np.ones(Nmx, dtype=np.int32)
    sp[np.random.default_rng(3).permutation(Nmx)[: Nmx // 2]] = 2
    EPS_MX, SIG_MX = ((1.0, 1.1523), (1.1523, 1.3702)), ((1.0, 1.0339), (1.0339, 1.0640))
    mm = [[pmc.LennardJones(EPS_MX[i][j], SIG_MX[i][j], rcut=4.0, shift_potential=False) for j in range(2)] for i in range(2)]
    s = pmc.System(pos, sp, Nmx / Lmx ** 3, T_mx, mm, seed=42)
    pool = [disp(p=0.9), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1)]
    s.total_energy()
    pmc.mc_sweep(s, pool, nsweeps=200)
    row = {"config": "published mixture (examples/lj-mixture): binary LJ rcut=4 unshifted, N=8000 rho=0.8 T=1.5 x=0.5, 90% Displacement + 10% DiscreteSwap "
                     "(4x4x4 cells of 125 particles; 200 untimed sweeps from the lattice)",
           "checkerboard_m1": cb_leg(s, pool, 100, 10, 1), "checkerboard_m4": cb_leg(s, pool, 100, 12, 4)}
    s.rounds = 1
    state, sstate = s.position.copy(), s.species.copy()
    mo_mx = O.model_matrix(O.lennard_jones, EPS_MX, SIG_MX, rcut=4.0, shift_potential=False)
    row["cpu_oracle_moves_per_s"] = _oracle_rate(lambda: O.System(state, sstate, Nmx / Lmx ** 3, T_mx, mo_mx, O.LINKEDLIST),
                                                 O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE, "probability": 0.9},
                                                              {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}]), 42, budget_cpu_s)
    rows.append(row)
    # -- the same example AS PUBLISHED (N = 1000: 2.7 cutoffs per side -- a single-cell box, round 5): one chain on one CU, 256 chains
    try:
        nm1 = 10
        Lm1 = (1000 / rho_mx) ** (1.0 / 3.0)
        g1 = np.stack(np.meshgrid(*[np.arange(nm1)] * 3, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
        pos1 = (g1 + 0.5) * (Lm1 / nm1)
        sp1 = np.ones(1000, dtype=np.int32)
        sp1[np.random.default_rng(3).permutation(1000)[:500]] = 2
        s1 = pmc.System(pos1, sp1, 1000 / Lm1 ** 3, T_mx, mm, seed=42)
        pool1 = [disp(p=0.9), pmc.Move(pmc.DiscreteSwap([1, 2], s1), pmc.DoubleUniform(), [], 0.1)]
        s1.total_energy()
        pmc.mc_sweep(s1, pool1, nsweeps=200)
        leg1 = cb_leg(s1, pool1, 200, 10, 1)
        chains = [pmc.System(pos1, sp1, 1000 / Lm1 ** 3, T_mx, mm, seed=42, chain=k) for k in range(256)]
        pmc.mc_sweep_many(chains, pool1, nsweeps=10)
        t0 = time.perf_counter()
        pmc.mc_sweep_many(chains, pool1, nsweeps=40)
        dtn = time.perf_counter() - t0
        st1, ss1 = s1.position.copy(), s1.species.copy()
        rows.append({"config": "published mixture AS PUBLISHED (examples/lj-mixture/run-validation.py:26-34,155-166): N=1000 rho=0.8 T=1.5 x=0.5 rcut=4, 90% Displacement + "
                               "10% DiscreteSwap; box of 2.7 cutoffs per side = ONE cell, the plain Metropolis chain with global DoubleUniform picks",
                     "checkerboard_m1": leg1, "checkerboard_m1_256_chains": {"moves_per_s": 256 * 1000 * 40 / dtn},
                     "cpu_oracle_moves_per_s": _oracle_rate(lambda: O.System(st1, ss1, 1000 / Lm1 ** 3, T_mx, mo_mx, O.LINKEDLIST),
                                                            O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE, "probability": 0.9},
                                                                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}]), 42, budget_cpu_s)})
        del chains
    except Exception as exc:
        rows.append({"config": "published mixture as published (N=1000)", "error": repr(exc)})
    # -- the reference's CI system (test/runtests.jl:22-91: config_0, N = 1290 2-D ternary JBB, T = 0.231) with a VerletList dr = 0.2 on
    #    the reference's own chain: engine 4 (round 6: the state in LDS, one wavefront resolving the steps while twelve evaluate the
    #    ones behind them), one chain and 256 chains of a parallel = true run in one launch; and the literal loop (engine 1) beside it
    try:
        from particlesmc_amd.io import load_configuration
        cfg = load_configuration(os.path.join(ROOT, "tests", "golden", "config_0.xyz"))
        cpos, csp, cbox = np.asarray(cfg["position"], dtype=np.float64), np.asarray(cfg["species"], dtype=np.int32), np.asarray(cfg["box"], dtype=np.float64)
        cpos = cpos - np.floor(cpos / cbox) * cbox
        crho = len(cpos) / float(np.prod(cbox))
        vpool = [disp()]

        def verlet_chain(k, engine=0):
            g = pmc.System(cpos, csp, crho, 0.231, pmc.JBB(), list_type=pmc.VerletList, list_parameters={"dr": 0.2}, seed=10, chain=k)
            if engine:
                g.set_sequential_engine(engine, 0)
            return g
        one = verlet_chain(0)
        pmc.mc_sweep(one, vpool, nsweeps=2, mode="sequential")
        t0 = time.perf_counter()
        st1 = pmc.mc_sweep(one, vpool, nsweeps=100, mode="sequential")
        dt1 = time.perf_counter() - t0
        vinfo = one.verlet_info()
        lit = verlet_chain(0, engine=1)
        pmc.mc_sweep(lit, vpool, nsweeps=1, mode="sequential")
        t0 = time.perf_counter()
        pmc.mc_sweep(lit, vpool, nsweeps=4, mode="sequential")
        dtl = time.perf_counter() - t0
        chains = [verlet_chain(k) for k in range(256)]
        pmc.mc_sweep_many(chains, vpool, nsweeps=1, mode="sequential")
        t0 = time.perf_counter()
        pmc.mc_sweep_many(chains, vpool, nsweeps=20, mode="sequential")
        dtn = time.perf_counter() - t0
        rows.append({"config": "the reference's CI system (test/runtests.jl:22-91): config_0, N=1290 2-D ternary JBB, T=0.231, VerletList dr=0.2, Displacement sigma=0.05; "
                               "the reference's sequential chain on engine 4 (state in LDS; per chain one resolver, twelve evaluators, one wavefront applying the list edits, one evaluating again)",
                     "reference_order_chain": {"moves_per_s_one_chain": 100 * one.N / dt1, "us_per_step_one_chain": 1e6 * dt1 / (100 * one.N),
                                               "kernel_moves_per_s_one_chain": 100 * one.N / (st1.sweep_kernel_ms * 1e-3),
                                               "moves_per_s_256_chains_one_launch": 256 * 20 * one.N / dtn,
                                               "moves_per_s_one_chain_literal_loop_engine_1": 4 * one.N / dtl,
                                               "steps_the_resolver_ran_alone": vinfo[1] / float(max(1, vinfo[3])),
                                               "list_rebuilds_per_step": vinfo[2] / float(max(1, vinfo[3])),
                                               "acceptance": [float(st1.accepted[0]) / max(1, st1.attempted[0])]},
                     "cpu_oracle_moves_per_s": _oracle_rate(lambda: O.System(cpos, csp, crho, 0.231, O.jbb(), O.VERLETLIST, 0.2),
                                                            O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}]), 10, budget_cpu_s)})
        del chains
    except Exception as exc:
        rows.append({"config": "the reference's CI system with a VerletList", "error": repr(exc)})
    # -- config 5: 8 tempering replicas of 2-D r^-14 soft spheres on this GPU, exchange every 10 sweeps
    n, side = 8, 256
    Nss = side * side
    L = (Nss / 0.5) ** 0.5
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    pos = (g + 0.5) * (L / side) + np.random.default_rng(1).uniform(-0.05, 0.05, (Nss, 2))
    sp = np.ones(Nss, dtype=np.int32)
    sp[np.random.default_rng(2).permutation(Nss)[: Nss // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, sg, 14) for sg in r_] for r_ in ([1.0, 1.2], [1.2, 1.4])]
    ladder = np.geomspace(1.0, 1.02, n)
    systems = [pmc.System(pos, sp, 0.5, ladder[r], mm, seed=5, chain=r) for r in range(n)]
    for s_ in systems:
        s_.total_energy()
    pool = [disp(0.065)]
    tempering.run_tempering_local(systems, pool, ladder, 20, 10, seed=9)
    t0 = time.perf_counter()
    label, acc, att, _ = tempering.run_tempering_local(systems, pool, ladder, 100, 10, seed=9)
    dt = time.perf_counter() - t0
    mo = O.model_matrix(lambda e, sg: O.soft_spheres(e, sg, 14), ((1.0, 1.0), (1.0, 1.0)), ((1.0, 1.2), (1.2, 1.4)))
    rows.append({"config": "C5: 8 replicas of 2-D binary r^-14 soft spheres, N=65536 each, rho=0.5, T=1.0..1.02, replica exchange every 10 sweeps (all on this GPU, shared launches)",
                 "checkerboard_m1": {"moves_per_s": n * Nss * 100 / dt, "exchange_acceptance": acc / max(1, att), "rounds": 1},
                 "cpu_oracle_moves_per_s": _oracle_rate(lambda: O.System(pos, sp, 0.5, 1.0, mo, O.LINKEDLIST),
                                                        O.make_pool([{"kind": "Displacement", "sigma": 0.065}]), 5, budget_cpu_s)})
    del systems
    # -- parallel = true: 256 chains of N = 4096 in shared launches (checkerboard schedule)
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
    chains = [pmc.System(pos, sp, rho, TEMPERATURE, pmc.LennardJones(1.0, 1.0), seed=42, chain=k) for k in range(256)]
    pool = [disp()]
    pmc.mc_sweep_many(chains, pool, nsweeps=50)
    t0 = time.perf_counter()
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=200)
    dt = time.perf_counter() - t0
    tests = sum(float(x.pair_tests) for x in sts)
    inr = sum(float(x.pairs_in_range) for x in sts)
    rows.append({"config": "256 chains (parallel = true) of N=4096 3D LJ rho=1 T=1, shared launches (pmc_sweep_many)",
                 "checkerboard_m1": {"moves_per_s": 256 * 4096 * 200 / dt, "rounds": 1,
                                     "roofline_frac": (tests + inr) * 9.0 / dt / 1e12 / FP64_VECTOR_PEAK_TFLOPS if tests > 0 else None,
                                     "acceptance": [sum(float(x.accepted[0]) for x in sts) / sum(float(x.attempted[0]) for x in sts)]}})
    # ... and the same kind of run on the reference's own (sequential) chain: 32 chains, speculative windows, shared rounds
    chains = chains[:32]
    for c_ in chains:
        c_.rounds = 1
    pmc.mc_sweep_many(chains, pool, nsweeps=2, mode="sequential")
    t0 = time.perf_counter()
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=20, mode="sequential")
    dt = time.perf_counter() - t0
    rate = 32 * 4096 * 20 / dt
    # ... and ONE such chain (a window of 64 steps, a round as one launch)
    pmc.mc_sweep(chains[0], pool, nsweeps=5, mode="sequential")
    t0 = time.perf_counter()
    st1 = pmc.mc_sweep(chains[0], pool, nsweeps=60, mode="sequential")
    dt1 = time.perf_counter() - t0
    info1 = chains[0].sequential_info()
    rows.append({"config": "32 chains (parallel = true) of N=4096 3D LJ rho=1 T=1 on the reference's sequential chain: speculative windows, every "
                           "launch of a round shared by the chains (pmc_sweep_many PMC_SEQUENTIAL); and one such chain alone",
                 "reference_order_chain": {"moves_per_s": rate, "acceptance": [sum(float(x.accepted[0]) for x in sts) / sum(float(x.attempted[0]) for x in sts)],
                                           "roofline_frac": _seq_flop_frac(4096, 3, chains[0].neighbour_list.ncells, rate, 62.0),
                                           "moves_per_s_one_chain": 4096 * 60 / dt1, "window_one_chain": int(info1[2]),
                                           "steps_per_round_one_chain": 4096 * 60 / max(1, int(info1[0])),
                                           "us_per_round_one_chain": 1e3 * float(st1.kernel_ms) / max(1, int(info1[0]))}})
    del chains
    # -- the reference's own benchmark system as many chains on its own (sequential) chain
    pos, sp, rho = lj_lattice(6, jitter=0.05, seed=42)
    chains = [pmc.System(pos, sp, rho, TEMPERATURE, pmc.LennardJones(1.0, 1.0), seed=42, chain=k) for k in range(1000)]
    pmc.mc_sweep_many(chains, pool, nsweeps=5, mode="sequential")
    t0 = time.perf_counter()
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=100, mode="sequential")
    dt = time.perf_counter() - t0
    rate = 1000 * 216 * 100 / dt
    rows.append({"config": "1000 chains (parallel = true) of the reference's benchmark system (benchmark/particles_benchmarks.jl:10-28: N=216 3D LJ rho=1), "
                           "the reference's sequential chain, state in LDS, one wavefront per chain (pmc_sweep_many PMC_SEQUENTIAL)",
                 "reference_order_chain": {"moves_per_s": rate, "us_per_step_per_chain": 1e6 * dt / (216 * 100),
                                           "acceptance": [sum(float(x.accepted[0]) for x in sts) / sum(float(x.attempted[0]) for x in sts)],
                                           "roofline_frac": _seq_flop_frac(216, 3, chains[0].neighbour_list.ncells, rate, 62.0)},
                 "cpu_oracle_moves_per_s": _oracle_rate(lambda: O.System(pos, sp, rho, TEMPERATURE, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST),
                                                        O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}]), 42, budget_cpu_s),
                 "reference_published": "5.1 us per mc_step! on the reference authors' CPU (benchmark/particles_benchmarks.jl:28) = 1.96e5 moves/s per chain"})
    # -- the system of the reference's ONLY published timing, as published: N = 216 3-D 50:50 BHHP soft spheres (n = 12),
    #    rho = 0.5342, T = 0.2, sigma_move = 0.1, EmptyList (all pairs, id order): 5.114 us per mc_step! on its authors' CPU
    from particlesmc_amd.workloads import random_gas
    posb, _ = random_gas(216, 0.5342, 3, seed=42, min_dist=0.85)
    spb = np.ones(216, dtype=np.int32)
    spb[np.random.default_rng(42).permutation(216)[:108]] = 2
    poolb = [disp(0.1)]
    one = pmc.System(posb, spb, 0.5342, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=42)
    pmc.mc_sweep(one, poolb, nsweeps=200, mode="sequential")
    t0 = time.perf_counter()
    stb = pmc.mc_sweep(one, poolb, nsweeps=1000, mode="sequential")
    dt1 = time.perf_counter() - t0
    vinfo5 = one.verlet_info()
    eng3 = pmc.System(posb, spb, 0.5342, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=42)
    eng3.set_sequential_engine(3, 0)
    pmc.mc_sweep(eng3, poolb, nsweeps=20, mode="sequential")
    t0 = time.perf_counter()
    pmc.mc_sweep(eng3, poolb, nsweeps=200, mode="sequential")
    dt3 = time.perf_counter() - t0
    c256 = [pmc.System(posb, spb, 0.5342, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=42, chain=k) for k in range(256)]
    pmc.mc_sweep_many(c256, poolb, nsweeps=5, mode="sequential")
    t0 = time.perf_counter()
    pmc.mc_sweep_many(c256, poolb, nsweeps=200, mode="sequential")
    dt256 = time.perf_counter() - t0
    del c256
    chains = [pmc.System(posb, spb, 0.5342, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=42, chain=k) for k in range(1000)]
    pmc.mc_sweep_many(chains, poolb, nsweeps=5, mode="sequential")
    t0 = time.perf_counter()
    pmc.mc_sweep_many(chains, poolb, nsweeps=100, mode="sequential")
    dtn = time.perf_counter() - t0
    mob = O.model_matrix(lambda e, sg: O.soft_spheres(e, sg, 12), ((1.0, 1.0), (1.0, 1.0)), ((1.0, 1.2), (1.2, 1.4)))
    cpu = _oracle_rate(lambda: O.System(posb, spb, 0.5342, 0.2, mob, O.EMPTYLIST), O.make_pool([{"kind": "Displacement", "sigma": 0.1}]), 42, budget_cpu_s)
    rows.append({"config": "the reference's published benchmark as published (benchmark/particles_benchmarks.jl:10-28): N=216 3D 50:50 BHHP soft spheres n=12, "
                           "rho=0.5342, T=0.2, sigma_move=0.1, EmptyList; the reference's sequential chain on engine 5 (state in LDS, one resolver and six or twelve "
                           "evaluators per chain, evaluations repaired term by term): one chain, 256 and 1000 chains in one launch; engine 3 (state in LDS, one step "
                           "after the other) beside it",
                 "reference_order_chain": {"us_per_step_one_chain": 1e6 * dt1 / (216 * 1000), "moves_per_s_one_chain": 216 * 1000 / dt1,
                                           "us_per_step_one_chain_engine_3": 1e6 * dt3 / (216 * 200),
                                           "moves_per_s_256_chains": 256 * 216 * 200 / dt256,
                                           "moves_per_s_1000_chains": 1000 * 216 * 100 / dtn,
                                           "steps_repaired": vinfo5[2] / float(max(1, vinfo5[3])), "steps_evaluated_again": vinfo5[1] / float(max(1, vinfo5[3])),
                                           "acceptance": [float(stb.accepted[0]) / max(1.0, float(stb.attempted[0]))]},
                 "cpu_oracle_moves_per_s": cpu, "cpu_oracle_us_per_step": 1e6 / cpu,
                 "reference_published": {"us_per_mc_step": 5.114, "us_per_step_in_mc_sweep": 1737.0 / 216, "hardware": "not stated (source comment)",
                                         "source": "benchmark/particles_benchmarks.jl:28-29"}})
    del chains
    # -- an EmptyList system too large for the LDS (the benchmark's list type at 50 x its size): all pairs in id order per step;
    #    engine 3 with the positions in global memory (round 5; the literal loop took 146 us per step)
    posl, spl, rhol = lj_lattice(22, jitter=0.05, seed=3)
    big = pmc.System(posl, spl, rhol, TEMPERATURE, pmc.LennardJones(1.0, 1.0), list_type=pmc.EmptyList, seed=5)
    t0 = time.perf_counter()
    stl = pmc.mc_sweep(big, [disp()], nsweeps=1, mode="sequential")
    dtl = time.perf_counter() - t0
    cpul = _oracle_rate(lambda: O.System(posl, spl, rhol, TEMPERATURE, O.single(O.lennard_jones(1.0, 1.0)), O.EMPTYLIST),
                        O.make_pool([{"kind": "Displacement", "sigma": SIGMA_MOVE}]), 5, min(budget_cpu_s, 1.0))
    rows.append({"config": "EmptyList beyond the LDS: N=10648 3D LJ rho=1 T=1, all pairs in id order every step (src/neighbours.jl:13-44); the reference's "
                           "sequential chain, positions in global memory, everything else in LDS, four wavefronts",
                 "reference_order_chain": {"us_per_step_one_chain": 1e6 * dtl / len(posl), "moves_per_s_one_chain": len(posl) / dtl,
                                           "acceptance": [float(stl.accepted[0]) / max(1.0, float(stl.attempted[0]))]},
                 "cpu_oracle_moves_per_s": cpul, "cpu_oracle_us_per_step": 1e6 / cpul})
    return rows


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
    rdev = f"cuda:{dev_index}" if backend == "nccl" else "cpu"

    from particlesmc_amd.workloads import lj_lattice
    pos, sp, rho = lj_lattice(args.lattice, jitter=0.05, seed=42)
    N = len(pos)
    pmc = None
    system = None
    if stub:
        engine = _StubEngine(N, rank)
        sweep = engine.sweep
        ncell_ref, grid = [1, 1, 1], [0, 0, 0]
        e0 = 0.0

        def total_energy():
            return 0.0, 0

        def set_rounds(m):
            pass
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

        def set_rounds(m):
            system.rounds = m

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

    def timed(nsteps):
        """EXACTLY `nsteps` sweeps bracketed by barrier + synchronize on both sides; elapsed = MAX over ranks."""
        barrier()
        t0 = time.perf_counter()
        st = sweep(nsteps)           # synchronises its stream before returning
        sync()
        el = time.perf_counter() - t0
        barrier()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return st, el

    def leg(st, el, nsteps, m):
        att = float(max(1, st.attempted[0]))
        return {"rounds": m, "value": float(N) * nsteps * ngpus / el, "ms_per_step": el * 1e3 / nsteps, "steps": nsteps,
                "acceptance": float(st.accepted[0]) / att, "null_move_fraction": float(getattr(st, "null_moves", 0)) / att}

    from_lattice = None
    if args.only_headline:
        set_rounds(1)
        sweep(args.equilibrate)
    elif args.equilibrate > 0:
        # the state rounds 1-2 timed, with their schedule (m = 1): the first sweeps from the jittered lattice; same
        # warm-up, same number of steps, same barriers and max-over-ranks as the headline region
        set_rounds(1)
        sweep(LATTICE_WARMUP)
        stl, tl = timed(LATTICE_STEPS)
        from_lattice = leg(stl, tl, LATTICE_STEPS, 1)
        from_lattice["warmup"] = LATTICE_WARMUP
        from_lattice["note"] = ("like-for-like with rounds 1-2, whatever --steps / --warmup / --rounds say: the first 5 + 20 sweeps "
                                "from the jittered lattice, m = 1 (r01 7.76e8, r02 first half 7.81e8, r02 final 1.04e9)")
        sweep(args.equilibrate)
    # the timed legs: the headline (m = --rounds) first, then the other schedules from the state it left
    legs = {}
    order = [args.rounds] + ([] if args.only_headline else [m for m in (1, 2, 4) if m != args.rounds])
    head_state = None
    st = None
    elapsed = None
    for m in order:
        set_rounds(m)
        if m == args.rounds and not stub and rank == 0 and ngpus == 1 and not args.no_cpu:
            head_state = system.position.copy()          # the liquid the headline region runs in: the CPU legs start from it too
        if args.warmup > 0:
            sweep(args.warmup)                           # (after the download: the timed region must not start on an idle GPU)
        stm, elm = timed(args.steps)
        legs[m] = leg(stm, elm, args.steps, m)
        if m == args.rounds:
            st, elapsed = stm, elm
    # call granularity: the same 64 sweeps as 64 calls of one sweep, 16 of four, one of 64 (every call ends with a counter fold
    # and a synchronisation; with m > 1 a call shorter than m sweeps is a shorter pass: DESIGN.md 2.2 "passes and calls")
    by_call = None
    if not args.only_headline and not stub:
        by_call = {}
        for m in sorted({1, args.rounds}):
            set_rounds(m)
            sweep(max(4, m))
            by_call[str(m)] = {}
            for per_call in (1, 4, 64):
                ncalls = 64 // per_call
                barrier()
                t0 = time.perf_counter()
                for _ in range(ncalls):
                    sweep(per_call)
                sync()
                el = time.perf_counter() - t0
                barrier()
                by_call[str(m)][str(per_call)] = float(N) * 64 * ngpus / el
        set_rounds(args.rounds)
    long_run = None
    if args.long_steps > 0 and not args.only_headline:
        set_rounds(args.rounds)
        stl2, ell2 = timed(args.long_steps)
        long_run = leg(stl2, ell2, args.long_steps, args.rounds)
        run_fix = stl2.energy_fix
    else:
        run_fix = None
    # exactness property at full size: the running energy equals a recomputation bit for bit
    if run_fix is None:
        run_fix = sweep(0).energy_fix if stub else system.total_energy()[1]
    e1, efix = total_energy()
    energy_ok = bool(efix == run_fix) if not stub else True
    if dist is not None:
        ok = torch.tensor([1.0 if energy_ok else 0.0], dtype=torch.float64, device=rdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        energy_ok = bool(ok.item() == 1.0)

    # the path's one collective, on the ranks of this job (N > 1 over RCCL only), in child processes
    # (--share-devices with PMC_RCCL_LIB naming another library with RCCL's entry points -- tests/c/rccl_stub.c -- runs the same
    # leg with every rank on one device: the whole rank path of an N-GPU run, minus RCCL's own transport, on a one-GPU box)
    rccl = None
    rccl_lib = os.environ.get("PMC_RCCL_LIB")
    if dist is not None and not stub and (backend == "nccl" or (args.share_devices and rccl_lib)):
        try:
            rccl = rccl_exchange_check(rank, world, dev_index, dist)
        except Exception as exc:
            rccl = {"n_ranks_rccl": 0, "error": repr(exc)}
        if isinstance(rccl, dict):
            rccl["transport"] = os.path.basename(rccl_lib) if rccl_lib else "librccl.so"

    if rank == 0:
        value = legs[args.rounds]["value"]
        launches = int(st.sweep_kernel_launches)
        timed_steps = args.steps
        if launches > 0 and st.sweep_kernel_ms > 0:
            avg_launch_s = st.sweep_kernel_ms * 1e-3 / launches
            moves_per_launch = N * timed_steps / launches
        else:
            launches = args.steps
            avg_launch_s = elapsed / max(1, launches)
            moves_per_launch = float(N)
        out = {
            "metric": "attempted MC moves/sec, N=10^6 3D LJ rho=1.0 (one sweep = N attempted moves)",
            "value": value, "unit": "attempted moves/s", "n_gpus": ngpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={N} 3D LJ rho=1.0 T=1.0 rcut=2.5 shifted, Displacement sigma=0.05, "
                                   + (f"liquid equilibrated by {args.equilibrate} untimed sweeps from the jittered {args.lattice}^3 lattice"
                                      if args.equilibrate > 0 else f"jittered {args.lattice}^3 lattice")
                                   + ", one independent chain per GPU",
                       "schedule": f"checkerboard, {args.rounds} sweeps per grid shift (rounds per cell visit, pmc_set_rounds)",
                       "rounds": args.rounds, "grid": grid, "reference_cells": ncell_ref,
                       "equilibration_sweeps": args.equilibrate,
                       "backend": backend or "single process", "ranks": world},
            "sweeps_per_s": args.steps * ngpus / elapsed,
            "acceptance": legs[args.rounds]["acceptance"],
            # cell-leaving proposals are rejected without any dE work; they are part of the schedule and count
            # as attempted moves, so the fraction is printed next to the headline
            "null_move_fraction": legs[args.rounds]["null_move_fraction"],
            "energy_per_particle": e1 / N, "energy_per_particle_initial": e0 / N,
            "running_energy_equals_recomputed": energy_ok,
            # the same liquid under the other schedules, each timed like the headline (barriers, max over ranks)
            "by_rounds": {str(m): legs[m] for m in sorted(legs)},
            "roofline": roofline_objects(N, timed_steps, st, avg_launch_s, moves_per_launch, ncell_ref),
        }
        # SURVEY 8(d)'s two HBM figures as first-class siblings of `roofline` (they also stay inside it): the contract figure
        # B_alg x moves / time -- not a bound of this kernel: a staged neighbourhood serves every slot of a visit from LDS -- and
        # the bytes rocprofv3 measured for the same launch
        rf = out["roofline"]
        out["roofline_hbm_contract"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                        "achieved": rf["contract_hbm_equiv"]["GBps_equivalent"],
                                        "frac": rf["contract_hbm_equiv"]["over_hbm_peak"],
                                        "algorithmic_bytes_per_move": rf["contract_hbm_equiv"]["algorithmic_bytes_per_move"],
                                        "note": rf["contract_hbm_equiv"]["note"]}
        if rf.get("traffic") is not None:
            out["hbm_measured"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_per_launch": rf["traffic"],
                                   "achieved": rf["traffic"] / avg_launch_s / 1e9, "frac": rf["hbm_measured_frac"]["value"],
                                   "source": rf.get("traffic_source")}
        # the like-for-like schedule (m = 1: one grid shift per sweep, the library's default) next to the headline's
        if 1 in legs:
            out["value_m1"] = legs[1]["value"]
        if by_call is not None:
            out["by_call_size"] = {"unit": "attempted moves/s over 64 sweeps issued as calls of 1 / 4 / 64 sweeps (rank 0's clock)",
                                   "by_rounds": by_call}
        if long_run is not None:
            out["long_run"] = long_run
        if from_lattice is not None:
            out["from_lattice"] = from_lattice
        if rccl is not None:
            out["rccl_exchange"] = rccl
            out["n_ranks_rccl"] = rccl.get("n_ranks_rccl", 0)
        if stub:
            out["stub"] = True
        if ngpus == 1 and not stub and not args.no_others and not args.only_headline:
            # the reference's own chain on the headline's system (PMC_SEQUENTIAL: bit-identical to cpu_baseline's chain), then the
            # other configurations; both leave the headline's figures above untouched (they run after them)
            try:
                out["reference_order_chain"] = reference_order_chain(pmc, system, pool)
                out["reference_order_chain"]["roofline_frac"] = _seq_flop_frac(N, 3, ncell_ref, out["reference_order_chain"]["value"], 62.0)
            except Exception as exc:
                out["reference_order_chain"] = {"error": repr(exc)}
            if not args.no_cpu:
                try:
                    out["other_configs"] = other_configs(pmc)
                except Exception as exc:
                    out["other_configs"] = {"error": repr(exc)}
        if ngpus == 1 and not args.no_cpu and not stub:
            out["cpu_baseline"] = cpu_baseline(head_state, sp, rho, 3_000_000)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["cpu_baseline"]["cpu_model"] = host_cpu_model()
            if isinstance(out.get("reference_order_chain"), dict) and "value" in out["reference_order_chain"]:
                out["reference_order_chain"]["over_cpu_baseline"] = out["reference_order_chain"]["value"] / out["cpu_baseline"]["value"]
            try:
                out["cpu_baseline_multicore"] = cpu_baseline_multicore(head_state, rho, out["cpu_baseline"]["value"], args.cpu_cores or None)
            except Exception as exc:      # the single-core figure is the contract; this one is a courtesy
                out["cpu_baseline_multicore"] = {"error": repr(exc)}
            try:
                eff = move_efficiency(pmc, sorted({1, args.rounds, 4}))
                out["move_efficiency_vs_sequential"] = {"value": eff["by_rounds"][str(args.rounds)]["msd_ratio"], "rounds": args.rounds, **eff}
                out["effective_sequential_moves_per_s"] = value * out["move_efficiency_vs_sequential"]["value"]
            except Exception as exc:
                out["move_efficiency_vs_sequential"] = {"error": repr(exc)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    status = 0
    if args.strict_rccl and dist is not None:
        good = [1.0 if (rank != 0 or (isinstance(rccl, dict) and rccl.get("n_ranks_rccl") == world and rccl.get("decisions_equal_pt_decide"))) else 0.0]
        t = torch.tensor(good, dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        status = 0 if float(t.item()) == 1.0 else 4
    if dist is not None:
        dist.destroy_process_group()
    return status


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv, args.launch_timeout)        # parent: no torch, no HIP; the ranks are child processes
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
