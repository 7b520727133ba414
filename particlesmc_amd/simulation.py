"""Sweep-level driver: the accelerated analogue of Arianna's mc_sweep! / Metropolis / run!.

The reference reaches the hot path through Arianna (not in its tree):
`run!(Simulation(chains, algorithm_list, steps))` (src/ParticlesMC.jl:294-297) and
`mc_sweep!(system, pool, rng; mc_steps)` (benchmark/particles_benchmarks.jl:29).
Only what the hot path needs on either side is restated here: the Metropolis
algorithm entry, the schedulers and the output algorithms that read
`system.energy`, acceptance counters and frames (src/ParticlesMC.jl:249-291).
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from .moves import DiscreteSwap, Displacement, Move


def c_pool(pool):
    arr = (_lib.Move * len(pool))()
    for k, mv in enumerate(pool):
        if isinstance(mv.action, Displacement):
            arr[k].kind = _lib.MOVE_DISPLACEMENT
            arr[k].sigma = mv.sigma
        elif isinstance(mv.action, DiscreteSwap):
            arr[k].kind = _lib.MOVE_SWAP
            arr[k].species[0], arr[k].species[1] = mv.action.species
        else:
            raise TypeError(f"unsupported action {type(mv.action).__name__}")
        arr[k].probability = mv.probability
    return arr


class SweepStats:
    def __init__(self, st, npool):
        self.attempted = np.array(st.attempted[:npool], dtype=np.int64)
        self.accepted = np.array(st.accepted[:npool], dtype=np.int64)
        self.energy = st.energy
        self.energy_fix = st.energy_fix
        self.kernel_ms = st.kernel_ms
        self.sweep_kernel_ms = st.sweep_kernel_ms
        self.sweep_kernel_launches = st.sweep_kernel_launches
        self.pair_tests, self.pairs_in_range, self.null_moves = st.pair_tests, st.pairs_in_range, st.null_moves
        self.dense_evals, self.staged_candidates, self.written_back = st.dense_evals, st.staged_candidates, st.written_back


MODES = {"sequential": _lib.SEQUENTIAL, "checkerboard": _lib.CHECKERBOARD}


def mc_sweep(system, pool, rng=None, mc_steps=None, mode="checkerboard", nsweeps=None):
    """mc_sweep!(system, pool, rng; mc_steps) on the device.

    mc_steps must be a multiple of N (sweepstep = N, src/ParticlesMC.jl:246).  `rng` is
    accepted for signature compatibility; the device consumes the Philox streams keyed by
    the system's (seed, chain) and its step / sweep counters."""
    if nsweeps is None:
        nsweeps = 1 if mc_steps is None else mc_steps // system.N
        if mc_steps is not None and mc_steps % system.N:
            raise ValueError("mc_steps must be a multiple of length(system)")
    st = _lib.Stats()
    arr = c_pool(pool)
    _lib.check(_lib.lib().pmc_sweep(system._h, MODES[mode], arr, len(pool), int(nsweeps), C.byref(st)), system._h)
    return SweepStats(st, len(pool))


def mc_sweep_many(systems, pool, nsweeps=1, mode="checkerboard"):
    """The chains of a `parallel=true` run (src/ParticlesMC.jl:164,246) through pmc_sweep_many: chains of one shape
    share every kernel launch; otherwise every chain's sweeps are enqueued on its own HIP stream before any is waited for."""
    arr = c_pool(pool)
    L = _lib.lib()
    n = len(systems)
    if n == 0:
        return []
    hs = (C.c_void_p * n)(*[s._h for s in systems])
    st = (_lib.Stats * n)()
    # pmc_sweep_many: chains of one shape share every launch; anything else runs as begin ... end on every handle
    # (and the library ends every chain it has begun whatever happens to the others)
    _lib.check(L.pmc_sweep_many(hs, n, MODES[mode], arr, len(pool), int(nsweeps), st), systems[0]._h)
    return [SweepStats(st[k], len(pool)) for k in range(n)]


# ---- schedulers (Arianna build_schedule; semantics not pinned by the reference tree) -----
def build_schedule(steps, burn, block):
    if isinstance(block, (int, np.integer)):
        return list(range(burn, steps + 1, int(block)))
    block = [int(b) for b in block]
    span = max(block[-1], 1)
    out = []
    t0 = burn
    while t0 <= steps:
        for b in block:
            if t0 + b <= steps:
                out.append(t0 + b)
        t0 += span
    return sorted(set(out))


def energy(system):
    """@callback energy(system) = system.energy[1] / length(system) -- src/utils.jl:51-53."""
    return system.energy[0] / len(system)


CALLBACKS = {"energy": energy}


class Metropolis:
    def __init__(self, pool, seed=0, parallel=False, sweepstep=None, mode="checkerboard"):
        self.pool, self.seed, self.parallel, self.sweepstep, self.mode = pool, seed, parallel, sweepstep, mode


class Simulation:
    """Simulation(chains, algorithm_list, steps; path, verbose) -- src/ParticlesMC.jl:294."""

    def __init__(self, chains, algorithm_list, steps, path="./", verbose=False):
        self.chains, self.algorithm_list, self.steps, self.path, self.verbose = chains, algorithm_list, steps, path, verbose
        self.attempted = None
        self.accepted = None


def _write_frame(io, system, t, fmt, digits=6):
    from .io import store_trajectory
    store_trajectory(io, system, t, fmt, digits=digits)


def run(sim):
    """run!(simulation): sweeps on the device between scheduler times, outputs on the host."""
    metro = next(a for a in sim.algorithm_list if a.get("algorithm") == "Metropolis")
    pool = metro["pool"]
    mode = metro.get("mode", "checkerboard")
    outputs = [a for a in sim.algorithm_list if a.get("algorithm") != "Metropolis"]
    times = sorted(set(t for a in outputs for t in a["scheduler"] if 0 <= t <= sim.steps) | {sim.steps})
    npool = len(pool)
    att = np.zeros((len(sim.chains), npool), dtype=np.int64)
    acc = np.zeros((len(sim.chains), npool), dtype=np.int64)
    files = {}

    def fh(path, mode_="a"):
        if path not in files:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            files[path] = open(path, "w")
        return files[path]

    t_now = 0
    for t in [0] + [x for x in times if x > 0] if 0 in times else times:
        if t > t_now:
            for c, st in enumerate(mc_sweep_many(sim.chains, pool, nsweeps=t - t_now, mode=mode)):
                att[c] += st.attempted
                acc[c] += st.accepted
            t_now = t
        for a in outputs:
            if t not in a["scheduler"]:
                continue
            alg = a["algorithm"]
            for c, system in enumerate(sim.chains):
                cdir = os.path.join(sim.path, "chains", str(c + 1))
                if alg == "StoreCallbacks":
                    for cb in a["callbacks"]:
                        f = fh(os.path.join(cdir, f"{cb.__name__}.dat"))
                        f.write(f"{t} {float(cb(system))!r}\n")
                elif alg == "StoreTrajectories":
                    fmt = a["fmt"]
                    f = fh(os.path.join(cdir, "trajectory" + fmt.extension))
                    _write_frame(f, system, t, fmt)
                elif alg == "StoreLastFrames":
                    fmt = a["fmt"]
                    path = os.path.join(cdir, "lastframe" + fmt.extension)
                    os.makedirs(cdir, exist_ok=True)
                    with open(path, "w") as f:
                        _write_frame(f, system, t, fmt)
            if alg == "StoreAcceptance":
                for m in range(npool):
                    f = fh(os.path.join(sim.path, "moves", str(m + 1), "acceptance.dat"))
                    tot_att, tot_acc = att[:, m].sum(), acc[:, m].sum()
                    f.write(f"{t} {float(tot_acc) / float(tot_att) if tot_att else 0.0!r}\n")
            elif alg == "PrintTimeSteps" and sim.verbose:
                print(f"t = {t}", file=sys.stderr)
    for f in files.values():
        f.close()
    sim.attempted, sim.accepted = att, acc
    return sim
