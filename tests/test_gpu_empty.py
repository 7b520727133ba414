"""-m gpu: engine 5 of the reference-order chain -- small EmptyList systems (src/neighbours.jl:13-44) with their state in LDS, one
wavefront resolving the steps in order while six or twelve others evaluate the steps behind them, and an evaluation that earlier steps
have overtaken REPAIRED (the pair terms of the particles stored in between replaced in its packed list, the ordered sums run
again) instead of discarded (csrc/pmc_empty.h) -- against the oracle's restatement of the reference's sequential chain, bit for
bit: positions, system.energy, accept counts.

The named system is the reference's published benchmark (benchmark/particles_benchmarks.jl:10-28: N = 216 3-D 50:50 BHHP soft
spheres n = 12, rho = 0.5342, T = 0.2, sigma_move = 0.1, the default EmptyList); test/ss142d.jl:8-21 (chains of N = 8) rides
along.  The seeded fuzz draws dimension, size, density, temperature, species and model matrices (every potential kind), step
widths up to a good part of the box, particles stored boxes away, the split into calls and the engine of each call."""
import os

import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice
from test_gpu_fuzz import _models

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("PMC_EMPTY_FUZZ_CASES", "40"))
FIRST = int(os.environ.get("PMC_EMPTY_FUZZ_FIRST", "0"))


def _disp(pmc, sigma, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def _benchmark_system():
    return binary_lattice(216, frac_b=0.5, rho=0.5342, jitter=0.05, seed=42)


@pytest.mark.parametrize("waves", [0, 8, 16])
def test_engine5_runs_the_reference_benchmark_system(gpu, waves):
    """benchmark/particles_benchmarks.jl:10-28 as published, on engine 5 (sixteen wavefronts per chain, eight, the library's
    choice), three calls: after each the chain is the oracle's; most steps are repaired evaluations, a few (their own particle had
    been stored since) are evaluated again."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _benchmark_system()
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=42)
    o = O.System(pos, sp, rho, 0.2, O.bhhp(), O.EMPTYLIST)
    g.set_sequential_engine(5, waves)
    pm = [_disp(pmc, 0.1)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.1}])
    done = 0
    for nsw in (20, 1, 35):
        st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        att, acc = o.mc_steps(po, 42, 0, done, nsw * N)
        done += nsw * N
        assert st.attempted[0] == att[0] == nsw * N and st.accepted[0] == acc[0], (nsw, st.accepted, acc)
        assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy, nsw
        polls, again, patched, steps = g.verlet_info()
        assert steps == nsw * N and patched > 0.5 * steps and again < 0.05 * steps, (polls, again, patched, steps)


def test_engine5_alternates_with_the_other_engines(gpu):
    """Engines 5, 3 (the chain's state in LDS, one step after the other), 1 (the literal loop) and the default selection on one
    handle, call after call; two Displacement widths in the pool."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _benchmark_system()
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.3, pmc.BHHP(), list_type=pmc.EmptyList, seed=7, chain=3)
    o = O.System(pos, sp, rho, 0.3, O.bhhp(), O.EMPTYLIST)
    pm = [_disp(pmc, 0.1, 0.6), _disp(pmc, 0.4, 0.4)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.1, "probability": 0.6}, {"kind": "Displacement", "sigma": 0.4, "probability": 0.4}])
    done = 0
    for engine in (5, 3, 0, 1, 5):
        g.set_sequential_engine(engine, 0)
        st = pmc.mc_sweep(g, pm, nsweeps=3, mode="sequential")
        att, acc = o.mc_steps(po, 7, 3, done, 3 * N)
        done += 3 * N
        assert np.array_equal(st.attempted[:2], att) and np.array_equal(st.accepted[:2], acc), (engine, st.accepted, acc)
        assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy, engine


def test_engine5_pools_with_swaps_run_on_engine_3(gpu):
    """A swap re-labels two particles -- every pair term of both changes its model: engine 5 takes pools of Displacements only.
    Asked for with a swap in the pool it is refused; the default selection runs the pool on engine 3."""
    from oracle import oracle as O
    from particlesmc_amd import _lib
    pmc = gpu
    pos, sp, rho = _benchmark_system()
    N = len(pos)
    nA, nB = int((sp == 1).sum()), int((sp == 2).sum())
    pm = [_disp(pmc, 0.1, 0.8), pmc.Move(pmc.DiscreteSwap(0, 0, (1, 2), (nA, nB), 0.0), pmc.DoubleUniform(), {}, 0.2)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.1, "probability": 0.8}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    g = pmc.System(pos, sp, rho, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=11)
    o = O.System(pos, sp, rho, 0.2, O.bhhp(), O.EMPTYLIST)
    g.set_sequential_engine(5, 0)
    with pytest.raises(_lib.PMCError) as ei:
        pmc.mc_sweep(g, pm, nsweeps=1, mode="sequential")
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED
    g.set_sequential_engine(0, 0)
    st = pmc.mc_sweep(g, pm, nsweeps=4, mode="sequential")
    att, acc = o.mc_steps(po, 11, 0, 0, 4 * N)
    assert np.array_equal(st.accepted[:2], acc)
    assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy and np.array_equal(g.species, o.species())
    # ... and an EmptyList system of thousands of particles is refused by engine 5 (its packed terms do not fit the LDS)
    pos2, sp2, rho2 = lj_lattice(14, jitter=0.05, seed=2, rho=0.8, d=3)
    g2 = pmc.System(pos2, sp2, rho2, 1.0, pmc.LennardJones(1.0, 1.0), list_type=pmc.EmptyList, seed=1)
    with pytest.raises(_lib.PMCError) as ei:
        g2.set_sequential_engine(5, 0)
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED


@pytest.mark.parametrize("nchains,n", [(9, 216), (100, 8), (300, 8), (520, 216)])
def test_engine5_chains_of_a_parallel_run(gpu, nchains, n):
    """pmc_sweep_many on EmptyList chains of one shape: one workgroup per chain in ONE launch of k_sq_empty, more chains than the
    chip has CUs (eight wavefronts per chain then, two chains to a CU) included -- the reference's published benchmark system, and test/ss142d.jl's chains of N = 8 (2-D soft spheres).
    Every chain against the oracle's."""
    from oracle import oracle as O
    pmc = gpu
    rng = np.random.default_rng(nchains)
    if n == 216:
        pos, sp, rho = _benchmark_system()
        d, T, sig = 3, 0.2, 0.1
    else:
        pos, sp, rho = binary_lattice(8, frac_b=0.5, rho=0.5, jitter=0.05, seed=3, d=2)
        d, T, sig = 2, 1.0, 0.065
    N = len(pos)
    starts = [pos + rng.uniform(-0.02, 0.02, size=pos.shape) for _ in range(nchains)]
    gs = [pmc.System(starts[k], sp, rho, T, pmc.BHHP(), list_type=pmc.EmptyList, seed=42, chain=k) for k in range(nchains)]
    pm = [_disp(pmc, sig)]
    po = O.make_pool([{"kind": "Displacement", "sigma": sig}])
    nsw = 5 if n == 216 else 40
    sts = pmc.mc_sweep_many(gs, pm, nsweeps=nsw, mode="sequential")
    check = range(nchains) if nchains <= 20 else list(range(0, nchains, 17)) + [nchains - 1]
    for k in check:
        o = O.System(starts[k], sp, rho, T, O.bhhp(), O.EMPTYLIST)
        att, acc = o.mc_steps(po, 42, k, 0, nsw * N)
        assert sts[k].accepted[0] == acc[0], (k, sts[k].accepted, acc)
        assert np.array_equal(gs[k].position, o.positions()) and gs[k].energy[0] == o.energy, k
    assert gs[0].verlet_info()[3] == nsw * N               # engine 5 ran the call


def _draw(pmc, O, case):
    for attempt in range(50):
        rng = np.random.default_rng(515000 + case + 100000 * attempt)
        d = int(rng.choice([2, 3]))
        n = int(rng.integers(2, 27)) if d == 2 else int(rng.integers(2, 9))
        rho = float(np.exp(rng.uniform(np.log(0.05), np.log(1.3))))
        S = int(rng.choice([1, 2, 3]))
        pos, sp, _ = lj_lattice(n, jitter=0.08 * rho ** (-1.0 / d), seed=case, rho=rho, d=d)
        N = len(pos)
        if S > 1:
            sp = rng.integers(1, S + 1, size=N).astype(np.int32)
            sp[:min(S, N)] = np.arange(1, min(S, N) + 1)
        far = int(rng.choice([0, 1, 3]))
        if far:
            pos = pos + rng.integers(-far, far + 1, size=pos.shape) * (N / rho) ** (1.0 / d)
        T = float(np.exp(rng.uniform(np.log(0.1), np.log(5.0))))
        seed, chain = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 8))
        mm, mo = _models(pmc, O, rng, S)
        try:
            O.System(pos, sp, rho, T, mo, O.EMPTYLIST)
        except ValueError as e:
            assert "bad arguments" in str(e), e
            continue
        return rng, d, S, pos, sp, rho, T, seed, chain, mm, mo
    raise AssertionError(f"case {case}: no acceptable geometry in 50 draws")


@pytest.mark.parametrize("case", range(FIRST, FIRST + NCASES))
def test_random_empty_list_chain(gpu, case):
    """Seeded random EmptyList chains under Displacement pools (one or two widths, up to a good part of the box): 4 to 676
    particles, 2-D and 3-D, dilute to dense, frozen to hot, 1-3 species with random model matrices.  One to three calls, the
    engine drawn per call (5 with eight, sixteen or the library's choice of wavefronts; the default; 3; and 1 for the smallest)."""
    from oracle import oracle as O
    pmc = gpu
    rng, d, S, pos, sp, rho, T, seed, chain, mm, mo = _draw(pmc, O, case)
    N = len(pos)
    g = pmc.System(pos, sp, rho, T, mm, list_type=pmc.EmptyList, seed=seed, chain=chain)
    o = O.System(pos, sp, rho, T, mo, O.EMPTYLIST)
    assert g.energy[0] == o.energy
    box = (N / rho) ** (1.0 / d)
    sig1 = float(rng.uniform(0.02, 0.3)) if rng.random() < 0.6 else float(rng.uniform(0.3, 0.4 * box))
    pm, po = [_disp(pmc, sig1)], [{"kind": "Displacement", "sigma": sig1}]
    if rng.random() < 0.4:
        sig2 = float(rng.uniform(0.01, 0.2 * box))
        pm = [_disp(pmc, sig1, 0.7), _disp(pmc, sig2, 0.3)]
        po = [{"kind": "Displacement", "sigma": sig1, "probability": 0.7}, {"kind": "Displacement", "sigma": sig2, "probability": 0.3}]
    pool_o = O.make_pool(po)
    done = 0
    for call in range(int(rng.integers(1, 4))):
        engine = int(rng.choice([5, 5, 0, 3, 1]))
        nsw = int(rng.integers(1, 12))
        if engine == 1 and N * nsw > 1500:
            engine = 0
        g.set_sequential_engine(engine, int(rng.choice([0, 8, 16])) if engine == 5 else 0)
        st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        att, acc = o.mc_steps(pool_o, seed, chain, done, nsw * N)
        done += nsw * N
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc), (call, engine, st.accepted, acc)
        assert np.array_equal(g.position, o.positions()), (call, engine)
        assert g.energy[0] == o.energy, (call, engine, g.energy[0], o.energy)


def test_engine5_wavefronts_per_chain_are_eight_or_sixteen(gpu):
    """pmc_set_sequential_engine(h, 5, window): with engine 5 the window is the number of wavefronts per chain -- 0 (the library's
    choice), 8 or 16; anything else is a bad argument and leaves the selection as it was."""
    from oracle import oracle as O
    from particlesmc_amd import _lib
    pmc = gpu
    pos, sp, rho = _benchmark_system()
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.2, pmc.BHHP(), list_type=pmc.EmptyList, seed=5)
    o = O.System(pos, sp, rho, 0.2, O.bhhp(), O.EMPTYLIST)
    g.set_sequential_engine(5, 8)
    for bad in (4, 12, 64):
        with pytest.raises(_lib.PMCError) as ei:
            g.set_sequential_engine(5, bad)
        assert ei.value.code == _lib.PMC_ERR_BADARG
    st = pmc.mc_sweep(g, [_disp(pmc, 0.1)], nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.1}]), 5, 0, 0, 3 * N)
    assert st.accepted[0] == acc[0] and np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy
    assert g.verlet_info()[3] == 3 * N                     # (engine 5 ran the call)
