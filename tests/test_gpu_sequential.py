"""-m gpu: the FAST reference-order chain (PMC_SEQUENTIAL, engine 2: speculative windows, pmc_seq.h) against the oracle's
restatement of the reference's sequential chain (src/moves.jl:11-20,57-81,187-207; oracle orc_mc_steps_sequential).

Bar: bit-exact -- positions, cell membership, head / list (or the CellList vectors), SpeciesList, accept counts per move
and the running system.energy[1] -- whatever the window, the list type, the pool and the grouping into calls."""
import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice, random_gas

pytestmark = pytest.mark.gpu


def _disp(pmc, sigma, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def _swap(pmc, species, counts, p):
    return pmc.Move(pmc.DiscreteSwap(0, 0, species, counts, 0.0), pmc.DoubleUniform(), {}, p)


def _same_state(g, o, linked=True, species=False):
    assert np.array_equal(g.position, o.positions()), "positions differ"
    assert g.energy[0] == o.energy, (g.energy[0], o.energy)
    assert np.array_equal(g.get_cells(), o.cells()), "cs differ"
    if linked:
        hg, lg = g.get_linked_list()
        ho, lo = o.linked_list()
        assert np.array_equal(hg, ho), "heads differ"
        assert np.array_equal(lg, lo), "list differs"
    if species:
        assert np.array_equal(g.species, o.species())
        for s in (1, 2):
            assert np.array_equal(g.species_ids(s), o.species_ids(s)), f"sp_ids[{s}] differ"


@pytest.mark.parametrize("n,rho,sigma,window", [(6, 0.9, 0.15, 0), (8, 0.9, 0.15, 64), (12, 1.0, 0.05, 0), (16, 1.0, 0.3, 128),
                                                (24, 1.0, 0.05, 0), (24, 1.0, 0.6, 512)])
def test_lj_displacements_bitexact(gpu, n, rho, sigma, window):
    """Displacement-only LJ chains from 3 cells per side (everything interacts) to 9 (windows of hundreds of steps);
    sigma up to 0.6 makes a quarter of the trials change cell, accepted or not."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(n, jitter=0.1, seed=3, rho=rho)
    g = pmc.System(pos, sp, rho, 0.8, pmc.LennardJones(1.0, 1.0), seed=99)
    o = O.System(pos, sp, rho, 0.8, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    g.set_sequential_engine(2, window)
    nsw = 3 if n <= 16 else 2
    st = pmc.mc_sweep(g, [_disp(pmc, sigma)], nsweeps=nsw, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": sigma}]), 99, 0, 0, nsw * len(pos))
    assert st.attempted[0] == att[0] and st.accepted[0] == acc[0], (st.accepted, acc)
    _same_state(g, o)
    rounds, evals, win, folded = g.sequential_info()
    assert folded == nsw * len(pos) and rounds > 0


def test_engines_agree_and_calls_compose(gpu):
    """The literal one-wavefront loop (engine 1), the windows (engine 2) and the oracle are one chain; splitting the sweeps
    over calls, or alternating the engines between calls, changes nothing."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(10, jitter=0.1, seed=5, rho=0.95)
    pool = [_disp(pmc, 0.2)]
    o = O.System(pos, sp, rho, 1.1, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.2}]), 7, 0, 0, 4 * len(pos))
    a = pmc.System(pos, sp, rho, 1.1, pmc.LennardJones(1.0, 1.0), seed=7)
    a.set_sequential_engine(2, 0)
    sa = pmc.mc_sweep(a, pool, nsweeps=4, mode="sequential")
    b = pmc.System(pos, sp, rho, 1.1, pmc.LennardJones(1.0, 1.0), seed=7)
    nacc = 0
    for k, eng in enumerate((2, 1, 2, 1)):
        b.set_sequential_engine(eng, 64 if k == 0 else 0)
        nacc += pmc.mc_sweep(b, pool, nsweeps=1, mode="sequential").accepted[0]
    assert sa.accepted[0] == acc[0] == nacc
    _same_state(a, o)
    _same_state(b, o)


@pytest.mark.parametrize("N,window", [(1000, 0), (4096, 256)])
def test_kob_andersen_with_swaps_bitexact(gpu, N, window):
    """BASELINE config 3's pool (90 % Displacement, 10 % DiscreteSwap) on the sequential chain: species, SpeciesList,
    both accept counts."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(N, frac_b=0.2, rho=1.2, jitter=0.05, seed=8)
    nA, nB = int((sp == 1).sum()), int((sp == 2).sum())
    g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=21)
    o = O.System(pos, sp, rho, 1.0, O.kob_andersen(), O.LINKEDLIST)
    g.set_sequential_engine(2, window)
    pool = [_disp(pmc, 0.08, 0.9), _swap(pmc, (1, 2), (nA, nB), 0.1)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.08, "probability": 0.9},
                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}])
    for call in range(2):
        st = pmc.mc_sweep(g, pool, nsweeps=2, mode="sequential")
        att, acc = o.mc_steps(opool, 21, 0, call * 2 * N, 2 * N)
        assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc), (st.accepted, acc)
        _same_state(g, o, species=True)


def test_high_swap_acceptance_mixture(gpu):
    """The published mixture (examples/lj-mixture: size ratio 1.06, rcut = 4, unshifted) accepts most swaps: windows full
    of accepted DiscreteSwaps; the reference's own N = 1000 has two cells per side."""
    from oracle import oracle as O
    from mixture import EPS, SIG, RCUT
    pmc = gpu
    N = 1000
    pos, sp, rho = binary_lattice(N, frac_b=0.5, rho=0.8, jitter=0.05, seed=4)
    mp = [[pmc.LennardJones(EPS[i][j], SIG[i][j], rcut=RCUT, shift_potential=False) for j in range(2)] for i in range(2)]
    mo = O.model_matrix(lambda e, s: O.lennard_jones(e, s, RCUT, False), EPS, SIG)
    g = pmc.System(pos, sp, rho, 1.5, mp, seed=5)
    o = O.System(pos, sp, rho, 1.5, mo, O.LINKEDLIST)
    g.set_sequential_engine(2, 0)
    pool = [_disp(pmc, 0.1, 0.5), _swap(pmc, (1, 2), (N // 2, N // 2), 0.5)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.1, "probability": 0.5},
                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.5}])
    st = pmc.mc_sweep(g, pool, nsweeps=2, mode="sequential")
    att, acc = o.mc_steps(opool, 5, 0, 0, 2 * N)
    assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc), (st.accepted, acc)
    assert acc[1] > 0.3 * att[1]
    _same_state(g, o, species=True)


def test_cell_list_vectors_bitexact(gpu):
    """CellList (src/neighbours.jl:61-217): cells[c] vectors in filter! / push! order, energies summed in vector order."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.1, seed=11, rho=0.9)
    g = pmc.System(pos, sp, rho, 0.9, pmc.LennardJones(1.0, 1.0), list_type=pmc.CellList, seed=3)
    o = O.System(pos, sp, rho, 0.9, O.single(O.lennard_jones(1.0, 1.0)), O.CELLLIST)
    g.set_sequential_engine(2, 0)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.4)], nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.4}]), 3, 0, 0, 3 * len(pos))
    assert st.accepted[0] == acc[0]
    _same_state(g, o, linked=False)
    ncell = int(np.prod(o.cell_geometry()[1]))
    for c in range(1, ncell + 1):
        assert np.array_equal(g.get_cell_members(c), o.cell_members(c)), c


def test_soft_spheres_2d_unwrapped(gpu):
    """2-D r^-12 binary soft spheres (BHHP) from positions many boxes away (never wrapped: src/moves.jl:46-48)."""
    from oracle import oracle as O
    pmc = gpu
    N = 900
    pos, rho = random_gas(N, 0.7, 2, seed=2, min_dist=0.9)
    rng = np.random.default_rng(1)
    sp = np.ones(N, dtype=np.int32)
    sp[rng.permutation(N)[: N // 2]] = 2
    g = pmc.System(pos, sp, rho, 0.5, pmc.BHHP(), seed=13)
    o = O.System(pos, sp, rho, 0.5, O.bhhp(), O.LINKEDLIST)
    g.set_sequential_engine(2, 0)
    pool = [_disp(pmc, 0.15, 0.8), _swap(pmc, (1, 2), (N // 2, N - N // 2), 0.2)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.15, "probability": 0.8},
                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    st = pmc.mc_sweep(g, pool, nsweeps=4, mode="sequential")
    att, acc = o.mc_steps(opool, 13, 0, 0, 4 * N)
    assert np.array_equal(st.accepted, acc), (st.accepted, acc)
    _same_state(g, o, species=True)


def test_config2_one_sweep_bitexact(gpu):
    """BASELINE config 2 (N = 32768 LJ liquid, rho = 1, T = 1, rcut = 2.5): one full sweep of the sequential chain."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(32, jitter=0.05, seed=42)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
    o = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=1, mode="sequential")     # engine: auto
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 42, 0, 0, len(pos))
    assert st.accepted[0] == acc[0]
    _same_state(g, o)


# ---- engine 3: the chain's state in LDS, one wavefront per chain, many chains per launch ------------------------------------------

@pytest.mark.parametrize("waves", [1, 4])
@pytest.mark.parametrize("n,rho,sigma", [(6, 0.9, 0.15), (6, 1.0, 0.6), (8, 0.9, 0.3), (10, 0.95, 0.1)])
def test_lds_engine_lj_bitexact(gpu, n, rho, sigma, waves, monkeypatch):
    """The reference's benchmark system (benchmark/particles_benchmarks.jl:10-28: N = 216 LJ) and its neighbours in size,
    engine 3 against the oracle; sigma = 0.6 makes a quarter of the trials change cell."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(n, jitter=0.1, seed=3, rho=rho)
    g = pmc.System(pos, sp, rho, 0.8, pmc.LennardJones(1.0, 1.0), seed=99)
    o = O.System(pos, sp, rho, 0.8, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    g.set_sequential_engine(3, 0)
    monkeypatch.setenv("PMC_SEQ_WAVES", str(waves))         # wavefronts per chain (the library picks by the number of chains)
    for call in range(2):
        st = pmc.mc_sweep(g, [_disp(pmc, sigma)], nsweeps=3, mode="sequential")
        att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": sigma}]), 99, 0, call * 3 * len(pos), 3 * len(pos))
        assert st.attempted[0] == att[0] and st.accepted[0] == acc[0], (st.accepted, acc)
        _same_state(g, o)


@pytest.mark.parametrize("waves", [1, 4])
def test_lds_engine_mixture_with_swaps_and_cell_list(gpu, waves, monkeypatch):
    """Engine 3 on the published mixture's own size (N = 1000, rcut = 4: two cells per side, swaps mostly accepted) with a
    LinkedList, and on a CellList system."""
    from oracle import oracle as O
    from mixture import EPS, SIG, RCUT
    pmc = gpu
    monkeypatch.setenv("PMC_SEQ_WAVES", str(waves))
    N = 1000
    pos, sp, rho = binary_lattice(N, frac_b=0.5, rho=0.8, jitter=0.05, seed=4)
    mp = [[pmc.LennardJones(EPS[i][j], SIG[i][j], rcut=RCUT, shift_potential=False) for j in range(2)] for i in range(2)]
    mo = O.model_matrix(lambda e, s: O.lennard_jones(e, s, RCUT, False), EPS, SIG)
    g = pmc.System(pos, sp, rho, 1.5, mp, seed=5)
    o = O.System(pos, sp, rho, 1.5, mo, O.LINKEDLIST)
    g.set_sequential_engine(3, 0)
    pool = [_disp(pmc, 0.1, 0.5), _swap(pmc, (1, 2), (N // 2, N // 2), 0.5)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.1, "probability": 0.5},
                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.5}])
    st = pmc.mc_sweep(g, pool, nsweeps=2, mode="sequential")
    att, acc = o.mc_steps(opool, 5, 0, 0, 2 * N)
    assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc), (st.accepted, acc)
    _same_state(g, o, species=True)
    pos, sp, rho = lj_lattice(8, jitter=0.1, seed=11, rho=0.9)
    g = pmc.System(pos, sp, rho, 0.9, pmc.LennardJones(1.0, 1.0), list_type=pmc.CellList, seed=3)
    o = O.System(pos, sp, rho, 0.9, O.single(O.lennard_jones(1.0, 1.0)), O.CELLLIST)
    g.set_sequential_engine(3, 0)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.4)], nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.4}]), 3, 0, 0, 3 * len(pos))
    assert st.accepted[0] == acc[0]
    _same_state(g, o, linked=False)
    for c in range(1, int(np.prod(o.cell_geometry()[1])) + 1):
        assert np.array_equal(g.get_cell_members(c), o.cell_members(c)), c


def test_many_sequential_chains_in_one_launch(gpu):
    """pmc_sweep_many(PMC_SEQUENTIAL): the chains of a `parallel = true` run (src/ParticlesMC.jl:164,246) on the reference's
    own chain, one wavefront per chain in shared launches; every chain equals the oracle's chain of its (seed, chain) and
    temperature, and the same chain run alone."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(6, jitter=0.1, seed=1, rho=0.9)
    nch = 24
    temps = [0.7 + 0.05 * k for k in range(nch)]
    chains = [pmc.System(pos, sp, rho, temps[k], pmc.LennardJones(1.0, 1.0), seed=17, chain=k) for k in range(nch)]
    pool = [_disp(pmc, 0.2)]
    stats = pmc.mc_sweep_many(chains, pool, nsweeps=4, mode="sequential")
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.2}])
    for k in (0, 1, 7, nch - 1):
        o = O.System(pos, sp, rho, temps[k], O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
        att, acc = o.mc_steps(opool, 17, k, 0, 4 * len(pos))
        assert stats[k].accepted[0] == acc[0], k
        _same_state(chains[k], o)
    alone = pmc.System(pos, sp, rho, temps[5], pmc.LennardJones(1.0, 1.0), seed=17, chain=5)
    st = pmc.mc_sweep(alone, pool, nsweeps=4, mode="sequential")
    assert st.accepted[0] == stats[5].accepted[0] and np.array_equal(alone.position, chains[5].position)
    assert alone.energy[0] == chains[5].energy[0]


@pytest.mark.parametrize("engine", [0, 1, 2, 3])
def test_config1_readme_three_particles(gpu, engine):
    """BASELINE config 1: the README's literal system (README.md:72-110: three 2-D LJ particles, density 1, T = 1, sigma_move
    0.05, 500 Metropolis steps).  Its box (sqrt 3) is smaller than rcut = 2.5, which the LinkedList constructor cannot take
    (src/neighbours.jl:237-239: ncells = 0; tests/test_abi.py checks the refusal), so the literal numbers run with the
    reference's EmptyList, and the LinkedList form with the largest legal cutoff (one cell per side).  500 sweeps = 1500 steps
    of the sequential chain against the oracle, every engine that accepts the list."""
    from oracle import oracle as O
    pmc = gpu
    pos = np.array([[0.1585, 0.4965], [1.7215, 0.7468], [0.7606, 1.1439]])
    sp = np.ones(3, dtype=np.int32)
    cases = [(pmc.LinkedList, O.LINKEDLIST, 1.7)]
    if engine in (0, 1):
        cases.append((pmc.EmptyList, O.EMPTYLIST, 2.5))
    for lt_p, lt_o, rcut in cases:
        g = pmc.System(pos, sp, 1.0, 1.0, pmc.LennardJones(1.0, 1.0, rcut=rcut), list_type=lt_p, seed=10)
        o = O.System(pos, sp, 1.0, 1.0, O.single(O.lennard_jones(1.0, 1.0, rcut)), lt_o)
        assert g.energy[0] == o.energy
        g.set_sequential_engine(engine, 0)
        st = pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=500, mode="sequential")
        att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 10, 0, 0, 1500)
        assert st.attempted[0] == att[0] == 1500 and st.accepted[0] == acc[0]
        assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy


def test_many_larger_chains_interleaved_windows(gpu):
    """pmc_sweep_many(PMC_SEQUENTIAL) for chains too large for the LDS engine: every chain runs the speculative windows on its
    own stream and all chains' batches are enqueued before any is waited for.  Every chain equals the oracle's chain of its
    (seed, chain) and temperature."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(16, jitter=0.08, seed=2, rho=0.95)          # N = 4096: 98 KB of positions
    nch = 5
    temps = [0.8 + 0.1 * k for k in range(nch)]
    chains = [pmc.System(pos, sp, rho, temps[k], pmc.LennardJones(1.0, 1.0), seed=23, chain=k) for k in range(nch)]
    pool = [_disp(pmc, 0.1)]
    stats = pmc.mc_sweep_many(chains, pool, nsweeps=2, mode="sequential")
    assert chains[0].sequential_info()[0] > 0                              # engine 2 ran (rounds of windows)
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.1}])
    for k in range(nch):
        o = O.System(pos, sp, rho, temps[k], O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
        att, acc = o.mc_steps(opool, 23, k, 0, 2 * len(pos))
        assert stats[k].accepted[0] == acc[0] and stats[k].attempted[0] == att[0], k
        _same_state(chains[k], o)


@pytest.mark.parametrize("engine", [2, 3])
def test_dilute_gas_and_dense_cluster(gpu, engine):
    """Edges of the window logic: a dilute gas (many more cells than particles: the window is larger than the number of
    particles, so admission is cut by repeated particles all the time) and a system with one crowded region (rows of very
    different lengths), large displacements so that most trials change cell."""
    from oracle import oracle as O
    pmc = gpu
    pos, rho = random_gas(300, 0.02, 3, seed=4, min_dist=1.0)
    rng = np.random.default_rng(3)
    pos[:60] = pos[0] + rng.uniform(-1.6, 1.6, (60, 3)) * np.linspace(0.4, 1.0, 60)[:, None]     # a droplet around particle 0
    sp = np.ones(len(pos), dtype=np.int32)
    T = 2.0
    try:
        o = O.System(pos, sp, rho, T, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    except ValueError:
        pytest.skip("the droplet has overlapping particles for this seed")
    g = pmc.System(pos, sp, rho, T, pmc.LennardJones(1.0, 1.0), seed=31)
    assert g.energy[0] == o.energy
    if engine == 3 and int(np.prod(g.neighbour_list.ncells)) * 4 > 60000:
        pytest.skip("too many cells for the LDS engine")
    try:
        g.set_sequential_engine(engine, 0)
    except pmc.PMCError:
        pytest.skip("the state does not fit the LDS")
    pool = [_disp(pmc, 0.8)]
    for call in range(2):
        st = pmc.mc_sweep(g, pool, nsweeps=5, mode="sequential")
        att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.8}]), 31, 0, call * 5 * len(pos), 5 * len(pos))
        assert st.accepted[0] == acc[0], (st.accepted, acc)
        _same_state(g, o)


@pytest.mark.parametrize("waves", [1, 4])
def test_reference_benchmark_system_empty_list(gpu, waves, monkeypatch):
    """The system of the reference's ONLY published timing (benchmark/particles_benchmarks.jl:10-28: N = 216 3-D 50:50 BHHP soft
    spheres, rho = 0.5342, T = 0.2, sigma_move = 0.1, the default EmptyList -- every particle a neighbour, summed in id order):
    engine 3 keeps it in LDS too (no cells, no rows), alone and as many chains, against the oracle's EmptyList chain."""
    from oracle import oracle as O
    pmc = gpu
    monkeypatch.setenv("PMC_SEQ_WAVES", str(waves))
    N, rho, T = 216, 0.5342, 0.2
    pos, _ = random_gas(N, rho, 3, seed=42, min_dist=0.85)
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(42).permutation(N)[: N // 2]] = 2
    g = pmc.System(pos, sp, rho, T, pmc.BHHP(), list_type=pmc.EmptyList, seed=42)
    o = O.System(pos, sp, rho, T, O.bhhp(), O.EMPTYLIST)
    assert g.energy[0] == o.energy
    pool = [_disp(pmc, 0.1)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.1}])
    st = pmc.mc_sweep(g, pool, nsweeps=10, mode="sequential")           # engine: the library's choice (3)
    att, acc = o.mc_steps(opool, 42, 0, 0, 10 * N)
    assert st.accepted[0] == acc[0] and np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy
    g.set_sequential_engine(1, 0)                                        # ... and the literal loop continues the same chain
    st = pmc.mc_sweep(g, pool, nsweeps=1, mode="sequential")
    att, acc = o.mc_steps(opool, 42, 0, 10 * N, N)
    assert st.accepted[0] == acc[0] and np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy
    chains = [pmc.System(pos, sp, rho, T, pmc.BHHP(), list_type=pmc.EmptyList, seed=42, chain=k) for k in range(6)]
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=3, mode="sequential")
    for k in (0, 5):
        ok_ = O.System(pos, sp, rho, T, O.bhhp(), O.EMPTYLIST)
        att, acc = ok_.mc_steps(opool, 42, k, 0, 3 * N)
        assert sts[k].accepted[0] == acc[0] and np.array_equal(chains[k].position, ok_.positions()) and chains[k].energy[0] == ok_.energy


@pytest.mark.parametrize("engine", [2, 3])
def test_every_model_kind_on_the_fast_engines(gpu, engine):
    """The reference's only known answer (test/runtests.jl:36-38: config_0 under JBB, SmoothLennardJones, 2-D ternary) and soft
    spheres with an odd exponent (the Float64-power branch, src/models.jl:66) on the sequential chain of engines 2 and 3."""
    import os
    from oracle import oracle as O
    pmc = gpu
    gold = os.path.join(os.path.dirname(__file__), "golden", "config_0.lmp")
    s = pmc.load_chains(gold, args={"temperature": [0.231], "model": ["JBB"], "list_type": "LinkedList"})[0]
    assert abs(s.energy[0] / len(s) - (-2.676832)) < 1e-6
    cfg = pmc.load_configuration(gold)
    pos = pmc.fold_back(cfg["position"], cfg["box"])
    rho = cfg["N"] / np.prod(cfg["box"])
    g = pmc.System(pos, cfg["species"], rho, 0.231, pmc.JBB(), seed=10)
    o = O.System(pos, cfg["species"], rho, 0.231, O.jbb(), O.LINKEDLIST)
    assert g.energy[0] == o.energy
    g.set_sequential_engine(engine, 0)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 10, 0, 0, 3 * len(pos))
    assert st.accepted[0] == acc[0]
    _same_state(g, o)
    pos, sp, rho = binary_lattice(1000, frac_b=0.5, rho=0.8, jitter=0.05, seed=12)
    mm = [[pmc.SoftSpheres(1.0, sg, 9) for sg in row] for row in ([1.0, 1.1], [1.1, 1.2])]
    mo = O.model_matrix(lambda e, sg: O.soft_spheres(e, sg, 9), [[1.0, 1.0], [1.0, 1.0]], [[1.0, 1.1], [1.1, 1.2]])
    g = pmc.System(pos, sp, rho, 1.0, mm, seed=2)
    o = O.System(pos, sp, rho, 1.0, mo, O.LINKEDLIST)
    g.set_sequential_engine(engine, 0)
    pool = [_disp(pmc, 0.08, 0.8), _swap(pmc, (1, 2), (500, 500), 0.2)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.08, "probability": 0.8}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    st = pmc.mc_sweep(g, pool, nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(opool, 2, 0, 0, 3 * len(pos))
    assert np.array_equal(st.accepted, acc), (st.accepted, acc)
    _same_state(g, o, species=True)


def test_ss142d_hundred_chains_of_eight_particles_in_one_launch(gpu):
    """test/ss142d.jl:8-21 as written, all of it: M = 100 chains of N = 8 (4 + 4) 2-D BHHP soft spheres, rho = 0.5, T = 1,
    EmptyList, Displacement sigma = 0.065 -- one pmc_sweep_many(PMC_SEQUENTIAL) launch, every chain's state in LDS, every chain
    against the oracle's chain of its (seed, chain)."""
    from oracle import oracle as O
    pmc = gpu
    N, rho, T, M = 8, 0.5, 1.0, 100
    box = (N / rho) ** 0.5
    rng = np.random.default_rng(7)
    starts, species = [], []
    for k in range(M):
        starts.append(rng.uniform(0.0, box, (N, 2)))
        sp = np.array([1] * 4 + [2] * 4, dtype=np.int32)
        rng.shuffle(sp)
        species.append(sp)
    chains = [pmc.System(starts[k], species[k], rho, T, pmc.BHHP(), list_type=pmc.EmptyList, seed=42, chain=k) for k in range(M)]
    pool = [_disp(pmc, 0.065)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.065}])
    stats = pmc.mc_sweep_many(chains, pool, nsweeps=200, mode="sequential")
    for k in range(M):
        o = O.System(starts[k], species[k], rho, T, O.bhhp(), O.EMPTYLIST)
        att, acc = o.mc_steps(opool, 42, k, 0, 200 * N)
        assert stats[k].attempted[0] == att[0] == 1600 and stats[k].accepted[0] == acc[0], k
        assert np.array_equal(chains[k].position, o.positions()) and chains[k].energy[0] == o.energy, k


_FORMS = {"one launch per round": {}, "four launches": {"PMC_SEQ_NOFUSE": "1"},
          "one wavefront per step": {"PMC_SEQ_EW": "1"}, "cell rule": {"PMC_SEQ_CELLRULE": "1"},
          "cell rule, four launches, one wavefront (round 4's engine)": {"PMC_SEQ_CELLRULE": "1", "PMC_SEQ_NOFUSE": "1", "PMC_SEQ_EW": "1"},
          "launch by launch (the profile scripts' PMC_SEQ_NOGRAPH)": {"PMC_SEQ_NOGRAPH": "1"}}


@pytest.mark.parametrize("form", list(_FORMS))
def test_every_launch_form_and_conflict_rule_of_the_windows(gpu, form, monkeypatch):
    """Round 5 gave the speculative windows two launch forms (a round as one launch: evaluate, the last workgroup commits, admits
    and prepares; four launches), replayed from a captured graph or launch by launch, four wavefronts per evaluated step and a distance-based
    conflict rule next to the cell rule.  Each combination on one Kob-Andersen liquid with swaps and wide moves (cell crossings,
    accepted swaps, held-back steps) and on a 2-D soft-sphere system, three calls each: the oracle's chain, bit for bit."""
    from oracle import oracle as O
    pmc = gpu
    for var in ("PMC_SEQ_NOFUSE", "PMC_SEQ_EW", "PMC_SEQ_CELLRULE", "PMC_SEQ_NOGRAPH"):
        monkeypatch.delenv(var, raising=False)
    for var, val in _FORMS[form].items():
        monkeypatch.setenv(var, val)
    pos, sp, rho = binary_lattice(2744, frac_b=0.2, rho=1.2, jitter=0.05, seed=11)
    g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=21)
    o = O.System(pos, sp, rho, 1.0, O.kob_andersen(), O.LINKEDLIST)
    nb = int((sp == 2).sum())
    pm = [_disp(pmc, 0.05, 0.5), _disp(pmc, 0.4, 0.3), _swap(pmc, (1, 2), (len(sp) - nb, nb), 0.2)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.5}, {"kind": "Displacement", "sigma": 0.4, "probability": 0.3},
                      {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    done = 0
    for window, nsw in ((0, 2), (128, 1), (512, 1)):
        g.set_sequential_engine(2, window)
        st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        att, acc = o.mc_steps(po, 21, 0, done, nsw * len(pos))
        done += nsw * len(pos)
        assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc), (form, window, st.accepted, acc)
        _same_state(g, o, species=True)
    # 2-D, one species, soft spheres, particles stored boxes away
    from inputs import lj_lattice as lat
    pos, sp, rho = lat(40, jitter=0.05, seed=4, rho=0.8, d=2)
    pos = pos + np.random.default_rng(1).integers(-2, 3, size=pos.shape) * (len(pos) / rho) ** 0.5
    g = pmc.System(pos, sp, rho, 0.7, pmc.SoftSpheres(1.0, 1.0, 12), seed=8)
    o = O.System(pos, sp, rho, 0.7, O.single(O.soft_spheres(1.0, 1.0, 12)), O.LINKEDLIST)
    g.set_sequential_engine(2, 0)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.1)], nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.1}]), 8, 0, 0, 3 * len(pos))
    assert np.array_equal(st.accepted, acc), (form, st.accepted, acc)
    _same_state(g, o)


def test_distance_rule_commits_more_steps_per_round(gpu, monkeypatch):
    """What the distance-based conflict rule is for: on the N = 4096 LJ liquid a round of the cell rule commits ~8 steps (27 of 216
    cells conflict), a round of the distance rule ~20 -- the same chain either way (`sequential_info`: rounds of the last call)."""
    pmc = gpu
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
    states, rounds = [], []
    for rule in (None, "1"):
        monkeypatch.delenv("PMC_SEQ_CELLRULE", raising=False)
        if rule:
            monkeypatch.setenv("PMC_SEQ_CELLRULE", rule)
        g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
        pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=10, mode="checkerboard")
        g.set_sequential_engine(2, 0)
        pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=6, mode="sequential")
        states.append((g.position.copy(), g.energy[0]))
        rounds.append(g.sequential_info()[0])
    assert np.array_equal(states[0][0], states[1][0]) and states[0][1] == states[1][1]
    per_round = [6 * 4096 / r for r in rounds]
    assert per_round[0] > 14.0 and per_round[0] > 1.8 * per_round[1], per_round


def test_empty_list_beyond_the_lds_runs_with_positions_in_global_memory(gpu):
    """An EmptyList system whose positions alone outgrow the LDS (N = 4913: 118 KB) used to fall to the literal one-wavefront loop
    (146 us per step at N = 10 648, 2.6 x slower than one host core).  Round 5: engine 3 keeps everything but the positions in
    LDS and runs the all-pairs sums on four wavefronts, the trips in goes (k_sq_lds<.., PG>): the oracle's chain, bit for bit --
    a Kob-Andersen mixture with swaps, two calls, the second through pmc_sweep_many with a second chain."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(4913, frac_b=0.2, rho=1.2, jitter=0.05, seed=3)
    nA, nB = int((sp == 1).sum()), int((sp == 2).sum())
    pm = [_disp(pmc, 0.08, 0.8), _swap(pmc, (1, 2), (nA, nB), 0.2)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.08, "probability": 0.8}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    gs = [pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), list_type=pmc.EmptyList, seed=5, chain=k) for k in range(2)]
    os_ = [O.System(pos, sp, rho, 1.0, O.kob_andersen(), O.EMPTYLIST) for _ in range(2)]
    st = pmc.mc_sweep(gs[0], pm, nsweeps=1, mode="sequential")
    att, acc = os_[0].mc_steps(po, 5, 0, 0, 4913)
    assert np.array_equal(st.accepted, acc), (st.accepted, acc)
    sts = pmc.mc_sweep_many(gs, pm, nsweeps=1, mode="sequential")
    for k, (g, o) in enumerate(zip(gs, os_)):
        att, acc = o.mc_steps(po, 5, k, 4913 if k == 0 else 0, 4913)
        assert np.array_equal(sts[k].accepted, acc), (k, sts[k].accepted, acc)
        assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy, k
        assert np.array_equal(g.species, o.species())
        for s_ in (1, 2):
            assert np.array_equal(g.species_ids(s_), o.species_ids(s_)), (k, s_)


@pytest.mark.parametrize("d,n,model", [(2, 81, "soft"), (3, 18, "lj")])
def test_empty_list_beyond_the_lds_other_instances(gpu, d, n, model):
    """The 2-D instance of the positions-in-global-memory kernel (N = 6561 soft spheres, particles stored boxes away) and a 3-D
    single-species Lennard-Jones system (N = 5832), one sweep each against the oracle."""
    from oracle import oracle as O
    from inputs import lj_lattice as lat
    pmc = gpu
    pos, sp, rho = lat(n, jitter=0.05, seed=9, rho=0.9, d=d)
    N = len(pos)
    if d == 2:
        pos = pos + np.random.default_rng(2).integers(-1, 2, size=pos.shape) * (N / rho) ** 0.5
    mg, mo = ((pmc.SoftSpheres(1.0, 1.0, 12), O.soft_spheres(1.0, 1.0, 12)) if model == "soft"
              else (pmc.LennardJones(1.0, 1.0), O.lennard_jones(1.0, 1.0)))
    g = pmc.System(pos, sp, rho, 0.8, mg, list_type=pmc.EmptyList, seed=13)
    o = O.System(pos, sp, rho, 0.8, O.single(mo), O.EMPTYLIST)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.07)], nsweeps=1, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.07}]), 13, 0, 0, N)
    assert np.array_equal(st.accepted, acc), (st.accepted, acc)
    assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy


def test_empty_list_too_large_for_engine_3_is_refused_not_overrun(gpu):
    """an EmptyList system of about 18 900 .. 22 200 particles does not fit the four-wavefront layout of the
    positions-in-global-memory kernel, while the one-wavefront layout (23 KB smaller) would -- engine 3 used to launch the
    four-wavefront kernel with one wavefront and that layout (three of four trips never computed, the segments past the end of
    the LDS).  Now engine 3 refuses the system when it is asked for, and the default selection runs the literal loop: N = 19 683,
    one sweep, the oracle's chain bit for bit."""
    from oracle import oracle as O
    from inputs import lj_lattice as lat
    from particlesmc_amd import _lib
    pmc = gpu
    pos, sp, rho = lat(27, jitter=0.05, seed=4, rho=0.9, d=3)
    N = len(pos)
    assert N == 19683
    g = pmc.System(pos, sp, rho, 0.9, pmc.LennardJones(1.0, 1.0), list_type=pmc.EmptyList, seed=17)
    with pytest.raises(_lib.PMCError) as ei:
        g.set_sequential_engine(3, 0)
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED
    o = O.System(pos, sp, rho, 0.9, O.single(O.lennard_jones(1.0, 1.0)), O.EMPTYLIST)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.07)], nsweeps=1, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.07}]), 17, 0, 0, N)
    assert np.array_equal(st.accepted, acc), (st.accepted, acc)
    assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy
    # two such chains in one call: the shared-launch path takes the same decision
    gs = [pmc.System(pos, sp, rho, 0.9, pmc.LennardJones(1.0, 1.0), list_type=pmc.EmptyList, seed=17, chain=k) for k in range(2)]
    for h in gs:
        with pytest.raises(_lib.PMCError):
            h.set_sequential_engine(3, 0)
