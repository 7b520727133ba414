"""-m gpu: pmc_sweep_many -- all the chains of a `parallel = true` run (src/ParticlesMC.jl:164,246) or all the replicas
of a tempering ladder in shared launches.  Every chain must come out bit-identical to pmc_sweep on it alone, and to the
oracle's restatement of the schedule."""
import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice

pytestmark = pytest.mark.gpu


def _disp(pmc, sigma=0.05, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def _same(a, b, sa, sb):
    for r in range(len(a)):
        assert np.array_equal(a[r].position, b[r].position), r
        assert np.array_equal(a[r].species, b[r].species), r
        assert sa[r].energy_fix == sb[r].energy_fix and np.array_equal(sa[r].accepted, sb[r].accepted), r
        assert np.array_equal(sa[r].attempted, sb[r].attempted)
        assert (sa[r].pairs_in_range, sa[r].null_moves) == (sb[r].pairs_in_range, sb[r].null_moves)
        # a work counter, not a result: how many distance tests it takes depends on the LDS staging capacity the launch was
        # sized with (learnt per handle; one capacity for all the chains of a shared launch)
        assert abs(sa[r].pair_tests - sb[r].pair_tests) <= 0.02 * sb[r].pair_tests
        assert a[r].total_energy()[1] == sa[r].energy_fix


@pytest.mark.parametrize("team", ["auto", "2", "4"])
def test_batched_launches_equal_one_chain_at_a_time_and_the_oracle(gpu, monkeypatch, team):
    from oracle import oracle as O
    pmc = gpu
    if team != "auto":
        monkeypatch.setenv("PMC_TEAM", team)
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=2)
    pool = [_disp(pmc, 0.06)]
    temps = [0.8 + 0.1 * r for r in range(8)]
    a = [pmc.System(pos, sp, rho, T, pmc.LennardJones(1.0, 1.0), seed=7, chain=r) for r, T in enumerate(temps)]
    b = [pmc.System(pos, sp, rho, T, pmc.LennardJones(1.0, 1.0), seed=7, chain=r) for r, T in enumerate(temps)]
    sa = sb = None
    for _ in range(3):                                      # several calls: sweep counters, epochs and tickets carry over
        sa = pmc.mc_sweep_many(a, pool, nsweeps=4)
        sb = [pmc.mc_sweep(s, pool, nsweeps=4) for s in b]
        _same(a, b, sa, sb)
    assert len({int(s.accepted[0]) for s in sa}) > 1       # the chains differ from each other
    # ... and each chain is the oracle's chain
    for r in (0, 5):
        o = O.System(pos, sp, rho, temps[r], O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
        cb = O.Checkerboard(pos, sp, o.box, temps[r], O.single(O.lennard_jones(1.0, 1.0)), seed=7, chain=r)
        cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.06}]), 0, 12)
        assert np.array_equal(a[r].position, cb.positions()) and sa[r].energy_fix == cb.energy_fix


def test_batched_launches_with_swaps_and_in_two_dimensions(gpu):
    pmc = gpu
    pos, sp, rho = binary_lattice(8000, frac_b=0.2, rho=1.2, jitter=0.05, seed=3)

    def mk(r):
        return pmc.System(pos, sp, rho, 0.9 + 0.05 * r, pmc.KobAndersen(), seed=11, chain=r)

    def pool(s):
        return [_disp(pmc, 0.05, 0.85), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.15)]

    a, b = [mk(r) for r in range(4)], [mk(r) for r in range(4)]
    sa = pmc.mc_sweep_many(a, pool(a[0]), nsweeps=6)
    sb = [pmc.mc_sweep(s, pool(s), nsweeps=6) for s in b]
    _same(a, b, sa, sb)
    assert sum(int(s.accepted[1]) for s in sa) > 0
    # the reference-order side sees the state the batched sweeps left (lists rebuilt, SpeciesList of the current species)
    for x, y in zip(a, b):
        assert np.array_equal(x.get_cells(), y.get_cells()) and np.array_equal(x.species_ids(2), y.species_ids(2))
    n = 48
    N, rho2 = n * n, 0.8
    L = (N / rho2) ** 0.5
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    p2 = (g + 0.5) * (L / n) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
    s2 = np.ones(N, dtype=np.int32)
    s2[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, sg, 14) for sg in row] for row in ([1.0, 1.2], [1.2, 1.4])]

    def mk2(r):
        return pmc.System(p2, s2, rho2, 1.0 + 0.02 * r, mm, seed=5, chain=r)

    a, b = [mk2(r) for r in range(3)], [mk2(r) for r in range(3)]
    sa = pmc.mc_sweep_many(a, [_disp(pmc, 0.065)], nsweeps=5)
    sb = [pmc.mc_sweep(s, [_disp(pmc, 0.065)], nsweeps=5) for s in b]
    _same(a, b, sa, sb)


def test_chains_of_different_shapes_and_the_sequential_engine_fall_back_to_one_stream_per_chain(gpu):
    pmc = gpu
    pool = [_disp(pmc, 0.06)]
    shapes = [lj_lattice(12, jitter=0.05, seed=4), lj_lattice(14, jitter=0.05, seed=4)]
    a = [pmc.System(p, s, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=3, chain=r) for r, (p, s, rho) in enumerate(shapes)]
    b = [pmc.System(p, s, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=3, chain=r) for r, (p, s, rho) in enumerate(shapes)]
    _same(a, b, pmc.mc_sweep_many(a, pool, nsweeps=3), [pmc.mc_sweep(s, pool, nsweeps=3) for s in b])
    sa = pmc.mc_sweep_many(a, pool, nsweeps=1, mode="sequential")
    sb = [pmc.mc_sweep(s, pool, nsweeps=1, mode="sequential") for s in b]
    for r in range(2):
        assert np.array_equal(a[r].position, b[r].position) and a[r].energy[0] == b[r].energy[0]
        assert np.array_equal(sa[r].accepted, sb[r].accepted)
    # the same handle twice cannot be in flight twice: refused, and the handle is usable afterwards
    with pytest.raises(pmc.PMCError):
        pmc.mc_sweep_many([a[0], a[0]], pool, nsweeps=1)
    pmc.mc_sweep(a[0], pool, nsweeps=1)


@pytest.mark.parametrize("team", ["auto", "8"])
def test_batched_launches_mix_chains_with_and_without_periodic_images(gpu, monkeypatch, team):
    """Chains of one shape in shared launches where some carry periodic images (an unwrapped restart) and some sit in the
    primary box: the launch takes the image instances of the kernel, replicas without images hand it null image pointers.
    m = 2; every chain == itself alone == the oracle's chain."""
    from oracle import oracle as O
    pmc = gpu
    if team != "auto":
        monkeypatch.setenv("PMC_TEAM", team)
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=9)
    L = (len(pos) / rho) ** (1.0 / 3.0)
    rng = np.random.default_rng(4)
    inputs = [pos,                                                             # in the box
              pos + rng.integers(-3, 4, size=pos.shape) * L,                   # every particle in some image
              pos + (rng.random(pos.shape) < 0.1) * L,                         # a tenth of the coordinates one box up
              pos]

    def mk(r):
        s = pmc.System(inputs[r], sp, rho, 0.9 + 0.1 * r, pmc.LennardJones(1.0, 1.0), seed=21, chain=r)
        s.rounds = 2
        return s

    a, b = [mk(r) for r in range(4)], [mk(r) for r in range(4)]
    pool = [_disp(pmc, 0.07)]
    for _ in range(2):
        sa = pmc.mc_sweep_many(a, pool, nsweeps=5)
        sb = [pmc.mc_sweep(s, pool, nsweeps=5) for s in b]
        _same(a, b, sa, sb)
    for r in range(4):
        o = O.System(inputs[r], sp, rho, 0.9 + 0.1 * r, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
        cb = O.Checkerboard(inputs[r], sp, o.box, 0.9 + 0.1 * r, o.models, seed=21, chain=r)
        opool = O.make_pool([{"kind": "Displacement", "sigma": 0.07}])
        cb.sweeps(opool, 0, 5, rounds=2)
        cb.sweeps(opool, 5, 5, rounds=2)
        assert np.array_equal(a[r].position, cb.positions()) and sa[r].energy_fix == cb.energy_fix, r
        assert np.abs(a[r].position - inputs[r]).max() < 1.0                    # nobody changed image


@pytest.mark.parametrize("nchains,team", [(40, 4), (150, 2), (330, 1)])
def test_team_width_of_a_shared_launch_follows_all_its_chains(gpu, nchains, team):
    """A chain of N = 1728 has 8 cells per colour and runs teams of eight wavefronts on its own; 40 / 150 / 330 of them in one
    launch have 320 / 1200 / 2640 cells active at the same time and take teams of 4 / 2 / 1 for that launch (the rule of
    pmc_create applied to the launch).  Every chain still equals itself alone, a few of them the oracle, and the handles keep
    their own width afterwards."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=13)

    def mk(r):
        s = pmc.System(pos, sp, rho, 0.8 + 0.001 * r, pmc.LennardJones(1.0, 1.0), seed=5, chain=r)
        s.rounds = 2
        return s

    a = [mk(r) for r in range(nchains)]
    pool = [_disp(pmc, 0.06)]
    sa = pmc.mc_sweep_many(a, pool, nsweeps=3)
    sa = pmc.mc_sweep_many(a, pool, nsweeps=3)
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.06}])
    for r in sorted({0, 7, nchains // 2, nchains - 1}):
        b = mk(r)
        pmc.mc_sweep(b, pool, nsweeps=3)
        sb = pmc.mc_sweep(b, pool, nsweeps=3)                    # alone: its own width (8)
        assert np.array_equal(a[r].position, b.position) and sa[r].energy_fix == sb.energy_fix, r
        assert np.array_equal(sa[r].accepted, sb.accepted) and sa[r].null_moves == sb.null_moves
        o = O.System(pos, sp, rho, 0.8 + 0.001 * r, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
        cb = O.Checkerboard(pos, sp, o.box, 0.8 + 0.001 * r, o.models, seed=5, chain=r)
        cb.sweeps(opool, 0, 3, rounds=2)
        cb.sweeps(opool, 3, 3, rounds=2)
        assert np.array_equal(a[r].position, cb.positions()) and sa[r].energy_fix == cb.energy_fix, r
    assert all(s.total_energy()[1] == st.energy_fix for s, st in zip(a[:20], sa[:20]))
    assert len({int(st.accepted[0]) for st in sa}) > nchains // 4         # the chains differ


def test_shared_launch_with_more_than_8192_cells_per_chain(gpu):
    """Two chains of config 5's 2-D system at N = 65536 (102 x 102 = 10404 cells each: more than one pass of the shared
    launch's one-block scan used to allow): the shared launch == each chain alone, running energy == recomputation."""
    pmc = gpu
    n, rho = 256, 0.5
    N = n * n
    L = (N / rho) ** 0.5
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    pos = (g + 0.5) * (L / n) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, s, 14) for s in row] for row in ([1.0, 1.2], [1.2, 1.4])]

    def mk(r):
        s = pmc.System(pos, sp, rho, 1.0 + 0.01 * r, mm, seed=5, chain=r)
        s.rounds = 2
        return s

    a, b = [mk(r) for r in range(2)], [mk(r) for r in range(2)]
    assert int(np.prod(a[0].grid()[0])) > 8192
    pool = [_disp(pmc, 0.065)]
    sa = pmc.mc_sweep_many(a, pool, nsweeps=5)
    sb = [pmc.mc_sweep(s, pool, nsweeps=5) for s in b]
    _same(a, b, sa, sb)
    assert sa[0].sweep_kernel_launches == 3 and sa[0].accepted[0] != sa[1].accepted[0]
