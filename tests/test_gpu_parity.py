"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars: bit-exact for cell membership, linked lists, per-particle energies in reference
order, accept counts and positions; 1e-6 relative for total energies against the plain
FP64 reference sum (written where used)."""
import ctypes as C
import os

import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice, random_gas

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-6          # north_star: total energy within 1e-6 relative


def _pair(pmc, O, pos, sp, rho, T, model_p, model_o, seed=11):
    sysg = pmc.System(pos, sp, rho, T, model_p, seed=seed)
    syso = O.System(pos, sp, rho, T, model_o, O.LINKEDLIST)
    return sysg, syso


def _disp_pool(pmc, sigma=0.05):
    return [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, 1.0)]


def test_device_rng_bitwise(gpu):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(6)
    s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    rng = np.random.default_rng(3)
    n = 20000
    w1 = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    w2 = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    w1[:4] = [0, 0xffffffff, 1, 0x80000000]
    w2[:4] = [0, 0xffffffff, 0x20000000, 0xe0000000]
    z0, z1, lnu = np.zeros(n), np.zeros(n), np.zeros(n)
    up = C.POINTER(C.c_uint32)
    dp = C.POINTER(C.c_double)
    rc = pmc.lib().pmc_debug_rng(s._h, n, w1.ctypes.data_as(up), w2.ctypes.data_as(up), z0.ctypes.data_as(dp),
                                 z1.ctypes.data_as(dp), lnu.ctypes.data_as(dp))
    assert rc == 0
    L = O.lib()
    for k in range(n):
        a, b = O.box_muller(int(w1[k]), int(w2[k]))
        assert a == z0[k] and b == z1[k], k
        assert L.orc_det_log(L.orc_uniform53(int(w1[k]), int(w2[k]))) == lnu[k], k


def test_guard_free_division_is_ieee_in_range(gpu):
    """dev_div_inrange (the throughput kernels' 1/r2 and sigma2/r2) equals the IEEE quotient bit for bit
    over the whole range a cutoff allows, including very close pairs."""
    pmc = gpu
    pos, sp, rho = lj_lattice(6)
    s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    rng = np.random.default_rng(5)
    n = 400000
    den = np.concatenate([rng.uniform(0.3, 16.0, n // 2), 10.0 ** rng.uniform(-12, 2, n // 2)])
    num = np.concatenate([np.ones(n // 4), rng.uniform(0.5, 2.0, n // 4), np.ones(n // 4), rng.uniform(0.5, 2.0, n // 4)])
    qi, qf = np.zeros(n), np.zeros(n)
    dp = C.POINTER(C.c_double)
    rc = pmc.lib().pmc_debug_div(s._h, n, num.ctypes.data_as(dp), den.ctypes.data_as(dp), qi.ctypes.data_as(dp),
                                 qf.ctypes.data_as(dp))
    assert rc == 0
    assert np.array_equal(qi, num / den)          # device IEEE division == host IEEE division
    assert np.array_equal(qf, qi)


def test_golden_jbb_energy_on_device(gpu):
    """test/runtests.jl:36-38: energy/N = -2.676832 +- 1e-6 under JBB, LinkedList."""
    from oracle import oracle as O
    pmc = gpu
    chains = pmc.load_chains(os.path.join(GOLD, "config_0.lmp"), args={"temperature": [0.231], "model": ["JBB"],
                                                                       "list_type": "LinkedList"})
    s = chains[0]
    assert s.N == 1290 and s.d == 2
    assert abs(s.energy[0] / len(s) - (-2.676832)) < 1e-6
    cfg = pmc.load_configuration(os.path.join(GOLD, "config_0.lmp"))
    pos = pmc.fold_back(cfg["position"], cfg["box"])
    rho = cfg["N"] / np.prod(cfg["box"])
    ref = O.System(pos, cfg["species"], rho, 0.231, O.jbb(), O.LINKEDLIST)
    assert s.energy[0] == ref.energy                         # same pairwise sum of bit-identical e_i
    assert np.array_equal(s.get_cells(), ref.cells())
    assert np.array_equal(pmc.compute_energy_particles(s), ref.energy_particles())
    e, efix = s.total_energy()                               # exact fixed-point engine, 2-D
    assert abs(e - ref.energy) <= REL_TOL * abs(ref.energy)


@pytest.mark.parametrize("n,jit", [(12, 0.05), (16, 0.1)])
def test_cells_lists_energies_bitexact_3d(gpu, n, jit):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(n, jitter=jit, seed=n)
    pos[::7] += np.array([3.0, -2.0, 1.0]) * (n / rho ** (1 / 3))   # unwrapped images
    g, o = _pair(pmc, O, pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), O.single(O.lennard_jones(1.0, 1.0)))
    assert np.array_equal(g.box, o.box)
    cell, nc = o.cell_geometry()
    assert np.array_equal(g.neighbour_list.box, cell) and tuple(nc) == g.neighbour_list.ncells
    assert np.array_equal(g.get_cells(), o.cells())
    hg, lg = g.get_linked_list()
    ho, lo = o.linked_list()
    assert np.array_equal(hg, ho) and np.array_equal(lg, lo)
    for c in (1, 2, int(np.prod(nc)) // 2, int(np.prod(nc))):
        assert np.array_equal(g.get_neighbour_cells(c), o.neighbour_cells(c))
    assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles())
    assert g.energy[0] == o.energy
    for i in (1, 17, len(pos)):
        assert pmc.compute_energy_particle(g, i) == o.energy_particle(i)


def test_total_energy_exact_and_close(gpu):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(16, jitter=0.1, seed=5)
    g, o = _pair(pmc, O, pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), O.single(O.lennard_jones(1.0, 1.0)))
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=11)
    e, efix = g.total_energy()
    assert efix == cb.energy_fix
    assert abs(e - o.energy) <= REL_TOL * abs(o.energy)
    n_mc, w = g.grid()
    on, ow = cb.grid()
    assert np.array_equal(n_mc, on) and np.array_equal(w, ow)


@pytest.mark.parametrize("n,sweeps", [(12, 4), (20, 2)])
def test_checkerboard_sweeps_bitexact(gpu, n, sweeps):
    """Positions, accept counts and the running energy after K sweeps equal the CPU restatement
    of the same schedule bit for bit; the running energy equals a recomputation exactly."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(n, jitter=0.1, seed=n + 1)
    g, o = _pair(pmc, O, pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), O.single(O.lennard_jones(1.0, 1.0)),
                 seed=2024)
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=2024)
    pool_o = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    for rnd in range(2):
        st = pmc.mc_sweep(g, _disp_pool(pmc), nsweeps=sweeps)
        att, acc = cb.sweeps(pool_o, rnd * sweeps, sweeps)
        assert st.attempted[0] == att[0] == sweeps * len(pos)
        assert st.accepted[0] == acc[0]
        assert np.array_equal(g.position, cb.positions())
        assert st.energy_fix == cb.energy_fix
    run_fix = st.energy_fix
    e, efix = g.total_energy()
    assert efix == run_fix == cb.total_energy_fix()
    acc_rate = st.accepted[0] / st.attempted[0]
    assert 0.2 < acc_rate < 0.95


def test_sequential_sweep_bitexact(gpu):
    """PMC_SEQUENTIAL follows the reference chain (src/moves.jl:57-81): positions, linked lists,
    accept counts and system.energy equal the oracle's sequential driver bit for bit."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(8, jitter=0.1, seed=3, rho=0.9)
    g, o = _pair(pmc, O, pos, sp, rho, 0.8, pmc.LennardJones(1.0, 1.0), O.single(O.lennard_jones(1.0, 1.0)), seed=99)
    st = pmc.mc_sweep(g, _disp_pool(pmc, 0.15), nsweeps=3, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.15}]), 99, 0, 0, 3 * len(pos))
    assert st.attempted[0] == att[0] and st.accepted[0] == acc[0]
    assert np.array_equal(g.position, o.positions())
    assert g.energy[0] == o.energy
    assert np.array_equal(g.get_cells(), o.cells())
    hg, lg = g.get_linked_list()
    ho, lo = o.linked_list()
    assert np.array_equal(hg, ho) and np.array_equal(lg, lo)


def test_per_move_verbs_match_oracle(gpu):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(1000, frac_b=0.2, rho=1.2, jitter=0.05, seed=8)
    g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=1)
    o = O.System(pos, sp, rho, 1.0, O.kob_andersen(), O.LINKEDLIST)
    assert g.energy[0] == o.energy
    rng = np.random.default_rng(0)
    for k in range(40):
        if k % 2 == 0:
            i = int(rng.integers(1, 1001))
            delta = rng.normal(0, 0.4, 3)
            act = pmc.Displacement(i, delta, 0.0)
            e1, e2 = pmc.perform_action(g, act)
            r1, r2, rde = o.displacement_perform(i, delta)
            assert (e1, e2, act.de) == (r1, r2, rde)
            if k % 4 == 0:
                pmc.invert_action(act, g)
                pmc.revert_action(g, act)
                o.displacement_revert(i, -delta, rde)
        else:
            ia = int(rng.choice(g.species_ids(1)))
            ib = int(rng.choice(g.species_ids(2)))
            act = pmc.DiscreteSwap(ia, ib, (1, 2), (800, 200), 0.0)
            e1, e2 = pmc.perform_action(g, act)
            r1, r2, rde = o.swap_perform(ia, ib, 1, 2)
            assert (e1, e2, act.de) == (r1, r2, rde)
            if k % 3 == 0:
                pmc.invert_action(act, g)
                pmc.revert_action(g, act)
                o.swap_revert(ib, ia, 1, 2, rde)
        assert g.energy[0] == o.energy
        assert np.array_equal(g.species, o.species())
    assert np.array_equal(g.position, o.positions())
    assert np.array_equal(g.get_cells(), o.cells())
    for s_ in (1, 2):
        assert np.array_equal(g.species_ids(s_), o.species_ids(s_))


def test_kob_andersen_swaps_checkerboard_and_sequential(gpu):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(4096, frac_b=0.2, rho=1.2, jitter=0.05, seed=4)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 0.9),
            None]
    g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=31)
    pool[1] = pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.1)
    o = O.System(pos, sp, rho, 1.0, O.kob_andersen(), O.LINKEDLIST)
    pool_o = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.9},
                          {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}])
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=31)
    st = pmc.mc_sweep(g, pool, nsweeps=3)
    att, acc = cb.sweeps(pool_o, 0, 3)
    assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc)
    assert att.sum() == 3 * len(pos) and acc[1] > 0
    assert np.array_equal(g.position, cb.positions())
    assert np.array_equal(g.species, cb.species())
    assert st.energy_fix == cb.energy_fix == cb.total_energy_fix()
    assert np.count_nonzero(g.species == 2) == np.count_nonzero(sp == 2)
    # sequential chain with the global DoubleUniform pick
    g2 = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=32)
    st2 = pmc.mc_sweep(g2, pool, nsweeps=1, mode="sequential")
    att2, acc2 = o.mc_steps(pool_o, 32, 0, 0, len(pos))
    assert np.array_equal(st2.attempted, att2) and np.array_equal(st2.accepted, acc2)
    assert np.array_equal(g2.position, o.positions()) and np.array_equal(g2.species, o.species())
    assert g2.energy[0] == o.energy


def test_delta_e_batch(gpu):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.1, seed=9)
    g, o = _pair(pmc, O, pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), O.single(O.lennard_jones(1.0, 1.0)))
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=0)
    rng = np.random.default_rng(1)
    ids = rng.integers(1, len(pos) + 1, 64)
    deltas = rng.normal(0, 0.8, (64, 3))
    e1, e2 = g.delta_e(ids, deltas)
    for k in range(64):
        i = int(ids[k])
        assert e1[k] == cb.energy_at(i, pos[i - 1], 1)
        assert e2[k] == cb.energy_at(i, pos[i - 1] + deltas[k], 1)
        ref = o.energy_particle(i)
        assert abs(e1[k] - ref) <= 1e-9 * max(1.0, abs(ref))


def test_exact_per_particle_energies(gpu):
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(3375, frac_b=0.2, rho=1.2, jitter=0.05, seed=3)
    g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=1)
    o = O.System(pos, sp, rho, 1.0, O.kob_andersen(), O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=1)
    e = pmc.compute_energy_particles_exact(g)
    ref = o.energy_particles()
    assert np.allclose(e, ref, rtol=0, atol=5e-8)      # <= ~100 in-range terms, each rounded to 2^-32 (2.3e-10)
    for i in (1, 77, 3375):
        assert e[i - 1] == cb.energy_at(i, pos[i - 1], int(sp[i - 1]))
    assert int(round(e.sum() / 2 * 2.0 ** 32)) == cb.energy_fix


def test_soft_spheres_2d(gpu):
    """test/ss142d.jl shape (2-D 50:50 BHHP soft spheres) at a size the cell grid accepts; n = 12 and 14."""
    from oracle import oracle as O
    pmc = gpu
    N = 900
    pos, rho = random_gas(N, 0.5, 2, seed=2, min_dist=0.9)
    sp = np.ones(N, dtype=np.int32)
    sp[N // 2:] = 2
    for n in (12, 14):
        mm = [[pmc.SoftSpheres(1.0, s, n) for s in row] for row in ([1.0, 1.2], [1.2, 1.4])]
        g = pmc.System(pos, sp, 0.5, 1.0, mm, seed=5)
        o = O.System(pos, sp, 0.5, 1.0, O.bhhp(n), O.LINKEDLIST)
        assert np.array_equal(g.get_cells(), o.cells())
        assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles())
        cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=5)
        st = pmc.mc_sweep(g, _disp_pool(pmc, 0.065), nsweeps=3)
        att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.065}]), 0, 3)
        assert st.accepted[0] == acc[0] and np.array_equal(g.position, cb.positions())
        assert st.energy_fix == cb.energy_fix


def test_error_paths(gpu):
    pmc = gpu
    pos, sp, rho = lj_lattice(6)
    bad = pos.copy()
    bad[1] = bad[0]
    with pytest.raises(ValueError, match="infinite or NaN"):
        pmc.System(bad, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    # box 6 = 2.4 cutoffs: no two-coloured grid, but a single-cell box since round 5 (tests/test_gpu_single_cell.py)
    s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    assert tuple(int(v) for v in s.grid()[0]) == (1, 1, 1)
    assert pmc.mc_sweep(s, _disp_pool(pmc), nsweeps=1).attempted[0] == len(pos)
    pos, sp, rho = lj_lattice(4)                                         # box 4 = 1.6 cutoffs: refused; the reference's own chain runs
    s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    with pytest.raises(pmc.PMCError) as ei:
        pmc.mc_sweep(s, _disp_pool(pmc), nsweeps=1)
    assert ei.value.code == -4
    st = pmc.mc_sweep(s, _disp_pool(pmc), nsweeps=1, mode="sequential")
    assert st.attempted[0] == len(pos)
