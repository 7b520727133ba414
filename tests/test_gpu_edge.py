"""-m gpu: the paths a liquid at rho = 1 never takes -- dense cells (more than one wavefront of
particles, neighbourhoods beyond the LDS capacity), a 4-cell periodic box where every cell wraps,
2-D multi-species swaps, mixed model kinds, long linked-list chains, local parallel tempering."""
import os

import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _disp(pmc, sigma=0.05, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def test_dense_cells_chunks_and_lds_overflow(gpu):
    """rcut = 4 at rho = 1: ~77 particles per cell (two slot chunks per cell), ~2070 candidates per
    neighbourhood (beyond the 1024-candidate LDS capacity -> overflow loops), 4 cells per side
    (every neighbourhood wraps)."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(17, jitter=0.08, seed=3)
    model_p = pmc.LennardJones(1.0, 1.0, rcut=4.0, shift_potential=False)
    model_o = O.single(O.lennard_jones(1.0, 1.0, 4.0, False))
    g = pmc.System(pos, sp, rho, 1.3, model_p, seed=77)
    o = O.System(pos, sp, rho, 1.3, model_o, O.LINKEDLIST)
    assert tuple(g.grid()[0]) == (4, 4, 4)
    assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles()) and g.energy[0] == o.energy
    cb = O.Checkerboard(pos, sp, o.box, 1.3, o.models, seed=77)
    e, efix = g.total_energy()
    assert efix == cb.energy_fix and abs(e - o.energy) <= 1e-6 * abs(o.energy)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.07)], nsweeps=2)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.07}]), 0, 2)
    assert st.attempted[0] == att[0] == 2 * len(pos) and st.accepted[0] == acc[0]
    assert np.array_equal(g.position, cb.positions()) and st.energy_fix == cb.energy_fix == cb.total_energy_fix()


def test_2d_ternary_jbb_swaps_on_the_reference_fixture(gpu):
    """test/runtests.jl:93-129's pool (20 % Displacement, 40 % swap(1,3), 40 % swap(2,3)) on
    config_0 under JBB in the checkerboard schedule: 2-D, three species, SmoothLennardJones."""
    from oracle import oracle as O
    pmc = gpu
    g = pmc.load_chains(os.path.join(GOLD, "config_0.xyz"), args={"temperature": [0.231], "model": ["JBB"]}, seed=10)[0]
    pool = [_disp(pmc, 0.05, 0.2),
            pmc.Move(pmc.DiscreteSwap([1, 3], g), pmc.DoubleUniform(), [], 0.4),
            pmc.Move(pmc.DiscreteSwap([2, 3], g), pmc.DoubleUniform(), [], 0.4)]
    assert pool[1].action.particles_per_species == (600, 360) and pool[2].action.particles_per_species == (330, 360)
    pos0, sp0 = g.position, g.species
    cb = O.Checkerboard(pos0, sp0, g.box, 0.231, O.jbb(), seed=10)
    pool_o = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.2},
                          {"kind": "DiscreteSwap", "species": (1, 3), "probability": 0.4},
                          {"kind": "DiscreteSwap", "species": (2, 3), "probability": 0.4}])
    st = pmc.mc_sweep(g, pool, nsweeps=6)
    att, acc = cb.sweeps(pool_o, 0, 6)
    assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc) and acc[1] > 0
    assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())
    assert st.energy_fix == cb.energy_fix == cb.total_energy_fix()
    assert sorted(np.bincount(g.species)[1:]) == [330, 360, 600]
    assert abs(st.energy / 1290 + 2.67) < 0.05


def test_mixed_model_kinds(gpu):
    """A model matrix mixing LennardJones and SoftSpheres entries takes the dynamic-dispatch kernels."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(3375, frac_b=0.5, rho=0.9, jitter=0.05, seed=6)
    mm_p = [[pmc.LennardJones(1.0, 1.0), pmc.SoftSpheres(1.0, 1.1, 12)],
            [pmc.SoftSpheres(1.0, 1.1, 12), pmc.LennardJones(0.8, 0.9)]]
    arr = (O.Model * 4)()
    arr[0], arr[1] = O.lennard_jones(1.0, 1.0), O.soft_spheres(1.0, 1.1, 12)
    arr[2], arr[3] = O.soft_spheres(1.0, 1.1, 12), O.lennard_jones(0.8, 0.9)
    g = pmc.System(pos, sp, rho, 1.0, mm_p, seed=4)
    o = O.System(pos, sp, rho, 1.0, (arr, 2), O.LINKEDLIST)
    assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles())
    cb = O.Checkerboard(pos, sp, o.box, 1.0, (arr, 2), seed=4)
    pool = [_disp(pmc, 0.05, 0.8), pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.2)]
    st = pmc.mc_sweep(g, pool, nsweeps=3)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.8},
                                      {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}]), 0, 3)
    assert np.array_equal(st.accepted, acc) and np.array_equal(g.position, cb.positions())
    assert np.array_equal(g.species, cb.species()) and st.energy_fix == cb.energy_fix


def test_sequential_engine_long_chains_mixture(gpu):
    """examples/lj-mixture geometry (N = 1000, rcut = 4: 2 cells per side, 125-particle chains, every
    cell a neighbour of every cell): the reference-order engine stays bit-exact with the oracle, and
    its acceptance rates sit where the reference's published table puts them."""
    from oracle import oracle as O
    from mixture import EPS, SIG, RCUT, N, make_config, reference_rows
    pmc = gpu
    row = reference_rows()[0]
    pos, sp, rho = make_config(row["x"], row["density"], seed=1)
    mm_p = [[pmc.LennardJones(EPS[i][j], SIG[i][j], rcut=RCUT, shift_potential=False) for j in range(2)] for i in range(2)]
    mm_o = O.model_matrix(lambda e, s: O.lennard_jones(e, s, RCUT, False), EPS, SIG)
    g = pmc.System(pos, sp, rho, row["t"], mm_p, seed=42)
    o = O.System(pos, sp, rho, row["t"], mm_o, O.LINKEDLIST)
    assert g.neighbour_list.ncells == (2, 2, 2) and g.energy[0] == o.energy
    pool = [_disp(pmc, 0.05, 0.9), pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.1)]
    pool_o = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.9},
                          {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}])
    st = pmc.mc_sweep(g, pool, nsweeps=20, mode="sequential")
    att, acc = o.mc_steps(pool_o, 42, 0, 0, 20 * N)
    assert np.array_equal(st.attempted, att) and np.array_equal(st.accepted, acc)
    assert np.array_equal(g.position, o.positions()) and np.array_equal(g.species, o.species())
    assert g.energy[0] == o.energy
    # 20 sweeps from a lattice are not equilibrated; the 400-sweep statistical check is the CPU test
    assert abs(acc[0] / att[0] - row["acceptance_rate_displacement"]) < 0.12
    assert abs(acc[1] / att[1] - row["acceptance_rate_swap"]) < 0.15


def test_python_level_mc_step_equals_device_sequential_sweep(gpu):
    """Arianna's mc_step! contract driven from the host over the per-move verbs reproduces
    pmc_sweep(PMC_SEQUENTIAL) move for move (same Philox stream)."""
    pmc = gpu
    pos, sp, rho = binary_lattice(512, frac_b=0.25, rho=1.1, jitter=0.05, seed=2)
    a = pmc.System(pos, sp, rho, 0.9, pmc.KobAndersen(), seed=5)
    b = pmc.System(pos, sp, rho, 0.9, pmc.KobAndersen(), seed=5)
    pool = [_disp(pmc, 0.1, 0.7), pmc.Move(pmc.DiscreteSwap([1, 2], a), pmc.DoubleUniform(), [], 0.3)]
    nsteps = 256
    rng = pmc.SequentialStream(seed=5, chain=0, step=0)
    accepted = [0, 0]
    for _ in range(nsteps):
        m = rng.select_move(pool)
        mv = pool[m]
        accepted[m] += bool(pmc.mc_step(a, mv.action, mv.policy, mv.parameters, rng))
        rng.advance()
    import ctypes as C
    from particlesmc_amd import _lib
    from particlesmc_amd.simulation import c_pool
    # nsteps is not a whole sweep: drive the C ABI with a 512-particle sweep on b and compare the first half
    st = _lib.Stats()
    _lib.check(_lib.lib().pmc_sweep(b._h, _lib.SEQUENTIAL, c_pool(pool), 2, 1, C.byref(st)), b._h)
    c = pmc.System(pos, sp, rho, 0.9, pmc.KobAndersen(), seed=5)
    rng2 = pmc.SequentialStream(seed=5, chain=0, step=0)
    for _ in range(512):
        m = rng2.select_move(pool)
        pmc.mc_step(c, pool[m].action, pool[m].policy, pool[m].parameters, rng2)
        rng2.advance()
    assert np.array_equal(c.position, b.position) and np.array_equal(c.species, b.species)
    assert c.energy[0] == b.energy[0]
    assert sum(accepted) > 0


def test_local_parallel_tempering(gpu):
    """Four replicas of one system on one GPU, geometric ladder: labels stay a permutation, exchanges
    happen, and the mean energy increases with temperature."""
    from particlesmc_amd.tempering import run_tempering_local
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=1)
    ladder = [1.0, 1.15, 1.3225, 1.52]
    systems = [pmc.System(pos, sp, rho, T, pmc.LennardJones(1.0, 1.0), seed=100, chain=r) for r, T in enumerate(ladder)]
    for s in systems:
        s.total_energy()
    label, acc, att, traces = run_tempering_local(systems, [_disp(pmc, 0.08)], ladder, nsweeps=120, exchange_every=4, seed=9)
    assert sorted(label) == [0, 1, 2, 3] and att > 0 and acc > 0
    by_T = {k: [] for k in range(4)}
    for tr in traces:
        for lab, e in tr[10:]:
            by_T[lab].append(e)
    means = [np.mean(by_T[k]) for k in range(4)]
    assert all(means[k] < means[k + 1] for k in range(3))
    for r, s in enumerate(systems):
        assert s.temperature == ladder[label[r]]
        _, fix = s.total_energy()


def test_concurrent_chains_begin_end(gpu):
    """pmc_sweep_begin / pmc_sweep_end: four chains in flight together give exactly the results of
    four chains run one after the other; a second begin on a busy handle is refused."""
    import ctypes as C
    from particlesmc_amd import _lib
    from particlesmc_amd.simulation import c_pool
    pmc = gpu
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=2)
    pool = [_disp(pmc, 0.06)]
    a = [pmc.System(pos, sp, rho, 1.0 + 0.1 * r, pmc.LennardJones(1.0, 1.0), seed=7, chain=r) for r in range(4)]
    b = [pmc.System(pos, sp, rho, 1.0 + 0.1 * r, pmc.LennardJones(1.0, 1.0), seed=7, chain=r) for r in range(4)]
    sa = pmc.mc_sweep_many(a, pool, nsweeps=5)
    sb = [pmc.mc_sweep(s, pool, nsweeps=5) for s in b]
    for r in range(4):
        assert np.array_equal(a[r].position, b[r].position)
        assert sa[r].energy_fix == sb[r].energy_fix and sa[r].accepted[0] == sb[r].accepted[0]
    assert len({s.accepted[0] for s in sa}) > 1                      # chains differ from each other
    L = _lib.lib()
    arr = c_pool(pool)
    assert L.pmc_sweep_begin(a[0]._h, _lib.CHECKERBOARD, arr, 1, 1) == 0
    assert L.pmc_sweep_begin(a[0]._h, _lib.CHECKERBOARD, arr, 1, 1) == _lib.PMC_ERR_STATE
    st = _lib.Stats()
    assert L.pmc_sweep(a[0]._h, _lib.CHECKERBOARD, arr, 1, 1, C.byref(st)) == _lib.PMC_ERR_STATE
    assert L.pmc_sweep_end(a[0]._h, C.byref(st)) == 0 and st.attempted[0] == len(pos)
    assert L.pmc_sweep_end(a[0]._h, C.byref(st)) == _lib.PMC_ERR_STATE


# every row of the reference's published table (tests/golden/lj_mixture_calculated_energies.csv: 23 state points, three
# compositions, T = 1.2183 ... 1.6438, rho = 0.1 ... 0.8)
MIXTURE_ROWS = list(range(23))


# ... every row with the library's default schedule (m = 1), every third row also with bench.py's headline schedule (m = 4): more
# sweeps per grid shift must not move the equilibrium.  (All 23 rows pass at m = 4 too -- 6 minutes; the rows at rho = 0.8 have
# a 4 x 4 x 4 grid of 125-particle cells at this model's rcut = 4 and run at 10^6 moves/s.)
MIXTURE_CASES = [(r, 1) for r in MIXTURE_ROWS] + [(r, 4) for r in MIXTURE_ROWS[1::3]]


@pytest.mark.parametrize("row_index,rounds", MIXTURE_CASES)
def test_checkerboard_reproduces_published_mixture_thermodynamics(gpu, row_index, rounds):
    """    The parallel schedule (cell-local DoubleUniform swaps included) against the reference's own
    published results for examples/lj-mixture (run-validation.py:98-132,155-180; calculated-energies.csv):
    same state points, same models and move pool, the reference's protocol (1000 sweeps from the lattice,
    second half averaged), N = 8000 so that the checkerboard grid exists (the energy per particle of this
    short-range-cut model has no visible size dependence at this accuracy).  Mean energy within errors,
    swap acceptance within a few percent, Displacement acceptance lower only by the cell-leaving rejections,
    which the kernel counts (null_moves)."""
    from mixture import EPS, SIG, RCUT, reference_rows
    pmc = gpu
    row = reference_rows()[row_index]
    n = 20
    N = n ** 3
    L = (N / row["density"]) ** (1 / 3)
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    pos = (g + 0.5) * (L / n)
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(3 + row_index).permutation(N)[: int(round(N * row["x"]))]] = 2
    mm = [[pmc.LennardJones(EPS[i][j], SIG[i][j], rcut=RCUT, shift_potential=False) for j in range(2)] for i in range(2)]
    s = pmc.System(pos, sp, N / L ** 3, row["t"], mm, seed=42 + row_index)
    assert min(s.grid()[0]) >= 4
    pool = [_disp(pmc, 0.05, 0.9), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1)]
    s.total_energy()
    s.rounds = rounds
    # equilibration (the reference discards the first half of its 1000 sweeps; at N = 8000 the state points near coexistence --
    # row 7: T = 1.51, rho = 0.4 -- are still coarsening then, whatever the schedule: e = -3.02 over sweeps 500-1000 and -3.046
    # +- 0.004 over sweeps 500-3700 for m = 1, 2, 4 and 8 alike, reference -3.063 +- 0.010; hence 1500 below the liquid densities)
    pmc.mc_sweep(s, pool, nsweeps=1500 if row["density"] <= 0.5 else 500)
    en = []
    att = np.zeros(2)
    acc = np.zeros(2)
    null = 0
    per_call, ncalls = (5, 100) if rounds == 1 else (rounds, 10 * (50 // rounds))     # whole passes only: 500 resp. 480 sweeps
    for _ in range(ncalls):
        st = pmc.mc_sweep(s, pool, nsweeps=per_call)
        att += st.attempted
        acc += st.accepted
        null += st.null_moves
        en.append(st.energy / N)
    en = np.array(en)
    err = en.reshape(10, -1).mean(1).std(ddof=1) / np.sqrt(10)
    assert abs(en.mean() - row["energy"]) < 4 * np.hypot(err, row["energy_err"]) + 0.005
    # swap acceptance: a slot whose cell lacks one of the two species is a null move of the cell-local DoubleUniform (counted
    # as attempted); their share -- 13 % of the swap slots at rho = 0.1, x = 0.25, none in the liquids -- is measured on
    # the final configuration (cells weighted by their slots) and taken out of the reference's rate
    n_mc, w = s.grid()
    cell = np.floor(s.position / w).astype(np.int64) % n_mc
    lin = (cell[:, 0] * n_mc[1] + cell[:, 1]) * n_mc[2] + cell[:, 2]
    ncell = int(np.prod(n_mc))
    spn = s.species
    ca, cb_ = np.bincount(lin[spn == 1], minlength=ncell), np.bincount(lin[spn == 2], minlength=ncell)
    f_null_swap = float(((ca + cb_) * ((ca == 0) | (cb_ == 0))).sum()) / N
    assert f_null_swap < (0.2 if row["density"] < 0.25 else 0.02)
    assert abs(acc[1] / att[1] - row["acceptance_rate_swap"] * (1.0 - f_null_swap)) < 0.03
    # Displacement acceptance: the reference's, minus the slots that evaluated nothing (Displacements leaving their
    # cell; a cell that lacks a species makes a swap slot null as well, so the bound is one-sided by that little)
    a_d = acc[0] / att[0]
    null_frac = null / att.sum()
    assert null_frac < 0.08
    assert row["acceptance_rate_displacement"] - null_frac / 0.9 - 0.02 < a_d < row["acceptance_rate_displacement"] + 0.01
    _, fix = s.total_energy()
    assert fix == st.energy_fix


def test_dilute_gas_empty_cells_and_tiny_systems(gpu):
    """Ragged inputs: a dilute gas (most cells empty, many with one particle), and systems of one
    and two particles (the oracle runs the same schedule)."""
    from oracle import oracle as O
    pmc = gpu
    rng = np.random.default_rng(11)
    N, rho = 400, 0.01
    L = (N / rho) ** (1 / 3)
    pos = rng.uniform(0, L, (N, 3))
    sp = np.ones(N, dtype=np.int32)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=3)
    o = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    assert np.array_equal(g.get_cells(), o.cells()) and g.energy[0] == o.energy
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=3)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.5)], nsweeps=5)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.5}]), 0, 5)
    assert st.attempted[0] == att[0] == 5 * N and st.accepted[0] == acc[0]
    assert np.array_equal(g.position, cb.positions()) and st.energy_fix == cb.energy_fix
    for n in (1, 2):
        p2 = np.array([[1.0, 2.0, 3.0], [1.9, 2.2, 3.1]])[:n]
        s2 = np.ones(n, dtype=np.int32)
        rho2 = n / 12.0 ** 3
        g2 = pmc.System(p2, s2, rho2, 1.0, pmc.LennardJones(1.0, 1.0), seed=1)
        o2 = O.System(p2, s2, rho2, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
        assert g2.energy[0] == o2.energy
        cb2 = O.Checkerboard(p2, s2, o2.box, 1.0, o2.models, seed=1)
        st2 = pmc.mc_sweep(g2, [_disp(pmc, 0.1)], nsweeps=20)
        a2, c2 = cb2.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.1}]), 0, 20)
        assert st2.accepted[0] == c2[0] and np.array_equal(g2.position, cb2.positions())
        st3 = pmc.mc_sweep(g2, [_disp(pmc, 0.1)], nsweeps=5, mode="sequential")
        assert st3.attempted[0] == 5 * n


@pytest.mark.parametrize("sigma", [0.02, 1.3, 5.5, 30.0])
def test_proposals_that_leave_the_cell_or_the_box(gpu, sigma):
    """'Stays in its cell' (floor((x - shift) / w) of the old and the new position) is decided from an FP32 offset to the
    cell's centre, with the exact divisions only near a boundary.  Displacements of every size -- inside the cell, across
    cells, half a box (where folding the new position into the centre's image would alias it back onto the cell), several
    boxes -- on particles scattered over several periodic images: same decisions as the oracle, bit for bit."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=5)
    L = (len(pos) / rho) ** (1.0 / 3.0)
    pos = pos + np.random.default_rng(8).integers(-2, 3, size=pos.shape) * L
    g = pmc.System(pos, sp, rho, 1.5, pmc.LennardJones(1.0, 1.0), seed=21)
    o = O.System(pos, sp, rho, 1.5, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, o.box, 1.5, o.models, seed=21)
    for team in ("1", "4"):
        os.environ["PMC_TEAM"] = team
        try:
            g2 = pmc.System(g.position, sp, rho, 1.5, pmc.LennardJones(1.0, 1.0), seed=21)
        finally:
            del os.environ["PMC_TEAM"]
        st = pmc.mc_sweep(g2, [_disp(pmc, sigma)], nsweeps=6)
        if team == "1":
            att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": sigma}]), 0, 6)
        assert st.attempted[0] == att[0] and st.accepted[0] == acc[0]
        assert np.array_equal(g2.position, cb.positions()) and st.energy_fix == cb.energy_fix


def test_fused_staging_on_small_multi_species_and_two_dimensional_systems(gpu, monkeypatch):
    """One wavefront per cell applies the box filter while staging and learns its LDS capacity from a histogram; systems
    this small normally run in teams, which stage the old way.  Forced to one wavefront per cell -- with the capacity pinned
    below the survivor count of most visits, and left to be learnt -- the chains must be the ones the teams produce."""
    pmc = gpu

    def ka():
        pos, sp, rho = binary_lattice(8000, frac_b=0.2, rho=1.2, jitter=0.05, seed=3)
        mk = lambda: pmc.System(pos, sp, rho, 0.9, pmc.KobAndersen(), seed=11)
        pool = lambda s: [_disp(pmc, 0.05, 0.85), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.15)]
        return mk, pool

    def ss2d():
        n = 48
        N, rho2 = n * n, 0.8
        L = (N / rho2) ** 0.5
        g = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
        p2 = (g + 0.5) * (L / n) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
        s2 = np.ones(N, dtype=np.int32)
        s2[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
        mm = [[pmc.SoftSpheres(1.0, sg, 14) for sg in row] for row in ([1.0, 1.2], [1.2, 1.4])]
        mk = lambda: pmc.System(p2, s2, rho2, 1.0, mm, seed=5)
        pool = lambda s: [_disp(pmc, 0.065)]
        return mk, pool

    def lj_images():
        pos, sp, rho = lj_lattice(16, jitter=0.05, seed=2)
        L = (len(pos) / rho) ** (1.0 / 3.0)
        pos = pos + np.random.default_rng(4).integers(-1, 2, size=pos.shape) * L
        mk = lambda: pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=9)
        pool = lambda s: [_disp(pmc, 0.06)]
        return mk, pool

    for make in (ka, ss2d, lj_images):
        mk, pool = make()
        monkeypatch.setenv("PMC_TEAM", "4")
        ref = mk()
        sr = [pmc.mc_sweep(ref, pool(ref), nsweeps=3) for _ in range(2)][-1]
        for cap in (None, "128"):
            monkeypatch.setenv("PMC_TEAM", "1")
            if cap:
                monkeypatch.setenv("PMC_CAP", cap)
            else:
                monkeypatch.delenv("PMC_CAP", raising=False)
            s = mk()
            st = [pmc.mc_sweep(s, pool(s), nsweeps=3) for _ in range(2)][-1]     # the second call runs with the learnt capacity
            assert np.array_equal(s.position, ref.position) and np.array_equal(s.species, ref.species), (make.__name__, cap)
            assert st.energy_fix == sr.energy_fix and np.array_equal(st.accepted, sr.accepted), (make.__name__, cap)
            assert s.total_energy()[1] == st.energy_fix
        monkeypatch.delenv("PMC_CAP", raising=False)


def test_many_accepted_swaps_give_one_chain_whatever_the_team_width(gpu, monkeypatch):
    """A 50:50 mixture at T = 2 with 40 % DiscreteSwap slots accepts thousands of swaps per sweep.  In a team, a swap slot that
    follows accepted Displacements of other particles in its batch is corrected by eight exact pair terms instead of being
    evaluated again; an accepted swap ends its batch.  Whatever the team width -- including one wavefront per cell, which has no
    batches -- the chain (positions, species, exact energy, both accept counts) must be the same, the running energy must
    equal a recomputation, and the oracle must agree."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(8000, frac_b=0.5, rho=1.0, jitter=0.05, seed=7)

    def pool(s):
        return [_disp(pmc, 0.05, 0.6), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.4)]

    ref = None
    for team in ("8", "4", "2", "1"):
        monkeypatch.setenv("PMC_TEAM", team)
        s = pmc.System(pos, sp, rho, 2.0, pmc.KobAndersen(), seed=3)
        st = None
        for _ in range(2):
            st = pmc.mc_sweep(s, pool(s), nsweeps=10)
            assert s.total_energy()[1] == st.energy_fix
        sig = (s.position.tobytes(), s.species.tobytes(), int(st.energy_fix), tuple(int(v) for v in st.accepted[:2]))
        if ref is None:
            ref = sig
            assert st.accepted[1] > 2000
        assert sig == ref, team
    o = O.System(pos, sp, rho, 2.0, O.kob_andersen(), O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, o.box, 2.0, o.models, seed=3)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.6},
                                      {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.4}]), 0, 20)
    assert np.array_equal(np.frombuffer(ref[0], dtype=np.float64).reshape(-1, 3), cb.positions()) and ref[2] == cb.energy_fix


def test_eight_million_particles(gpu):
    """Maximum-size case: N = 8 * 10^6 (720 MB of device state).  Cell membership against numpy,
    exact energy bookkeeping across sweeps, reproducibility."""
    pmc = gpu
    pos, sp, rho = lj_lattice(200, jitter=0.05, seed=9)
    N = len(pos)
    a = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=1)
    assert a.neighbour_list.ncells == (79, 79, 79)
    mc = np.floor_divide(pos, a.neighbour_list.box).astype(np.int64) % 79
    assert np.array_equal(a.get_cells(), 1 + (mc[:, 0] * 79 + mc[:, 1]) * 79 + mc[:, 2])
    e_ref = a.energy[0]
    e_fix, _ = a.total_energy()
    assert abs(e_fix - e_ref) <= 1e-9 * abs(e_ref)
    st = pmc.mc_sweep(a, [_disp(pmc)], nsweeps=3)
    assert st.attempted[0] == 3 * N and 0.2 < st.accepted[0] / st.attempted[0] < 0.5
    assert a.total_energy()[1] == st.energy_fix


def test_energy_bias_policy(gpu):
    """EnergyBias (src/moves.jl:246-280) on the device: the energy-weighted picks and the proposal density are computed
    from the device-resident reference-order energies (pmc_energy_bias_sample / _logq) and equal the oracle's
    restatement bit for bit -- weights (fixed-sequence exp), Base's pairwise sum, the inverse-CDF walk, the log;
    mc_step with the policy keeps the chain consistent (energy bookkeeping against a recomputation)."""
    from oracle import oracle as O
    from particlesmc_amd.moves import energy_bias_logq, energy_bias_sample
    pmc = gpu
    pos, sp, rho = binary_lattice(729, frac_b=0.3, rho=1.1, jitter=0.08, seed=5)
    g = pmc.System(pos, sp, rho, 0.7, pmc.KobAndersen(), seed=9)
    o = O.System(pos, sp, rho, 0.7, O.kob_andersen(), O.LINKEDLIST)
    par = {"theta1": 0.3, "theta2": -0.2}
    rng = np.random.default_rng(2)
    for species in ((1, 2), (2, 1)):
        for _ in range(40):
            u1, u2 = float(rng.random()), float(rng.random())
            got = energy_bias_sample(g, species, par, u1, u2)
            assert got == o.energy_bias(species, (0.3, -0.2), (u1, u2))
            # the density of that pair, and of an arbitrary other pair, as log_proposal_density sees them
            assert energy_bias_logq(g, species, par, got[0], got[1]) == got[2]
    ids1, ids2 = o.species_ids(1), o.species_ids(2)
    for i, j in ((int(ids1[3]), int(ids2[5])), (int(ids1[-1]), int(ids2[0]))):
        assert energy_bias_logq(g, (1, 2), par, i, j) == o.energy_bias((1, 2), (0.3, -0.2), given=(i, j))[2]
    # the extremes of the inverse-CDF walk: u = 0 picks the first id of each list, u -> 1 the last
    assert energy_bias_sample(g, (1, 2), par, 0.0, 0.0)[:2] == (int(ids1[0]), int(ids2[0]))
    assert energy_bias_sample(g, (1, 2), par, 1.0 - 2.0 ** -53, 1.0 - 2.0 ** -53)[:2] == (int(ids1[-1]), int(ids2[-1]))
    with pytest.raises(pmc.PMCError):
        energy_bias_sample(g, (1, 3), par, 0.1, 0.1)                              # no such species
    # the policy inside the Arianna step contract, against the oracle driven with the same words
    act = pmc.DiscreteSwap([1, 2], g)
    pol = pmc.EnergyBias()
    stream = pmc.SequentialStream(seed=9, chain=0, step=0)
    acc = 0
    for _ in range(30):
        u1, u2 = stream.R[1] / 2.0 ** 32, stream.Q[0] / 2.0 ** 32
        oi, oj, olq = o.energy_bias((1, 2), (0.3, -0.2), (u1, u2))
        ok = bool(pmc.mc_step(g, act, pol, par, stream))
        # replay on the oracle: forward density, swap, backward density of the inverted action, Metropolis-Hastings
        assert (act.i, act.j) in ((oi, oj), (oj, oi))
        e1, e2, de = o.swap_perform(oi, oj, 1, 2)
        blq = o.energy_bias((1, 2), (0.3, -0.2), given=(oj, oi))[2]       # the inverted action: j now carries species 1
        want = stream.log_uniform() < -(e2 - e1) / 0.7 + (blq - olq)
        assert ok == want
        if not want:
            o.swap_revert(oj, oi, 1, 2, de)
        acc += ok
        stream.advance()
        assert g.energy[0] == o.energy
    assert 0 < acc <= 30 and np.count_nonzero(g.species == 2) == np.count_nonzero(sp == 2)
    assert np.array_equal(g.species, o.species())


def test_soft_spheres_odd_exponent(gpu):
    """SoftSpheres with odd n takes the Float64 exponent branch (src/models.jl:66: ndiv2 = n / 2). Julia's own
    Float64^Float64 cannot be restated without its tables (parity unpinned at the ulp level for this branch);
    host, device and oracle all evaluate sqrt(x^n) with the compensated integer power, so they agree bit for
    bit with each other, and with libm's pow to a couple of ulp."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(2197, frac_b=0.5, rho=0.8, jitter=0.05, seed=12)
    mm = [[pmc.SoftSpheres(1.0, s, 9) for s in row] for row in ([1.0, 1.1], [1.1, 1.2])]
    g = pmc.System(pos, sp, rho, 1.0, mm, seed=2)
    mo = O.model_matrix(lambda e, s: O.soft_spheres(e, s, 9), [[1.0, 1.0], [1.0, 1.0]], [[1.0, 1.1], [1.1, 1.2]])
    o = O.System(pos, sp, rho, 1.0, mo, O.LINKEDLIST)
    assert np.array_equal(g.get_cells(), o.cells())
    assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles()) and g.energy[0] == o.energy
    # against libm pow on the host: x^(9/2), a few ulp
    r2 = np.linspace(0.7, 6.0, 1001)
    mine = np.array([pmc.potential(x, mm[0][1]) for x in r2])
    ref = 1.0 * (1.1 ** 2 / r2) ** 4.5 - mm[0][1].shift
    assert np.allclose(mine, ref, rtol=1e-14, atol=1e-15)
    e, fix = g.total_energy()
    assert abs(e - o.energy) <= 1e-6 * abs(o.energy)
    cb = O.Checkerboard(pos, sp, o.box, 1.0, mo, seed=2)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=4)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 0, 4)
    assert st.attempted[0] == att[0] == 4 * len(pos) and st.accepted[0] == acc[0]
    assert 0.1 < st.accepted[0] / st.attempted[0] < 0.95
    assert np.array_equal(g.position, cb.positions())
    assert g.total_energy()[1] == st.energy_fix == cb.energy_fix


def test_engines_interleave_without_manual_rebuilds(gpu):
    """Checkerboard sweeps, then per-move verbs and reference-order energies, then sweeps again: the
    LinkedList is rebuilt on demand (in the reference it is always current) and both energy
    bookkeepings stay consistent with the oracle at the same positions."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=21)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=8)
    pmc.mc_sweep(g, [_disp(pmc)], nsweeps=3)
    p1 = g.position
    o = O.System(p1, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    assert pmc.compute_energy_particle(g, 5) == o.energy_particle(5)          # no explicit build_neighbour_list
    assert np.array_equal(g.get_cells(), o.cells())
    act = pmc.Displacement(7, np.array([0.3, -0.2, 0.1]), 0.0)
    e1, e2 = pmc.perform_action(g, act)
    r1, r2, rde = o.displacement_perform(7, np.array([0.3, -0.2, 0.1]))
    assert (e1, e2, act.de) == (r1, r2, rde)
    st = pmc.mc_sweep(g, [_disp(pmc)], nsweeps=2)                             # back to the throughput engine
    cb = O.Checkerboard(o.positions(), sp, o.box, 1.0, o.models, seed=8)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 3, 2)
    assert st.accepted[0] == acc[0] and np.array_equal(g.position, cb.positions())
    assert st.energy_fix == cb.energy_fix


def test_ticket_fallback_after_a_starved_launch(gpu, monkeypatch):
    """when a fused launch is declared starved the handle falls back from the eight per-XCD ticket counters to the
    single monotonic one -- which must restart in step with the host's count, not at the aborted sweep's.  The fault injection
    pmc_debug_inject(PMC_INJECT_STARVED_LAUNCH) declares the handle's next good launch starved (one shot; ADVICE r04: it used to
    be an environment variable read on every call); afterwards the chain (re-uploaded, as the error text asks) must follow the
    oracle's schedule again, alone and in a shared launch."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
    pool = [_disp(pmc, 0.05, 1.0)]
    pool_o = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=9)
    pmc.mc_sweep(g, pool, nsweeps=2)
    g.debug_inject(1)
    with pytest.raises(pmc.PMCError):
        pmc.mc_sweep(g, pool, nsweeps=1)
    g.upload(pos, sp)
    g.set_counters(0, 0)
    cb = O.Checkerboard(pos, sp, g.box, 1.0, O.single(O.lennard_jones(1.0, 1.0)), seed=9)
    st = pmc.mc_sweep(g, pool, nsweeps=3)
    att, acc = cb.sweeps(pool_o, 0, 3)
    assert st.accepted[0] == acc[0] and np.array_equal(g.position, cb.positions()) and st.energy_fix == cb.energy_fix
    # ... and every chain of a shared launch
    chains = [pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=9, chain=k) for k in range(3)]
    pmc.mc_sweep_many(chains, pool, nsweeps=1)
    chains[0].debug_inject(1)
    with pytest.raises(pmc.PMCError):
        pmc.mc_sweep_many(chains, pool, nsweeps=1)
    for k, c in enumerate(chains):
        c.upload(pos, sp)
        c.set_counters(0, 0)
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=2)
    for k, c in enumerate(chains):
        cbk = O.Checkerboard(pos, sp, c.box, 1.0, O.single(O.lennard_jones(1.0, 1.0)), seed=9, chain=k)
        att, acc = cbk.sweeps(pool_o, 0, 2)
        assert sts[k].accepted[0] == acc[0] and np.array_equal(c.position, cbk.positions())
