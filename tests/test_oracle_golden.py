"""CPU: the oracle against the reference's own goldens and known answers.

These pin the oracle before it is trusted as the checker of the HIP path."""
import math
import os

import numpy as np
import pytest

from inputs import lj_lattice

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle
    oracle.lib()
    return oracle


def _load(name):
    """Minimal reader, independent of particlesmc_amd.io: returns (pos, species, box)."""
    lines = open(os.path.join(GOLD, name)).read().split("\n")
    if name.endswith(".lmp"):
        start = next(k for k, l in enumerate(lines) if l.startswith("ITEM: ATOMS")) + 1
        n = int(lines[3])
        lo, hi = [float(v) for v in lines[5].split()]
        box = np.array([hi - lo, hi - lo])
    else:
        n = int(lines[0])
        start = 2
        box = np.array([32.8962, 32.8962])
    rows = np.array([[float(v) for v in l.split()] for l in lines[start:start + n]])
    return rows[:, 1:3].copy(), rows[:, 0].astype(np.int32), box


def _fold(O, pos, box):
    L = O.lib()
    out = pos.copy()
    for i in range(pos.shape[0]):
        for a in range(pos.shape[1]):
            out[i, a] = pos[i, a] - L.orc_jl_fld(pos[i, a], box[a]) * box[a]     # utils.jl:12
    return out


@pytest.mark.parametrize("fname", ["config_0.xyz", "config_0.exyz", "config_0.lmp"])
@pytest.mark.parametrize("lt", [0, 1])
def test_jbb_known_answer(O, fname, lt):
    """test/runtests.jl:36-38: energy/N == -2.676832 (atol 1e-6) for EmptyList and LinkedList."""
    pos, sp, box = _load(fname)
    assert len(pos) == 1290 and sorted(np.bincount(sp)[1:]) == [330, 360, 600]
    rho = len(pos) / np.prod(box)
    if fname != "config_0.lmp":
        assert rho == 1.192060351952634            # test/params.toml density
    s = O.System(_fold(O, pos, box), sp, rho, 0.231, O.jbb(), lt)
    assert abs(s.energy / s.N - (-2.676832)) < 1e-6


def test_lists_agree_on_initial_state(O):
    pos, sp, box = _load("config_0.xyz")
    rho = len(pos) / np.prod(box)
    p = _fold(O, pos, box)
    el = O.System(p, sp, rho, 0.231, O.jbb(), O.EMPTYLIST)
    ll = O.System(p, sp, rho, 0.231, O.jbb(), O.LINKEDLIST)
    assert abs(el.energy - ll.energy) < 1e-9
    e1, e2 = el.energy_particles(), ll.energy_particles()
    assert np.allclose(e1, e2, rtol=0, atol=1e-11)


def _trace(O, lt, pool, sweeps, seed=10, dr=0.0):
    pos, sp, box = _load("config_0.xyz")
    rho = len(pos) / np.prod(box)
    s = O.System(_fold(O, pos, box), sp, rho, 0.231, O.jbb(), lt, dr)
    out = [s.energy / s.N]
    att = acc = 0
    for t in range(sweeps):
        a, c = s.mc_steps(pool, seed, 0, t * s.N, s.N)
        att, acc = att + a, acc + c
        out.append(s.energy / s.N)
    return np.array(out), att, acc, s


def test_same_seed_traces_equal_across_list_types_displacement(O):
    """test/runtests.jl:83-91 (differential): same seed, Displacement sigma = 0.05."""
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    el, att_e, acc_e, _ = _trace(O, O.EMPTYLIST, pool, 8)
    ll, att_l, acc_l, _ = _trace(O, O.LINKEDLIST, pool, 8)
    assert np.allclose(el, ll, rtol=0, atol=1e-6)
    assert np.array_equal(acc_e, acc_l) and acc_e[0] > 0
    # the Verlet leg of the same reference test (list_parameters dr = 0.2) and the CellList
    vl, _, acc_v, sv = _trace(O, O.VERLETLIST, pool, 8, dr=0.2)
    cl, _, acc_c, sc = _trace(O, O.CELLLIST, pool, 8)
    assert np.allclose(ll, vl, rtol=0, atol=1e-6) and np.allclose(ll, cl, rtol=0, atol=1e-6)
    assert np.array_equal(acc_l, acc_v) and np.array_equal(acc_l, acc_c)
    assert abs(sv.energy - sv.total_energy()) < 1e-8 * abs(sv.energy)


def test_verlet_and_cell_list_goldens_and_invariants(O):
    """test/runtests.jl:24,38: energy/N = -2.676832 +- 1e-6 with VerletList(dr = 0.2); the same for the
    CellList.  Invariants after moves: every Verlet list holds exactly the particles within rcut + dr
    of the positions of the last update or closer than rcut now (no in-range pair is ever missed), lists are
    symmetric and duplicate-free, cells[] partitions the particles and agrees with cs."""
    pos, sp, box = _load("config_0.xyz")
    rho = len(pos) / np.prod(box)
    p = _fold(O, pos, box)
    vl = O.System(p, sp, rho, 0.231, O.jbb(), O.VERLETLIST, 0.2)
    cl = O.System(p, sp, rho, 0.231, O.jbb(), O.CELLLIST)
    assert abs(vl.energy / vl.N - (-2.676832)) < 1e-6
    assert abs(cl.energy / cl.N - (-2.676832)) < 1e-6
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.15}])       # large steps: many list updates
    for s in (vl, cl):
        s.mc_steps(pool, 3, 0, 0, 4 * s.N)
        cs = s.cells()
        _, ncells = s.cell_geometry()
        seen = []
        for c in range(1, int(np.prod(ncells)) + 1):
            m = s.cell_members(c)
            assert np.all(cs[m - 1] == c)
            seen.extend(m.tolist())
        assert sorted(seen) == list(range(1, s.N + 1))
    x = vl.positions()
    L = np.array(vl.box)
    nbs = [set(vl.neighbours_of(i).tolist()) for i in range(1, vl.N + 1)]
    assert all(len(vl.neighbours_of(i)) == len(nbs[i - 1]) for i in range(1, vl.N + 1))     # no duplicates
    rc = 2.5 * 1.0                                                                                   # largest JBB cutoff is below this
    for i in range(0, vl.N, 7):
        d = x - x[i]
        d -= np.round(d / L) * L
        r = np.sqrt((d * d).sum(axis=1))
        close = set((np.nonzero(r < 1.5)[0] + 1).tolist()) - {i + 1}
        assert close <= nbs[i]                                    # nobody inside the interaction range is missing
        assert all((i + 1) in nbs[j - 1] for j in nbs[i])         # symmetry
    assert abs(vl.energy - vl.total_energy()) < 1e-8 * abs(vl.energy)


def test_same_seed_traces_equal_across_list_types_swaps(O):
    """test/runtests.jl:93-129: 20 % Displacement + 40 % swap(1,3) + 40 % swap(2,3)."""
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.2},
                        {"kind": "DiscreteSwap", "species": (1, 3), "probability": 0.4},
                        {"kind": "DiscreteSwap", "species": (2, 3), "probability": 0.4}])
    el, _, acc_e, se = _trace(O, O.EMPTYLIST, pool, 4)
    ll, _, acc_l, sl = _trace(O, O.LINKEDLIST, pool, 4)
    assert np.allclose(el, ll, rtol=0, atol=1e-6)
    assert np.array_equal(acc_e, acc_l) and acc_e[1] > 0
    assert np.array_equal(se.species(), sl.species())
    assert sorted(np.bincount(sl.species())[1:]) == [330, 360, 600]


def test_running_energy_tracks_recomputation(O):
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    _, _, _, s = _trace(O, O.LINKEDLIST, pool, 3)
    assert abs(s.energy - s.total_energy()) < 1e-8 * abs(s.energy)


def test_analytic_known_answers(O):
    m = O.lennard_jones(1.0, 1.0)
    assert m.shift == -0.016316891136000003               # 4 (2.5^-12 - 2.5^-6), SURVEY 7.1
    n = 8
    g = np.array([[i, j, k] for i in range(n) for j in range(n) for k in range(n)], dtype=float)
    s = O.System(g, np.ones(len(g), dtype=np.int32), 1.0, 1.0, O.single(m), O.LINKEDLIST)
    # shells r^2 = 1..6 with multiplicities 6, 12, 8, 6, 24, 24 (r^2 = 6.25 is the cutoff)
    ref = 0.0
    for r2, mult in zip((1, 2, 3, 4, 5, 6), (6, 12, 8, 6, 24, 24)):
        x3 = (1.0 / r2) ** 3
        ref += mult * (4.0 * (x3 * x3 - x3) - m.shift)
    assert abs(s.energy / len(g) - ref / 2) < 1e-12
    assert abs(s.energy / len(g) - (-3.329660801490727)) < 1e-12


def test_philox_known_answers(O):
    """Random123 kat_vectors for philox4x32-10."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2)[:2] == [0x408f276d, 0x41c83b0e]


def test_deterministic_math_accuracy(O):
    rng = np.random.default_rng(1)
    L = O.lib()
    for w in rng.integers(0, 2 ** 32, 5000):
        s, c = O.det_sincos2pi(int(w))
        th = 2 * math.pi * int(w) / 2 ** 32
        assert abs(s - math.sin(th)) < 2e-15 and abs(c - math.cos(th)) < 2e-15
    for x in rng.random(5000):
        assert abs(L.orc_det_log(x) - math.log(x)) <= 4e-16 * abs(math.log(x)) + 1e-300
    # antisymmetry under theta -> theta + pi keeps the Gaussian proposal exactly symmetric
    for w in (1, 12345, 0x7fffffff, 0x12345678):
        s1, c1 = O.det_sincos2pi(w)
        s2, c2 = O.det_sincos2pi((w + 0x80000000) & 0xffffffff)
        assert s1 == -s2 and c1 == -c2
    z = np.array([O.box_muller(int(a), int(b)) for a, b in rng.integers(0, 2 ** 32, (20000, 2))]).ravel()
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02


def test_julia_fld_semantics(O):
    L = O.lib()
    assert L.orc_jl_fld(-0.3, 2.5) == -1.0 and L.orc_jl_fld(2.5, 2.5) == 1.0 and L.orc_jl_fld(0.0, 2.5) == 0.0
    assert L.orc_jl_fld(7.4999999999999991, 2.5) == 2.0
    assert L.orc_jl_fld(-2.5, 2.5) == -1.0 and L.orc_jl_fld(-2.5000000000000004, 2.5) == -2.0
    assert L.orc_jl_fld_i(-1, 13) == -1 and L.orc_jl_fld_i(13, 13) == 1 and L.orc_jl_fld_i(-13, 13) == -1
    # N = 10^6 -> 39 cells per side, N = 32768 -> 12
    for N, nc in ((10 ** 6, 39), (32768, 12)):
        box = (N / 1.0) ** (1.0 / 3.0)
        cell = box / L.orc_jl_fld(box, 2.5)
        assert int(box / cell) == nc


def test_pow_int_matches_exact_power(O):
    from fractions import Fraction
    L = O.lib()
    rng = np.random.default_rng(2)
    for x in rng.uniform(0.05, 3.0, 300):
        for n in (6, 7, 9):
            exact = float(Fraction(float(x)) ** n)
            got = L.orc_pow_int(float(x), n)
            assert abs(got - exact) <= abs(exact) * 2.3e-16


def test_neighbour_cells_unique_when_few_cells(O):
    """build_neighbour_cells (neighbours.jl:94-111): unique! when < 3 cells per dimension."""
    pos, sp, rho = lj_lattice(6, jitter=0.0)             # box 6 -> 2 cells per side
    s = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    _, nc = s.cell_geometry()
    assert tuple(nc) == (2, 2, 2)
    for c in range(1, 9):
        nb = s.neighbour_cells(c)
        assert len(nb) == 8 and sorted(nb) == list(range(1, 9))
    # first-dimension-fastest offsets (Iterators.product), last-dimension-fastest scalar index
    pos, sp, rho = lj_lattice(11, jitter=0.0)            # box 11 -> 4 cells per side
    s = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    nb = s.neighbour_cells(1 + (1 * 16 + 1 * 4 + 1))      # cell (1,1,1)
    assert list(nb[:4]) == [1, 17, 33, 5]                 # (0,0,0),(1,0,0),(2,0,0),(0,1,0)


def test_linked_list_invariants_after_moves(O):
    pos, sp, rho = lj_lattice(8, jitter=0.1, seed=4)
    s = O.System(pos, sp, rho, 2.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    s.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.4}]), 5, 0, 0, 4 * s.N)
    head, lst = s.linked_list()
    cs = s.cells()
    seen = np.zeros(s.N, dtype=int)
    for c, h in enumerate(head, start=1):
        j = h
        while j != -1:
            seen[j - 1] += 1
            assert cs[j - 1] == c
            j = lst[j - 1]
    assert np.all(seen == 1)
    # a rebuilt list holds the same membership
    s2 = O.System(s.positions(), sp, rho, 2.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    assert np.array_equal(s2.cells(), cs)


def test_nonfinite_initial_energy_is_an_error(O):
    pos, sp, rho = lj_lattice(6)
    pos[1] = pos[0]
    with pytest.raises(ValueError, match="infinite or NaN"):
        O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)


def test_checkerboard_exact_energy_bookkeeping(O):
    pos, sp, rho = lj_lattice(11, jitter=0.1, seed=6)
    ref = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, ref.box, 1.0, ref.models, seed=77)
    e0 = cb.energy_fix
    assert abs(e0 / 2.0 ** 32 - ref.energy) <= 1e-9 * abs(ref.energy)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 0, 5)
    assert att[0] == 5 * len(pos) and 0 < acc[0] < att[0]
    assert cb.energy_fix == cb.total_energy_fix()          # exact, not approximate
    ref2 = O.System(cb.positions(), sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    assert abs(cb.energy_fix / 2.0 ** 32 - ref2.energy) <= 1e-9 * abs(ref2.energy)


def test_min_image_reciprocal_form_gives_identical_energies(O):
    """DESIGN.md "min-image": dx - rint(dx * (1/L)) * L changes no pair energy when L >= 4 rcut."""
    pos, sp, rho = lj_lattice(12, jitter=0.3, seed=8)
    L = (len(pos) / rho) ** (1 / 3)
    pos = pos + np.random.default_rng(0).integers(-3, 4, size=pos.shape) * L
    rc2, inv = 6.25, 1.0 / L
    rng = np.random.default_rng(1)
    i = rng.integers(0, len(pos), 200000)
    j = rng.integers(0, len(pos), 200000)
    dx = pos[i] - pos[j]
    a = dx - np.rint(dx / L) * L
    b = dx - np.rint(dx * inv) * L
    r2a, r2b = (a * a).sum(1), (b * b).sum(1)
    inr = (r2a <= rc2) | (r2b <= rc2)
    assert inr.sum() > 500 and np.array_equal(r2a[inr], r2b[inr])


@pytest.mark.timeout(600)
def test_checkerboard_statistics_match_sequential_chain(O):
    """The parallel schedule samples the same ensemble as the reference's sequential chain:
    <e> and the Displacement acceptance (after removing the cell-leaving rejections) agree."""
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=1)
    T = 1.5
    model = O.single(O.lennard_jones(1.0, 1.0))
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.08}])
    seq = O.System(pos, sp, rho, T, model, O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, seq.box, T, model, seed=3)
    N = len(pos)
    burn, prod = 60, 140
    seq.mc_steps(pool, 3, 0, 0, burn * N)
    cb.sweeps(pool, 0, burn)
    es, ec = [], []
    acc_s = acc_c = 0
    for t in range(prod):
        _, a = seq.mc_steps(pool, 3, 0, (burn + t) * N, N)
        acc_s += a[0]
        es.append(seq.energy / N)
        _, a = cb.sweeps(pool, burn + t, 1)
        acc_c += a[0]
        ec.append(cb.energy_fix / 2.0 ** 32 / N)
    es, ec = np.array(es), np.array(ec)
    # block errors ~ 0.01; means must agree within 4 sigma of the combined error
    def err(x):
        b = x[: len(x) // 10 * 10].reshape(10, -1).mean(1)
        return b.std(ddof=1) / np.sqrt(10)
    assert abs(es.mean() - ec.mean()) < 4 * np.hypot(err(es), err(ec)) + 5e-3
    a_s, a_c = acc_s / (prod * N), acc_c / (prod * N)
    assert abs(a_s - a_c) < 0.06 and a_c < a_s + 0.01


@pytest.mark.timeout(600)
def test_checkerboard_with_swaps_matches_sequential_chain(O):
    """The same comparison for a binary mixture with DiscreteSwap moves: the parallel schedule draws both swap partners
    from ONE cell (DoubleUniform restricted to the active domain), the reference chain from the whole box; <e> and BOTH
    acceptance rates must agree (Displacement up to the cell-leaving rejections)."""
    from mixture import EPS, SIG
    from inputs import binary_lattice
    pos, sp, rho = binary_lattice(1728, frac_b=0.5, rho=0.8, jitter=0.05, seed=2)
    T = 1.5
    models = O.model_matrix(lambda e, s: O.lennard_jones(e, s, 2.5 * s), EPS, SIG)
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.08, "probability": 0.8},
                        {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    seq = O.System(pos, sp, rho, T, models, O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, seq.box, T, models, seed=5)
    N = len(pos)
    burn, prod = 80, 160
    seq.mc_steps(pool, 5, 0, 0, burn * N)
    cb.sweeps(pool, 0, burn)
    es, ec = [], []
    att_s, acc_s, att_c, acc_c = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2)
    for t in range(prod):
        a, c = seq.mc_steps(pool, 5, 0, (burn + t) * N, N)
        att_s += a
        acc_s += c
        es.append(seq.energy / N)
        a, c = cb.sweeps(pool, burn + t, 1)
        att_c += a
        acc_c += c
        ec.append(cb.energy_fix / 2.0 ** 32 / N)
    es, ec = np.array(es), np.array(ec)

    def err(x):
        b = x[: len(x) // 10 * 10].reshape(10, -1).mean(1)
        return b.std(ddof=1) / np.sqrt(10)
    assert abs(es.mean() - ec.mean()) < 4 * np.hypot(err(es), err(ec)) + 5e-3
    rs, rc = acc_s / att_s, acc_c / att_c
    assert abs(rs[1] - rc[1]) < 0.03, (rs, rc)                      # swap acceptance: cell-local pairs vs box-wide pairs
    assert abs(rs[0] - rc[0]) < 0.06 and rc[0] < rs[0] + 0.01       # Displacement: lower only by the cell-leaving rejections
    assert abs(att_c[1] / att_c.sum() - 0.2) < 0.01


@pytest.mark.timeout(900)
@pytest.mark.parametrize("row_index", [0, 3, 6, 9, 13, 16, 18, 22])
def test_oracle_reproduces_the_references_published_mixture_results(O, row_index):
    """Statistical pin at the driver level: the reference's own results for
    examples/lj-mixture (calculated-energies.csv) -- mean energy and BOTH acceptance rates -- are
    reproduced by the oracle's sequential chain with the same protocol (N = 1000, rcut = 4,
    unshifted, 90/10 Displacement/DiscreteSwap, sigma = 0.05; here 600 sweeps, second half
    averaged).  This is the only reference data that constrains the Metropolis accept rule."""
    from mixture import EPS, SIG, RCUT, N, make_config, reference_rows
    row = reference_rows()[row_index]         # 8 of the 23 rows: all three compositions, rho = 0.1 ... 0.8, T = 1.2183 ... 1.6438
    pos, sp, rho = make_config(row["x"], row["density"], seed=1 + row_index)
    models = O.model_matrix(lambda e, s: O.lennard_jones(e, s, RCUT, False), EPS, SIG)
    s = O.System(pos, sp, rho, row["t"], models, O.LINKEDLIST)
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.9},
                        {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}])
    sweeps = 600
    en = []
    att = np.zeros(2)
    acc = np.zeros(2)
    for t in range(sweeps):
        a, c = s.mc_steps(pool, 42, 0, t * N, N)
        att += a
        acc += c
        en.append(s.energy / N)
    e = np.array(en[sweeps // 2:])
    blocks = e[: len(e) // 10 * 10].reshape(10, -1).mean(1)
    err = blocks.std(ddof=1) / np.sqrt(10)
    assert abs(e.mean() - row["energy"]) < 4 * np.hypot(err, row["energy_err"]) + 0.02
    assert abs(acc[0] / att[0] - row["acceptance_rate_displacement"]) < 0.02
    assert abs(acc[1] / att[1] - row["acceptance_rate_swap"]) < 0.04
    assert abs(att[1] / att.sum() - 0.1) < 0.01


def test_checkerboard_rounds_and_periodic_images_keep_the_books_exact(O):
    """The schedule parameter m (passes of m sweeps under one grid shift) and the persistent periodic images of the
    restatement: the running energy equals a recomputation bit for bit for inputs in the box, shifted by a
    few boxes, hundreds of boxes, and beyond the 511-box limit of the fold; calls aligned to passes compose; m = 1 is the
    plain schedule; and a configuration shifted by whole boxes accepts what the in-box one accepts (its folded coordinates
    differ from it by rounding only)."""
    from inputs import lj_lattice
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=5)
    N = len(pos)
    L = (N / rho) ** (1.0 / 3.0)
    box = np.array([L] * 3)
    models = O.single(O.lennard_jones(1.0, 1.0))
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.08}])
    accepted = {}
    for name, span in (("inbox", 0), ("near", 3), ("far", 400), ("beyond", 100000)):
        p = pos + (np.random.default_rng(7).integers(-span, span + 1, size=pos.shape) * L if span else 0.0)
        cb = O.Checkerboard(p, sp, box, 1.0, models, seed=3)
        att, acc = cb.sweeps(pool, 0, 12, rounds=4)
        assert att[0] == 12 * N and cb.energy_fix == cb.total_energy_fix(), name
        assert np.abs(cb.positions() - p).max() < 1.5, name            # nobody changed image
        accepted[name] = int(acc[0])
        # calls aligned to passes compose: 8 + 4 sweeps at m = 4 == 12 sweeps at m = 4
        cb2 = O.Checkerboard(p, sp, box, 1.0, models, seed=3)
        cb2.sweeps(pool, 0, 8, rounds=4)
        cb2.sweeps(pool, 8, 4, rounds=4)
        assert np.array_equal(cb2.positions(), cb.positions()) and cb2.energy_fix == cb.energy_fix, name
    assert abs(accepted["near"] - accepted["inbox"]) <= 3 and abs(accepted["far"] - accepted["inbox"]) <= 3, accepted
    a = O.Checkerboard(pos, sp, box, 1.0, models, seed=3)
    b = O.Checkerboard(pos, sp, box, 1.0, models, seed=3)
    a.sweeps(pool, 0, 5)
    b.sweeps(pool, 0, 5, rounds=1)
    assert np.array_equal(a.positions(), b.positions()) and a.energy_fix == b.energy_fix
    c = O.Checkerboard(pos, sp, box, 1.0, models, seed=3)
    c.sweeps(pool, 0, 5, rounds=2)
    assert not np.array_equal(a.positions(), c.positions())             # m = 2 is a different (equally valid) chain


def test_single_cell_boxes_in_the_oracle(O):
    """Round 5: the parallel schedule on a box of 2 ... 4 cutoffs per side is ONE cell spanning the periodic box (orc_cb_create,
    n_mc = 1) -- the plain Metropolis chain with a uniformly picked particle and global DoubleUniform picks.  On the reference's own
    validation example (N = 1000 binary LJ, rcut = 4: examples/lj-mixture/run-validation.py:26-34): the exact total energy equals
    the reference-order total, no slot is a null move, both acceptance rates and <e> agree with the reference's sequential chain
    (src/moves.jl:57-81, 187-241) from the same start; boxes below two cutoffs, mixed boxes and more than 1024 particles stay
    refused."""
    from mixture import EPS, SIG, RCUT, make_config
    pos, sp, rho = make_config(0.5, 0.8, seed=4)
    T = 1.5
    models = O.model_matrix(O.lennard_jones, EPS, SIG, rcut=RCUT, shift_potential=False)
    spec = [{"kind": "Displacement", "sigma": 0.05, "probability": 0.9}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}]
    pool = O.make_pool(spec)
    seq = O.System(pos, sp, rho, T, models, O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, seq.box, T, models, seed=11)
    n_mc, w = cb.grid()
    assert tuple(int(v) for v in n_mc) == (1, 1, 1) and np.array_equal(w, seq.box)
    N = len(pos)
    assert abs(cb.energy_fix / 2.0 ** 32 - seq.energy) <= 1e-9 * abs(seq.energy)
    burn, prod = 40, 60
    seq.mc_steps(pool, 11, 0, 0, burn * N)
    cb.sweeps(pool, 0, burn)
    es, ec = [], []
    att_s, acc_s, att_c, acc_c = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2)
    for t in range(prod):
        a, b = seq.mc_steps(pool, 11, 0, (burn + t) * N, N)
        att_s += a; acc_s += b
        es.append(seq.energy / N)
        a, b = cb.sweeps(pool, burn + t, 1)
        att_c += a; acc_c += b
        ec.append(cb.energy_fix / 2.0 ** 32 / N)
    assert att_c.sum() == prod * N                                   # every slot is an attempted move ...
    assert abs(acc_s[0] / att_s[0] - acc_c[0] / att_c[0]) < 0.02     # ... and none is a cell-leaving null move: same acceptance
    assert abs(acc_s[1] / att_s[1] - acc_c[1] / att_c[1]) < 0.06
    es, ec = np.array(es), np.array(ec)
    def err(x):
        b = x[: len(x) // 10 * 10].reshape(10, -1).mean(1)
        return b.std(ddof=1) / np.sqrt(10)
    assert abs(es.mean() - ec.mean()) < 4 * np.hypot(err(es), err(ec)) + 0.02
    # the energy the schedule keeps is exact: a recomputation from the final state gives the same integer
    assert cb.total_energy_fix() == cb.energy_fix
    # refused: below two cutoffs per side; more than 1024 particles in a small box
    small, sp_s, rho_s = lj_lattice(5, jitter=0.05, seed=1, rho=1.0)
    with pytest.raises(ValueError):
        O.Checkerboard(small, sp_s, np.array([5.0, 5.0, 5.0]), 1.0, O.single(O.lennard_jones(1.0, 1.0, 3.0)), seed=1)
    big, sp_b, rho_b = lj_lattice(11, jitter=0.05, seed=1, rho=0.8)
    Lb = (len(big) / 0.8) ** (1 / 3)
    with pytest.raises(ValueError):
        O.Checkerboard(big, sp_b, np.array([Lb, Lb, Lb]), 1.0, O.single(O.lennard_jones(1.0, 1.0, 4.0)), seed=1)
