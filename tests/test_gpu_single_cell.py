"""-m gpu: the throughput engine on boxes of 2 ... 4 cutoffs per side (round 5).  Such a box cannot be two-coloured periodically
(`PMC_ERR_UNSUPPORTED` until round 4); when it is small in every dimension and holds at most 1024 particles the whole periodic box
is ONE cell: one colour, one visit per pass, its N proposal slots in order against all N particles, evaluated eight at a time by a
team with the exact pair corrections of any batch -- the plain Metropolis chain of src/moves.jl:57-81 with a uniformly picked
particle and the reference's global DoubleUniform (src/moves.jl:226-241), keyed (cell 0, slot t, sweep g).  The reference's own
validation example is such a box: N = 1000 binary LJ, rcut = 4 sigma (examples/lj-mixture/run-validation.py:26-34,155-166).
Everything bit-exact against the oracle's restatement (orc_cb_create / orc_cb_sweeps_rounds with n_mc = 1)."""
import numpy as np
import pytest

from inputs import lj_lattice
from mixture import EPS, SIG, RCUT, N as NMIX, make_config, reference_rows

pytestmark = pytest.mark.gpu


def _disp(pmc, p=1.0, sigma=0.05):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def _mixture(pmc, O):
    mm = [[pmc.LennardJones(EPS[i][j], SIG[i][j], rcut=RCUT, shift_potential=False) for j in range(2)] for i in range(2)]
    mo = O.model_matrix(O.lennard_jones, EPS, SIG, rcut=RCUT, shift_potential=False)
    return mm, mo


def _mix_pool(pmc, O, g):
    pool = [_disp(pmc, 0.9), pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.1)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.9}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.1}])
    return pool, opool


def _same(g, cb, st, acc, npool):
    assert list(st.accepted[:npool]) == list(acc)
    assert st.energy_fix == cb.energy_fix
    assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())


@pytest.mark.parametrize("rho_t,T,rounds", [(0.8, 1.5, 1), (0.3, 1.3, 1), (0.8, 1.5, 4), (0.1, 1.3, 2)])
def test_published_mixture_at_its_own_size_against_oracle(gpu, rho_t, T, rounds):
    """examples/lj-mixture as published (N = 1000: 2.7 cutoffs per side at rho = 0.8, 5.4 at rho = 0.1 -- there the ordinary grid of
    4 x 4 x 4 cells takes over), 90 % Displacement + 10 % DiscreteSwap: the device against the oracle, and the exact energy against the
    reference-order total."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = make_config(0.5, rho_t)
    mm, mo = _mixture(pmc, O)
    g = pmc.System(pos, sp, rho, T, mm, seed=42)
    cb = O.Checkerboard(pos, sp, np.asarray(g.box), T, mo, seed=42)
    n_mc = tuple(int(v) for v in g.grid()[0])
    assert n_mc == tuple(int(v) for v in cb.grid()[0]) and n_mc == ((1, 1, 1) if rho_t > 0.15 else (4, 4, 4))
    e, fix = g.total_energy()
    assert fix == cb.energy_fix and abs(e - g.energy[0]) <= 1e-9 * abs(g.energy[0])
    pool, opool = _mix_pool(pmc, O, g)
    g.rounds = rounds
    done = 0
    for n in (3, 2):
        st = pmc.mc_sweep(g, pool, nsweeps=n)
        att, acc = cb.sweeps(opool, done, n, rounds=rounds)
        done += n
        assert list(st.attempted[:2]) == list(att)
        _same(g, cb, st, acc, 2)
    assert acc[1] > 20 and g.total_energy()[1] == cb.energy_fix


@pytest.mark.parametrize("team", ["auto", "1", "2", "4"])
@pytest.mark.parametrize("n,rc", [(8, 3.0), (6, 2.5)])
def test_single_species_every_team_width(gpu, monkeypatch, n, rc, team):
    """N = 512 (the register copy of the pre-filter holds exactly that) and N = 216, every team width: the one cell visit runs
    on one wavefront (cells of more than 64 particles: slot values in chunks) or as a team's speculative batches."""
    from oracle import oracle as O
    pmc = gpu
    if team == "auto":
        monkeypatch.delenv("PMC_TEAM", raising=False)
    else:
        monkeypatch.setenv("PMC_TEAM", team)
    pos, sp, rho = lj_lattice(n, jitter=0.05, seed=3, rho=0.9)
    g = pmc.System(pos, sp, rho, 1.2, pmc.LennardJones(1.0, 1.0, rcut=rc), seed=5)
    cb = O.Checkerboard(pos, sp, np.asarray(g.box), 1.2, O.single(O.lennard_jones(1.0, 1.0, rc)), seed=5)
    assert tuple(int(v) for v in g.grid()[0]) == (1, 1, 1)
    g.rounds = 2
    st = pmc.mc_sweep(g, [_disp(pmc, 1.0, 0.08)], nsweeps=5)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.08}]), 0, 5, rounds=2)
    assert st.attempted[0] == att[0] == 5 * len(pos) and st.null_moves == 0          # nothing leaves a single-cell box
    _same(g, cb, st, acc, 1)


def test_two_dimensions_images_delta_e_and_shared_launches(gpu):
    """2-D binary soft spheres with a long cutoff (box 26.8 = 2.98 cutoffs), started from positions shifted by whole boxes
    (periodic images: the default instances, direct loop); pmc_delta_e on the one cell; eight chains of a shared launch."""
    from oracle import oracle as O
    pmc = gpu
    side, rho2 = 24, 0.8
    N = side * side
    L = (N / rho2) ** 0.5
    gq = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    pos = (gq + 0.5) * (L / side) + np.random.default_rng(1).uniform(-0.05, 0.05, (N, 2))
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(2).permutation(N)[: N // 2]] = 2
    mm = [[pmc.SoftSpheres(1.0, s_, 12, rcut=9.0) for s_ in row] for row in ([1.0, 1.2], [1.2, 1.4])]
    mo = O.model_matrix(lambda e, s_: O.soft_spheres(e, s_, 12, 9.0), ((1.0, 1.0), (1.0, 1.0)), ((1.0, 1.2), (1.2, 1.4)))
    pm = [_disp(pmc, 0.7, 0.065), None]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.065, "probability": 0.7}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.3}])
    for shifted in (False, True):
        p = pos + (np.random.default_rng(7).integers(-2, 3, size=pos.shape) * L if shifted else 0.0)
        g = pmc.System(p, sp, rho2, 1.0, mm, seed=9)
        pm[1] = pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.3)
        cb = O.Checkerboard(p, sp, np.asarray(g.box), 1.0, mo, seed=9)
        assert tuple(int(v) for v in g.grid()[0][:2]) == (1, 1) and g.total_energy()[1] == cb.energy_fix
        if not shifted:
            rng = np.random.default_rng(3)
            ids = rng.integers(1, N + 1, 16)
            deltas = rng.normal(0, 0.5, (16, 2))
            e1, e2 = g.delta_e(ids, deltas)
            for k in range(16):
                i = int(ids[k])
                assert e1[k] == cb.energy_at(i, p[i - 1], int(sp[i - 1])) and e2[k] == cb.energy_at(i, p[i - 1] + deltas[k], int(sp[i - 1]))
        st = pmc.mc_sweep(g, pm, nsweeps=4)
        att, acc = cb.sweeps(po, 0, 4)
        _same(g, cb, st, acc, 2)
    chains = [pmc.System(pos, sp, rho2, 1.0 + 0.05 * k, mm, seed=9, chain=k) for k in range(8)]
    pm[1] = pmc.Move(pmc.DiscreteSwap([1, 2], chains[0]), pmc.DoubleUniform(), [], 0.3)
    sts = pmc.mc_sweep_many(chains, pm, nsweeps=3)
    for k, c in enumerate(chains):
        cbk = O.Checkerboard(pos, sp, np.asarray(c.box), 1.0 + 0.05 * k, mo, seed=9, chain=k)
        att, acc = cbk.sweeps(po, 0, 3)
        _same(c, cbk, sts[k], acc, 2)


def test_boxes_that_stay_refused(gpu):
    """Below two cutoffs per side the two roundings of the minimum image can differ inside the cutoff, above 1024 particles the
    one cell does not fit a visit's staging: both stay PMC_ERR_UNSUPPORTED (the reference-order engines run them)."""
    from particlesmc_amd import _lib
    pmc = gpu
    pool = [_disp(pmc)]
    pos, sp, rho = lj_lattice(5, jitter=0.05, seed=1, rho=1.0)            # box 5 = 1.67 cutoffs of 3.0
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0, rcut=3.0), seed=1)
    with pytest.raises(pmc.PMCError) as ei:
        pmc.mc_sweep(g, pool, nsweeps=1)
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED
    pos, sp, rho = lj_lattice(11, jitter=0.05, seed=1, rho=0.8)           # N = 1331 in a box of 2.96 cutoffs of 4.0
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0, rcut=4.0), seed=1)
    with pytest.raises(pmc.PMCError) as ei:
        pmc.mc_sweep(g, pool, nsweeps=1)
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED
    st = pmc.mc_sweep(g, pool, nsweeps=1, mode="sequential")              # ... and the reference's own chain runs
    assert st.attempted[0] == len(pos)


@pytest.mark.parametrize("row_index", list(range(23)))
def test_published_mixture_thermodynamics_at_the_references_own_size(gpu, row_index):
    """The reference's validation protocol at the reference's own N = 1000 (run-validation.py:155-180: 1000 sweeps from the lattice,
    second half averaged), every row of calculated-energies.csv, on the throughput engine -- which is the plain Metropolis chain
    with global DoubleUniform picks here, so BOTH acceptance rates are compared as published (no cell-local correction; the rows at
    rho = 0.1 run the ordinary 4 x 4 x 4 grid and keep the bounds of tests/test_gpu_edge.py)."""
    from oracle import oracle as O
    pmc = gpu
    row = reference_rows()[row_index]
    pos, sp, rho = make_config(row["x"], row["density"], seed=row_index)
    mm, _ = _mixture(pmc, O)
    s = pmc.System(pos, sp, rho, row["t"], mm, seed=42 + row_index)
    single = tuple(int(v) for v in s.grid()[0]) == (1, 1, 1)
    pool = [_disp(pmc, 0.9), pmc.Move(pmc.DiscreteSwap([1, 2], s), pmc.DoubleUniform(), [], 0.1)]
    s.total_energy()
    pmc.mc_sweep(s, pool, nsweeps=1500 if row["density"] <= 0.5 else 500)
    en, att, acc = [], np.zeros(2), np.zeros(2)
    for _ in range(100):
        st = pmc.mc_sweep(s, pool, nsweeps=10)
        att += st.attempted[:2]
        acc += st.accepted[:2]
        en.append(st.energy / NMIX)
    en = np.array(en)
    err = en.reshape(10, -1).mean(1).std(ddof=1) / np.sqrt(10)
    assert abs(en.mean() - row["energy"]) < 4 * np.hypot(err, row["energy_err"]) + 0.01
    if single:
        assert st.null_moves == 0
        assert abs(acc[0] / att[0] - row["acceptance_rate_displacement"]) < 0.02
        assert abs(acc[1] / att[1] - row["acceptance_rate_swap"]) < 0.03
    assert s.total_energy()[1] == st.energy_fix
