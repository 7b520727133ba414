"""-m gpu: engine 4 of the reference-order chain -- VerletList systems (src/neighbours.jl:387-543) with their state in LDS, one
wavefront resolving the steps in order while twelve others evaluate the steps behind them (csrc/pmc_verlet.h) -- against the
oracle's restatement of the reference's sequential chain (orc_mc_steps_sequential), bit for bit: positions, system.energy,
accept counts, species and the SpeciesList, cs, cells[c] and neighbours[i] in vector order.

The named systems are the reference's own: its CI system (test/runtests.jl:22-91: config_0, N = 1290 2-D ternary JBB, VerletList
dr = 0.2, Displacement sigma = 0.05; :124-129 with 80 % DiscreteSwap).  The seeded fuzz draws what nobody named: dimension,
density, dr, species and model matrices, step widths up to most of a cell, swaps, the split into calls and the engine of each."""
import os

import numpy as np
import pytest

from inputs import lj_lattice
from test_gpu_fuzz import _models, _pools

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("PMC_VERLET_FUZZ_CASES", "40"))
FIRST = int(os.environ.get("PMC_VERLET_FUZZ_FIRST", "0"))


def _config0(O):
    from test_oracle_golden import _fold, _load
    pos, sp, box = _load("config_0.xyz")
    return _fold(O, pos, box), sp, len(pos) / np.prod(box)


def _disp(pmc, sigma, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


def _swap(pmc, species, counts, p):
    return pmc.Move(pmc.DiscreteSwap(0, 0, species, counts, 0.0), pmc.DoubleUniform(), {}, p)


def _same_state(g, o, S=1, structures=True):
    assert np.array_equal(g.position, o.positions())
    assert g.energy[0] == o.energy, (g.energy[0], o.energy)
    assert np.array_equal(g.species, o.species())
    for s_ in range(1, S + 1):
        assert np.array_equal(g.species_ids(s_), o.species_ids(s_)), s_
    if structures:
        assert np.array_equal(g.get_cells(), o.cells())
        for c in range(1, int(np.prod(o.cell_geometry()[1])) + 1):
            assert np.array_equal(g.get_cell_members(c), o.cell_members(c)), c
        for i in range(1, g.N + 1):
            assert np.array_equal(g.get_verlet_neighbours(i), o.neighbours_of(i)), i


@pytest.mark.parametrize("swaps", [False, True])
def test_engine4_runs_the_reference_ci_system(gpu, swaps):
    """test/runtests.jl:41-91 (Displacement only) and :124-129 (80 % DiscreteSwap) on engine 4, three calls: after each, every
    array the reference keeps is the oracle's, and the engine ran every step of the call itself."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _config0(O)
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=pmc.VerletList, list_parameters={"dr": 0.2}, seed=10)
    o = O.System(pos, sp, rho, 0.231, O.jbb(), O.VERLETLIST, 0.2)
    if swaps:
        nA, nB = int((sp == 1).sum()), int((sp == 2).sum())
        pm = [_disp(pmc, 0.05, 0.2), _swap(pmc, (1, 2), (nA, nB), 0.8)]
        po = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.2}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.8}])
    else:
        pm = [_disp(pmc, 0.05)]
        po = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    g.set_sequential_engine(4, 0)
    done = 0
    for nsw in (3, 1, 6):
        st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        att, acc = o.mc_steps(po, 10, 0, done, nsw * N)
        done += nsw * N
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc), (nsw, st.accepted, acc)
        _same_state(g, o, S=3)
        polls, alone, rebuilds, steps = g.verlet_info()
        assert steps == nsw * N                           # no list outgrew its LDS row: nothing fell back to the literal loop
        assert rebuilds > 0.1 * (nsw * N) * (0.2 if swaps else 1.0)     # sigma = 0.05 against dr / 2 = 0.1: a third of the Displacements re-list
        if not swaps:
            assert alone < 0.25 * nsw * N                  # most steps are taken from their evaluation


def test_engine4_alternates_with_the_literal_loop(gpu):
    """Engines 4, 1 and the default selection on one handle, call after call: the chain does not notice (the state between calls
    is the reference's arrays in global memory, whichever engine left them)."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _config0(O)
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=pmc.VerletList, list_parameters={"dr": 0.2}, seed=3, chain=2)
    o = O.System(pos, sp, rho, 0.231, O.jbb(), O.VERLETLIST, 0.2)
    nA, nC = int((sp == 1).sum()), int((sp == 3).sum())
    pm = [_disp(pmc, 0.08, 0.7), _swap(pmc, (1, 3), (nA, nC), 0.3)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.08, "probability": 0.7}, {"kind": "DiscreteSwap", "species": (1, 3), "probability": 0.3}])
    done = 0
    for engine in (4, 1, 0, 4):
        g.set_sequential_engine(engine, 0)
        st = pmc.mc_sweep(g, pm, nsweeps=2, mode="sequential")
        att, acc = o.mc_steps(po, 3, 2, done, 2 * N)
        done += 2 * N
        assert np.array_equal(st.accepted[:2], acc), (engine, st.accepted, acc)
        _same_state(g, o, S=3)
    # the per-move verbs (engine 1's kernels: a batch of one) on the state engine 4 left
    act = pmc.Displacement(7, np.array([0.03, -0.02]), 0.0)
    e1, e2 = pmc.perform_action(g, act)
    oe1, oe2, ode = o.displacement_perform(7, np.array([0.03, -0.02]))
    assert (e1, e2, act.de) == (oe1, oe2, ode)
    _same_state(g, o, S=3)


def test_engine4_many_chains_share_one_launch(gpu):
    """pmc_sweep_many on VerletList chains of one shape: one workgroup per chain in ONE launch of k_sq_verlet, every chain with its
    own start, temperature and random stream -- each the oracle's chain."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _config0(O)
    N = len(pos)
    rng = np.random.default_rng(5)
    M = 7
    starts = [pos + rng.uniform(-0.01, 0.01, size=pos.shape) for _ in range(M)]
    temps = [0.231 * (1.0 + 0.1 * k) for k in range(M)]
    gs = [pmc.System(starts[k], sp, rho, temps[k], pmc.JBB(), list_type=pmc.VerletList, list_parameters={"dr": 0.2}, seed=42, chain=k) for k in range(M)]
    os_ = [O.System(starts[k], sp, rho, temps[k], O.jbb(), O.VERLETLIST, 0.2) for k in range(M)]
    pm = [_disp(pmc, 0.05)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    done = 0
    for nsw in (2, 3):
        sts = pmc.mc_sweep_many(gs, pm, nsweeps=nsw, mode="sequential")
        for k in range(M):
            att, acc = os_[k].mc_steps(po, 42, k, done, nsw * N)
            assert sts[k].accepted[0] == acc[0], (k, sts[k].accepted, acc)
            assert sts[k].sweep_kernel_launches == 1
            _same_state(gs[k], os_[k], S=3, structures=(k in (0, M - 1)))
            assert gs[k].verlet_info()[3] == nsw * N
        done += nsw * N


@pytest.mark.parametrize("env,val", [("PMC_SQV_VCAP", "28"), ("PMC_SQV_VCAP", "32"), ("PMC_SQV_VCAP", "36"), ("PMC_SQV_CCAP", "12"), ("PMC_SQV_CCAP", "13")])
def test_engine4_gives_way_to_the_literal_loop_when_a_list_outgrows_its_row(gpu, env, val, monkeypatch):
    """The LDS rows are shorter than the device rows of the literal loop.  A neighbour list (config_0: up to 31 ids) or a cell
    vector (up to 12) that does not fit -- at the start of a launch, or when a rebuild appends to it -- ends the launch without
    a trace; the rest of the call runs on the literal loop, and the chain is the oracle's all the same.  (Rows of 28 ids cannot
    hold the lists the system starts with; the other capacities may or may not overflow within the two sweeps -- rows of 32 and
    36 ids also run the general form of leaving a list, which rows of 40 ids do in registers.)"""
    from oracle import oracle as O
    pmc = gpu
    monkeypatch.setenv(env, val)
    pos, sp, rho = _config0(O)
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=pmc.VerletList, list_parameters={"dr": 0.2}, seed=10)
    o = O.System(pos, sp, rho, 0.231, O.jbb(), O.VERLETLIST, 0.2)
    nA, nB = int((sp == 1).sum()), int((sp == 2).sum())
    pm = [_disp(pmc, 0.05, 0.8), _swap(pmc, (1, 2), (nA, nB), 0.2)]
    po = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.8}, {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    st = pmc.mc_sweep(g, pm, nsweeps=2, mode="sequential")
    att, acc = o.mc_steps(po, 10, 0, 0, 2 * N)
    assert np.array_equal(st.attempted[:2], att) and np.array_equal(st.accepted[:2], acc), (st.accepted, acc)
    _same_state(g, o, S=3)
    if (env, val) == ("PMC_SQV_VCAP", "28"):
        assert g.verlet_info()[3] == 0                     # engine 4 ran nothing of the call
    assert g.verlet_info()[3] <= 2 * N
    monkeypatch.delenv(env)
    st = pmc.mc_sweep(g, pm, nsweeps=1, mode="sequential")  # ... and takes the chain back with rows of the usual length
    att, acc = o.mc_steps(po, 10, 0, 2 * N, N)
    assert np.array_equal(st.accepted[:2], acc)
    _same_state(g, o, S=3)
    assert g.verlet_info()[3] == N


def test_engine4_is_refused_where_it_does_not_apply(gpu):
    """Engine 4 is for VerletList systems whose state fits the LDS: asked for on a LinkedList system, or on a VerletList system of
    20 000 particles, it is refused (PMC_ERR_UNSUPPORTED); the default selection runs such a VerletList system on the literal loop."""
    from oracle import oracle as O
    from particlesmc_amd import _lib
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=1, rho=0.9, d=3)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=1)
    with pytest.raises(_lib.PMCError) as ei:
        g.set_sequential_engine(4, 0)
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED
    pos, sp, rho = lj_lattice(142, jitter=0.05, seed=2, rho=0.9, d=2)           # N = 20 164
    N = len(pos)
    g = pmc.System(pos, sp, rho, 0.7, pmc.LennardJones(1.0, 1.0), list_type=pmc.VerletList, list_parameters={"dr": 0.3}, seed=8)
    with pytest.raises(_lib.PMCError) as ei:
        g.set_sequential_engine(4, 0)
    assert ei.value.code == _lib.PMC_ERR_UNSUPPORTED
    o = O.System(pos, sp, rho, 0.7, O.single(O.lennard_jones(1.0, 1.0)), O.VERLETLIST, 0.3)
    st = pmc.mc_sweep(g, [_disp(pmc, 0.1)], nsweeps=1, mode="sequential")
    att, acc = o.mc_steps(O.make_pool([{"kind": "Displacement", "sigma": 0.1}]), 8, 0, 0, N)
    assert st.accepted[0] == acc[0]
    assert np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy


def _draw(pmc, O, case):
    for attempt in range(50):
        rng = np.random.default_rng(313000 + case + 100000 * attempt)
        d = int(rng.choice([2, 3], p=[0.6, 0.4]))
        n = int(rng.integers(8, 40)) if d == 2 else int(rng.integers(4, 10))
        rho = float(np.exp(rng.uniform(np.log(0.08), np.log(1.3)))) if rng.random() < 0.3 else float(rng.uniform(0.5, 1.3))
        S = int(rng.choice([1, 2, 3]))
        pos, sp, _ = lj_lattice(n, jitter=0.08 * rho ** (-1.0 / d), seed=case, rho=rho, d=d)
        N = len(pos)
        if S > 1:
            sp = rng.integers(1, S + 1, size=N).astype(np.int32)
            sp[:S] = np.arange(1, S + 1)
        far = int(rng.choice([0, 1, 3]))
        if far:
            pos = pos + rng.integers(-far, far + 1, size=pos.shape) * (N / rho) ** (1.0 / d)
        T = float(np.exp(rng.uniform(np.log(0.1), np.log(5.0))))
        seed, chain = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 8))
        mm, mo = _models(pmc, O, rng, S)
        dr = float(rng.choice([0.05, 0.1, 0.2, 0.4, 0.8]))
        try:
            O.System(pos, sp, rho, T, mo, O.VERLETLIST, dr)
        except ValueError as e:
            assert "bad arguments" in str(e), e
            continue
        return rng, d, S, pos, sp, rho, T, seed, chain, mm, mo, dr
    raise AssertionError(f"case {case}: no acceptable geometry in 50 draws")


@pytest.mark.parametrize("case", range(FIRST, FIRST + NCASES))
def test_random_verlet_chain(gpu, case):
    """Seeded random VerletList chains: 2-D and 3-D, dilute to dense, dr from a twentieth of a cutoff to a third (rebuilds on nearly
    every step to hardly any), rows short and long (3-D rows of more than 64 ids take the general paths of the sums and of
    leaving a list), 1-3 species with random model matrices (the compiled-in potentials and the general kernel), pools with
    swaps, steps wide enough to cross cells.  One to three calls, the engine drawn per call (4, the default, 1 for small
    systems); a system whose state does not fit the LDS is refused by engine 4 and runs on the default selection."""
    from oracle import oracle as O
    from particlesmc_amd import _lib
    pmc = gpu
    rng, d, S, pos, sp, rho, T, seed, chain, mm, mo, dr = _draw(pmc, O, case)
    N = len(pos)
    g = pmc.System(pos, sp, rho, T, mm, list_type=pmc.VerletList, list_parameters={"dr": dr}, seed=seed, chain=chain)
    o = O.System(pos, sp, rho, T, mo, O.VERLETLIST, dr)
    assert g.energy[0] == o.energy
    pm, po = _pools(pmc, rng, S, g)
    if rng.random() < 0.5:
        pm[0].parameters["sigma"] = po[0]["sigma"] = float(rng.uniform(0.2, 0.7))
    pool_o = O.make_pool(po)
    done = 0
    ncalls = int(rng.integers(1, 4))
    for call in range(ncalls):
        engine = int(rng.choice([4, 4, 0, 1]))
        nsw = int(rng.integers(1, 4))
        if engine == 1 and N * nsw > 3000:
            engine = 0
        try:
            g.set_sequential_engine(engine, 0)
        except _lib.PMCError as e:
            assert engine == 4 and e.code == _lib.PMC_ERR_UNSUPPORTED, (engine, e)
            g.set_sequential_engine(0, 0)
        st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        att, acc = o.mc_steps(pool_o, seed, chain, done, nsw * N)
        done += nsw * N
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc), (call, engine, st.accepted, acc)
        _same_state(g, o, S=S, structures=(call == ncalls - 1))
