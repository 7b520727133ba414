"""-m gpu: seeded random chains through the three engines of the reference-order chain (PMC_SEQUENTIAL: the literal loop,
the speculative windows, the chain state in LDS), bit for bit against the oracle's restatement of the reference's sequential
chain (src/moves.jl:11-20,57-81,187-207; oracle orc_mc_steps_sequential).

test_gpu_sequential.py pins named systems; this file draws what nobody named: dimension, size (3 to 22 cells per side),
density from a dilute gas to a dense liquid, temperature from frozen to hot, 1-3 species with random model matrices, pools
with one or two displacement widths up to most of a cell and swaps, LinkedList / CellList / EmptyList, particles stored
several boxes away from their folded image, the engine, its window, the team width of the LDS engine and the split of the
sweeps into calls.  Every case is reproducible from its number; PMC_SEQ_FUZZ_CASES=<n> (and PMC_SEQ_FUZZ_FIRST=<k>)
run a longer soak (the round's: cases 0 .. 7999; it found that CellList cells can hold 8 x the mean)."""
import os

import numpy as np
import pytest

from inputs import lj_lattice
from test_gpu_fuzz import _models, _pools

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("PMC_SEQ_FUZZ_CASES", "48"))
FIRST = int(os.environ.get("PMC_SEQ_FUZZ_FIRST", "0"))


def _draw(pmc, O, case):
    for attempt in range(50):
        rng = np.random.default_rng(77000 + case + 100000 * attempt)
        d = int(rng.choice([2, 3], p=[0.3, 0.7]))
        lt = str(rng.choice(["linked", "cell", "empty"], p=[0.6, 0.25, 0.15]))
        if lt == "empty":
            n = int(rng.integers(10, 36)) if d == 2 else int(rng.integers(5, 11))
        else:
            n = int(rng.integers(16, 90)) if d == 2 else int(rng.integers(6, 23))
        rho = float(np.exp(rng.uniform(np.log(0.05), np.log(1.3)))) if rng.random() < 0.4 else float(rng.uniform(0.6, 1.3))
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
        olt = {"linked": O.LINKEDLIST, "cell": O.CELLLIST, "empty": O.EMPTYLIST}[lt]
        try:
            O.System(pos, sp, rho, T, mo, olt)
        except ValueError as e:
            assert "bad arguments" in str(e), e
            continue
        return rng, d, S, pos, sp, rho, T, seed, chain, mm, mo, lt, olt
    raise AssertionError(f"case {case}: no acceptable geometry in 50 draws")


def _engine2_switches(case, monkeypatch):
    """The launch forms and conflict rules of the speculative windows (round 5), drawn per case from a generator of their own:
    a round as one launch / four launches, replayed from a graph or launch by launch, one or four wavefronts per evaluated step, the
    distance rule or the cell rule for conflicts -- the same chain, bit for bit, whichever runs."""
    r = np.random.default_rng(991000 + case)
    for var in ("PMC_SEQ_CELLRULE", "PMC_SEQ_NOFUSE", "PMC_SEQ_EW", "PMC_SEQ_NOGRAPH"):
        monkeypatch.delenv(var, raising=False)
    if r.random() < 0.25:
        monkeypatch.setenv("PMC_SEQ_CELLRULE", "1")
    form = r.random()
    if form < 0.2:
        monkeypatch.setenv("PMC_SEQ_NOFUSE", "1")
    elif form < 0.35:
        monkeypatch.setenv("PMC_SEQ_NOGRAPH", "1")
    if r.random() < 0.2:
        monkeypatch.setenv("PMC_SEQ_EW", "1")


@pytest.mark.parametrize("case", range(FIRST, FIRST + NCASES))
def test_random_sequential_chain(gpu, case, monkeypatch):
    _run_case(gpu, case, monkeypatch)


@pytest.mark.parametrize("case", [283, 1321])
def test_cell_list_rows_grow_with_the_fullest_cell(gpu, case, monkeypatch):
    """What the soak found: CellList's cell count, floor(Int, box / cell) with cell = box / floor(box / rcut)
    (src/neighbours.jl:126-133), can come out one short -- box / (box / 7) = 6.999... gives 6 cells of width box / 7 -- and the
    seventh slab of every dimension folds onto the first: cells of 2, 4 and 8 x the mean (64 where the mean is 8, 216 where it
    is 32).  The reference's vectors grow, and so do the device rows."""
    _run_case(gpu, case, monkeypatch)


def _run_case(gpu, case, monkeypatch):
    from oracle import oracle as O
    pmc = gpu
    rng, d, S, pos, sp, rho, T, seed, chain, mm, mo, lt, olt = _draw(pmc, O, case)
    N = len(pos)
    glt = {"linked": pmc.LinkedList, "cell": pmc.CellList, "empty": pmc.EmptyList}[lt]
    monkeypatch.delenv("PMC_SEQ_WAVES", raising=False)
    monkeypatch.delenv("PMC_SEQ_WINDOW", raising=False)
    _engine2_switches(case, monkeypatch)
    if rng.random() < 0.5:
        monkeypatch.setenv("PMC_SEQ_WAVES", str(int(rng.choice([1, 4]))))
    g = pmc.System(pos, sp, rho, T, mm, list_type=glt, seed=seed, chain=chain)
    o = O.System(pos, sp, rho, T, mo, olt)
    assert g.energy[0] == o.energy
    pm, po = _pools(pmc, rng, S, g)
    if rng.random() < 0.5:                                  # wide trial moves: many cell crossings, accepted or not
        pm[0].parameters["sigma"] = po[0]["sigma"] = float(rng.uniform(0.2, 0.7))
    pool_o = O.make_pool(po)
    done = 0
    ncalls = int(rng.integers(1, 4))
    for call in range(ncalls):
        engine = int(rng.choice([0, 1, 2, 3]))
        window = int(rng.choice([0, 0, 64, 128, 512]))
        nsw = int(rng.integers(1, 3))
        if engine == 1 and N * nsw > 6000:
            nsw, engine = 1, (0 if N > 6000 else 1)          # the literal loop runs ~10^4 steps a second
        if engine == 2 and lt == "empty":
            engine = 0                                       # the windows need cells
        try:
            g.set_sequential_engine(engine, window)
            st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        except pmc.PMCError as e:
            # the only refusal there is: a chain that was TOLD to live in LDS and does not fit (auto never refuses)
            assert engine == 3 and e.code == -4, (engine, e)      # PMC_ERR_UNSUPPORTED
            g.set_sequential_engine(0, window)
            st = pmc.mc_sweep(g, pm, nsweeps=nsw, mode="sequential")
        att, acc = o.mc_steps(pool_o, seed, chain, done, nsw * N)
        done += nsw * N
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc), (call, engine, st.accepted, acc)
        assert np.array_equal(g.position, o.positions()), (call, engine)
        assert np.array_equal(g.species, o.species()), (call, engine)
        assert g.energy[0] == o.energy, (call, engine, g.energy[0], o.energy)
        if lt != "empty":
            assert np.array_equal(g.get_cells(), o.cells()), (call, engine)
        if lt == "linked":
            hg, lg = g.get_linked_list()
            ho, lo = o.linked_list()
            assert np.array_equal(hg, ho) and np.array_equal(lg, lo), (call, engine)
        if lt == "cell" and call == ncalls - 1:
            for c in range(1, int(np.prod(o.cell_geometry()[1])) + 1):
                assert np.array_equal(g.get_cell_members(c), o.cell_members(c)), (engine, c)
        for s_ in range(1, S + 1):
            assert np.array_equal(g.species_ids(s_), o.species_ids(s_)), (call, engine, s_)


@pytest.mark.parametrize("case", range(FIRST, FIRST + max(12, NCASES // 4)))
def test_random_sequential_chains_in_shared_launches(gpu, case, monkeypatch):
    """pmc_sweep_many(PMC_SEQUENTIAL) on drawn systems: 2-12 chains of one system (own start configuration, temperature and
    chain number each), engine and window drawn per call and set on every handle, every chain against the oracle's chain."""
    from oracle import oracle as O
    pmc = gpu
    rng, d, S, pos, sp, rho, T, seed, chain, mm, mo, lt, olt = _draw(pmc, O, 500000 + case)
    N = len(pos)
    glt = {"linked": pmc.LinkedList, "cell": pmc.CellList, "empty": pmc.EmptyList}[lt]
    monkeypatch.delenv("PMC_SEQ_WAVES", raising=False)
    monkeypatch.delenv("PMC_SEQ_WINDOW", raising=False)
    _engine2_switches(500000 + case, monkeypatch)
    if rng.random() < 0.5:
        monkeypatch.setenv("PMC_SEQ_WAVES", str(int(rng.choice([1, 4]))))
    nch = int(rng.integers(2, 13)) if N <= 4000 else int(rng.integers(2, 5))
    box = (N / rho) ** (1.0 / d)
    starts, temps = [], []
    for k in range(nch):
        starts.append(pos + rng.uniform(-0.02, 0.02, size=pos.shape) * box / N ** (1.0 / d))
        temps.append(float(T * rng.uniform(0.7, 1.4)))
    chains = [pmc.System(starts[k], sp, rho, temps[k], mm, list_type=glt, seed=seed, chain=k) for k in range(nch)]
    oracles = [O.System(starts[k], sp, rho, temps[k], mo, olt) for k in range(nch)]
    pm, po = _pools(pmc, rng, S, chains[0])
    pool_o = O.make_pool(po)
    done = 0
    for call in range(int(rng.integers(1, 3))):
        engine = int(rng.choice([0, 2, 3]))
        if engine == 2 and lt == "empty":
            engine = 0
        window = int(rng.choice([0, 0, 64, 256]))
        nsw = int(rng.integers(1, 3))
        try:
            for g in chains:
                g.set_sequential_engine(engine, window)
            stats = pmc.mc_sweep_many(chains, pm, nsweeps=nsw, mode="sequential")
        except pmc.PMCError as e:
            assert engine == 3 and e.code == -4, (engine, e)
            for g in chains:
                g.set_sequential_engine(0, window)
            stats = pmc.mc_sweep_many(chains, pm, nsweeps=nsw, mode="sequential")
        for k in range(nch):
            g, o = chains[k], oracles[k]
            att, acc = o.mc_steps(pool_o, seed, k, done, nsw * N)
            assert np.array_equal(stats[k].attempted[: len(pm)], att) and np.array_equal(stats[k].accepted[: len(pm)], acc), (call, engine, k)
            assert np.array_equal(g.position, o.positions()) and np.array_equal(g.species, o.species()), (call, engine, k)
            assert g.energy[0] == o.energy, (call, engine, k)
            if lt != "empty":
                assert np.array_equal(g.get_cells(), o.cells()), (call, engine, k)
            if lt == "linked":
                hg, lg = g.get_linked_list()
                ho, lo = o.linked_list()
                assert np.array_equal(hg, ho) and np.array_equal(lg, lo), (call, engine, k)
            for s_ in range(1, S + 1):
                assert np.array_equal(g.species_ids(s_), o.species_ids(s_)), (call, engine, k, s_)
        done += nsw * N
