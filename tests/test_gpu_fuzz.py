"""-m gpu: seeded random configurations (dimension, density, species, model matrix, pool, temperature,
seed/chain, team width, filter/ordering switches) through both engines, bit for bit against the oracle.

The hand-picked cases of test_gpu_parity.py / test_gpu_edge.py pin named paths; this file sweeps the
combinations nobody named. Every draw is reproducible from its case number."""
import os

import numpy as np
import pytest

from inputs import lj_lattice

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("PMC_FUZZ_CASES", "36"))      # PMC_FUZZ_CASES / PMC_FUZZ_FIRST: longer soaks (round 4: cases 0 .. 1499)
FIRST = int(os.environ.get("PMC_FUZZ_FIRST", "0"))


def _models(pmc, O, rng, S):
    """A random symmetric S x S model matrix on both sides (host mirror, oracle)."""
    kind = rng.choice(["lj", "lj_noshift", "soft_int", "soft_odd", "smooth", "mixed"])
    arr = (O.Model * (S * S))()
    mm = [[None] * S for _ in range(S)]
    for i in range(S):
        for j in range(i, S):
            eps = float(rng.uniform(0.5, 1.5))
            sig = float(rng.uniform(0.8, 1.0))
            rcut = float(sig * rng.uniform(2.0, 2.6))
            k = kind if kind != "mixed" else rng.choice(["lj", "soft_int", "smooth"])
            if k == "lj":
                p, o = pmc.LennardJones(eps, sig, rcut=rcut), O.lennard_jones(eps, sig, rcut)
            elif k == "lj_noshift":
                p, o = pmc.LennardJones(eps, sig, rcut=rcut, shift_potential=False), O.lennard_jones(eps, sig, rcut, False)
            elif k == "soft_int":
                n = int(rng.choice([6, 12, 18]))
                p, o = pmc.SoftSpheres(eps, sig, n, rcut=rcut), O.soft_spheres(eps, sig, n, rcut)
            elif k == "soft_odd":
                n = int(rng.choice([7, 9, 13]))
                p, o = pmc.SoftSpheres(eps, sig, n, rcut=rcut), O.soft_spheres(eps, sig, n, rcut)
            else:
                p, o = pmc.SmoothLennardJones(eps, sig, rcut=rcut), O.smooth_lennard_jones(eps, sig, rcut)
            mm[i][j] = mm[j][i] = p
            arr[i * S + j] = o
            arr[j * S + i] = o
    return mm, (arr, S)


def _pools(pmc, rng, S, g):
    """Pool on both sides: 1-2 displacements, and swaps when there is more than one species."""
    pm = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": float(rng.uniform(0.03, 0.12))}, 1.0)]
    po = [{"kind": "Displacement", "sigma": pm[0].parameters["sigma"], "probability": 1.0}]
    if rng.random() < 0.4:
        s2 = float(rng.uniform(0.2, 0.5))
        pm.append(pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": s2}, 0.3))
        po.append({"kind": "Displacement", "sigma": s2, "probability": 0.3})
    if S > 1 and rng.random() < 0.75:
        a, b = sorted(int(x) for x in rng.choice(np.arange(1, S + 1), size=2, replace=False))
        w = float(rng.uniform(0.1, 0.4))
        pm.append(pmc.Move(pmc.DiscreteSwap([a, b], g), pmc.DoubleUniform(), [], w))
        po.append({"kind": "DiscreteSwap", "species": (a, b), "probability": w})
    tot = sum(m.probability for m in pm)
    for m, o in zip(pm, po):
        m.probability /= tot
        o["probability"] = m.probability
    return pm, po


def _oracle_accepts(O, pos, sp, rho, T, mo):
    """The reference's LinkedList constructor rejects boxes where Int(box / cell) is inexact
    (src/neighbours.jl:238); the oracle restates that as ValueError("bad arguments")."""
    try:
        O.System(pos, sp, rho, T, mo, O.LINKEDLIST)
        return True
    except ValueError as e:
        assert "bad arguments" in str(e), e
        return False


def _draw_configuration(pmc, O, case):
    """Case `case`, attempt 0, 1, ...: the first draw whose geometry the reference accepts.  No case skips;
    the rejected geometries have their own test below."""
    for attempt in range(50):
        rng = np.random.default_rng(1000 + case + 100000 * attempt)
        d = int(rng.choice([2, 3], p=[0.3, 0.7]))
        n = int(rng.integers(24, 60)) if d == 2 else int(rng.integers(11, 19))
        rho = float(rng.uniform(0.35, 1.25))
        S = int(rng.choice([1, 2, 3]))
        pos, sp, _ = lj_lattice(n, jitter=0.08 * rho ** (-1.0 / d), seed=case, rho=rho, d=d)
        if S > 1:
            sp = rng.integers(1, S + 1, size=len(sp)).astype(np.int32)
            sp[:S] = np.arange(1, S + 1)                       # every species present
        pos = pos + rng.integers(-1, 2, size=pos.shape) * (len(pos) / rho) ** (1.0 / d)   # unwrapped images
        T = float(rng.uniform(0.5, 2.0))
        seed, chain = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 8))
        env = {}
        for var, p in (("PMC_TEAM", 0.6), ("PMC_NO_FILTER", 0.25), ("PMC_NO_ORDER", 0.25), ("PMC_TICKETS", 0.25), ("PMC_CAP", 0.25)):
            if rng.random() < p:
                # (PMC_CAP: a staging capacity below most neighbourhoods -- the overflow paths; PMC_TICKETS=1: one ticket counter)
                env[var] = (str(int(rng.choice([1, 2, 4, 8]))) if var == "PMC_TEAM" else
                            str(int(rng.choice([128, 192, 256]))) if var == "PMC_CAP" else "1")
        mm, mo = _models(pmc, O, rng, S)
        if _oracle_accepts(O, pos, sp, rho, T, mo):
            return rng, d, S, pos, sp, rho, T, seed, chain, env, mm, mo
    raise AssertionError(f"case {case}: no acceptable geometry in 50 draws")


@pytest.mark.parametrize("case", range(FIRST, FIRST + NCASES))
def test_random_configuration_both_engines(gpu, case, monkeypatch):
    from oracle import oracle as O
    pmc = gpu
    rng, d, S, pos, sp, rho, T, seed, chain, env, mm, mo = _draw_configuration(pmc, O, case)
    for var in ("PMC_TEAM", "PMC_NO_FILTER", "PMC_NO_ORDER", "PMC_TICKETS", "PMC_CAP"):
        monkeypatch.delenv(var, raising=False)
    for var, val in env.items():
        monkeypatch.setenv(var, val)
    g = pmc.System(pos, sp, rho, T, mm, seed=seed, chain=chain)      # any exception here is a failure, not a skip
    pm, po = _pools(pmc, rng, S, g)
    pool_o = O.make_pool(po)

    # --- throughput engine against the restated schedule -----------------------------------------
    o = O.System(pos, sp, rho, T, mo, O.LINKEDLIST)
    assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles()) and g.energy[0] == o.energy
    cb = O.Checkerboard(pos, sp, o.box, T, mo, seed=seed, chain=chain)
    assert g.total_energy()[1] == cb.energy_fix
    done = 0
    for nsw in (1, 2):
        st = pmc.mc_sweep(g, pm, nsweeps=nsw)
        att, acc = cb.sweeps(pool_o, done, nsw)
        done += nsw
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc)
        assert att.sum() == nsw * len(pos)
        assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())
        assert st.energy_fix == cb.energy_fix
    assert g.total_energy()[1] == cb.energy_fix == cb.total_energy_fix()

    # --- reference-order engine continues from the same state -------------------------------------
    if len(pos) <= 3000:
        o2 = O.System(g.position, g.species, rho, T, mo, O.LINKEDLIST)
        g.energy[0] = o2.energy                            # hand-over: reference-order sum of the current state
        for s_ in range(1, S + 1):                         # ... and SpeciesList(species) of the current state
            assert np.array_equal(g.species_ids(s_), o2.species_ids(s_))
        st = pmc.mc_sweep(g, pm, nsweeps=1, mode="sequential")
        att, acc = o2.mc_steps(pool_o, seed, chain, 0, len(pos))
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc)
        assert np.array_equal(g.position, o2.positions()) and np.array_equal(g.species, o2.species())
        assert g.energy[0] == o2.energy
        assert np.array_equal(g.get_cells(), o2.cells())
        for s_ in range(1, S + 1):
            assert np.array_equal(g.species_ids(s_), o2.species_ids(s_))


@pytest.mark.parametrize("case", range(FIRST, FIRST + max(12, NCASES // 3)))
def test_random_operation_sequences(gpu, case):
    """Random interleavings of throughput sweeps, reference-order sweeps, per-move verbs, temperature changes and
    position/species round trips on ONE handle; after every operation the device state equals the oracle's."""
    from oracle import oracle as O
    pmc = gpu
    for attempt in range(50):
        rng = np.random.default_rng(5000 + case + 100000 * attempt)
        d = int(rng.choice([2, 3]))
        n = int(rng.integers(26, 40)) if d == 2 else int(rng.integers(11, 14))
        rho = float(rng.uniform(0.6, 1.2))
        S = int(rng.choice([1, 2, 3]))
        pos, sp, _ = lj_lattice(n, jitter=0.08 * rho ** (-1.0 / d), seed=100 + case, rho=rho, d=d)
        N = len(pos)
        if S > 1:
            sp = rng.integers(1, S + 1, size=N).astype(np.int32)
            sp[:S] = np.arange(1, S + 1)
        T = float(rng.uniform(0.6, 1.8))
        seed, chain = int(rng.integers(0, 2 ** 50)), int(rng.integers(0, 4))
        mm, mo = _models(pmc, O, rng, S)
        if _oracle_accepts(O, pos, sp, rho, T, mo):
            break
    else:
        raise AssertionError(f"case {case}: no acceptable geometry in 50 draws")
    g = pmc.System(pos, sp, rho, T, mm, seed=seed, chain=chain)      # any exception here is a failure, not a skip
    pm, po = _pools(pmc, rng, S, g)
    pool_o = O.make_pool(po)
    o = O.System(pos, sp, rho, T, mo, O.LINKEDLIST)
    assert g.energy[0] == o.energy

    def same_state():
        assert np.array_equal(g.position, o.positions()) and np.array_equal(g.species, o.species())
        assert g.energy[0] == o.energy
        assert np.array_equal(g.get_cells(), o.cells())
        hg, lg = g.get_linked_list()
        ho, lo = o.linked_list()
        assert np.array_equal(hg, ho) and np.array_equal(lg, lo)
        for s_ in range(1, S + 1):
            assert np.array_equal(g.species_ids(s_), o.species_ids(s_))

    # The throughput engine derives every particle's periodic image when it takes the state over -- after an upload, per-move
    # verbs or a reference-order sweep -- and keeps it while its own state stays valid (DESIGN.md 2.2 "images"); the oracle's
    # Checkerboard fixes the images at its creation, so it lives exactly as long: across consecutive "cb" steps.
    cb = None
    for step in range(10):
        op = rng.choice(["cb", "seq", "verbs", "temp", "roundtrip"], p=[0.3, 0.25, 0.25, 0.1, 0.1])
        if op == "cb":
            k = int(rng.integers(1, 3))
            _, sweep0 = g.get_counters()
            if cb is None:
                cb = O.Checkerboard(o.positions(), o.species(), o.box, T, mo, seed=seed, chain=chain)
            st = pmc.mc_sweep(g, pm, nsweeps=k)
            att, acc = cb.sweeps(pool_o, sweep0, k)
            assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc)
            assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())
            assert st.energy_fix == cb.energy_fix == cb.total_energy_fix()
            # hand-over to the reference-order side: a fresh System of the current state
            o = O.System(cb.positions(), cb.species(), rho, T, mo, O.LINKEDLIST)
            g.energy[0] = o.energy
            same_state()
        elif op == "seq":
            cb = None
            step0, _ = g.get_counters()
            st = pmc.mc_sweep(g, pm, nsweeps=1, mode="sequential")
            att, acc = o.mc_steps(pool_o, seed, chain, step0, N)
            assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc)
            same_state()
        elif op == "verbs":
            cb = None
            for _ in range(6):
                if S > 1 and rng.random() < 0.5:
                    a, b = (int(x) for x in rng.choice(np.arange(1, S + 1), size=2, replace=False))
                    ia, ib = int(rng.choice(g.species_ids(a))), int(rng.choice(g.species_ids(b)))
                    act = pmc.DiscreteSwap(ia, ib, (a, b), (1, 1), 0.0)
                    e = pmc.perform_action(g, act)
                    r = o.swap_perform(ia, ib, a, b)
                    assert (e[0], e[1], act.de) == r
                    if rng.random() < 0.5:
                        pmc.invert_action(act, g)
                        pmc.revert_action(g, act)
                        o.swap_revert(ib, ia, a, b, r[2])
                else:
                    i = int(rng.integers(1, N + 1))
                    delta = rng.normal(0, 0.3, d)
                    act = pmc.Displacement(i, delta, 0.0)
                    e = pmc.perform_action(g, act)
                    r = o.displacement_perform(i, delta)
                    assert (e[0], e[1], act.de) == r
                    if rng.random() < 0.5:
                        pmc.invert_action(act, g)
                        pmc.revert_action(g, act)
                        o.displacement_revert(i, -delta, r[2])
            same_state()
        elif op == "temp":
            T = float(rng.uniform(0.6, 1.8))
            g.temperature = T
            o.set_temperature(T)
            if cb is not None:
                cb.set_temperature(T)
        else:
            cb = None
            p_now, s_now = g.position, g.species           # download ... upload: a fresh list on both sides
            g.upload(p_now, s_now)
            o = O.System(p_now, s_now, rho, T, mo, O.LINKEDLIST)
            g.build_neighbour_list()
            g.energy[0] = o.energy
            same_state()


@pytest.mark.parametrize("case", range(6))
def test_random_dense_configurations(gpu, case, monkeypatch):
    """Long cutoffs (3.4-4.2 sigma) at liquid density: 50-100 particles per cell (several proposal chunks per
    cell), 1500-2500 candidates per neighbourhood (beyond the LDS capacity), 4-5 cells per side (wrapped
    neighbourhoods everywhere), with every team width."""
    from oracle import oracle as O
    pmc = gpu
    rng = np.random.default_rng(9000 + case)
    n = int(rng.integers(17, 21))
    rho = float(rng.uniform(0.9, 1.15))
    S = int(rng.choice([1, 2]))
    pos, sp, _ = lj_lattice(n, jitter=0.07, seed=300 + case, rho=rho)
    N = len(pos)
    if S > 1:
        sp = rng.integers(1, 3, size=N).astype(np.int32)
    L = (N / rho) ** (1.0 / 3.0)
    rcut = float(min(rng.uniform(3.4, 4.2), L / 4.0 * 0.999))
    T = float(rng.uniform(0.8, 1.6))
    seed = int(rng.integers(0, 2 ** 60))
    monkeypatch.setenv("PMC_TEAM", str([1, 2, 4, 8, 8, 4][case % 6]))
    arr = (O.Model * (S * S))()
    mm = [[None] * S for _ in range(S)]
    for i in range(S):
        for j in range(i, S):
            eps, sig = float(rng.uniform(0.7, 1.3)), float(rng.uniform(0.9, 1.0))
            if case % 2:
                p, o_ = pmc.SmoothLennardJones(eps, sig, rcut=rcut), O.smooth_lennard_jones(eps, sig, rcut)
            else:
                p, o_ = pmc.LennardJones(eps, sig, rcut=rcut), O.lennard_jones(eps, sig, rcut)
            mm[i][j] = mm[j][i] = p
            arr[i * S + j] = arr[j * S + i] = o_
    mo = (arr, S)
    g = pmc.System(pos, sp, rho, T, mm, seed=seed)
    pm, po = _pools(pmc, rng, S, g)
    o = O.System(pos, sp, rho, T, mo, O.LINKEDLIST)
    assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles()) and g.energy[0] == o.energy
    cb = O.Checkerboard(pos, sp, o.box, T, mo, seed=seed)
    assert max(g.grid()[0]) <= 5
    st = pmc.mc_sweep(g, pm, nsweeps=2)
    att, acc = cb.sweeps(O.make_pool(po), 0, 2)
    assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc)
    assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())
    assert st.energy_fix == cb.energy_fix == cb.total_energy_fix() == g.total_energy()[1]


def test_geometry_the_reference_rejects_is_rejected_with_badarg(gpu):
    """Int(box / cell) InexactError of the reference (src/neighbours.jl:238): the oracle says "bad arguments",
    the library PMC_ERR_BADARG with the geometry message -- and nothing else is ever swallowed by this file."""
    from oracle import oracle as O
    from particlesmc_amd import _lib
    pmc = gpu
    found = 0
    for k in range(400):
        rng = np.random.default_rng(77000 + k)
        n = int(rng.integers(11, 19))
        rho = float(rng.uniform(0.35, 1.25))
        pos, sp, _ = lj_lattice(n, jitter=0.05, seed=k, rho=rho, d=3)
        rcut = float(rng.uniform(1.6, 2.6))
        mo = O.single(O.lennard_jones(1.0, 1.0, rcut))
        if _oracle_accepts(O, pos, sp, rho, 1.0, mo):
            continue
        with pytest.raises(pmc.PMCError) as ei:
            pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0, rcut=rcut))
        assert ei.value.code == _lib.PMC_ERR_BADARG and "box / cell is not a positive integer" in str(ei.value)
        found += 1
        if found == 3:
            break
    assert found >= 1, "no rejected geometry among 400 draws: the scan no longer exercises the rejection"
