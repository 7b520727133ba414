"""-m gpu: seeded random configurations (dimension, density, species, model matrix, pool, temperature,
seed/chain, team width, filter/ordering switches) through both engines, bit for bit against the oracle.

The hand-picked cases of test_gpu_parity.py / test_gpu_edge.py pin named paths; this file sweeps the
combinations nobody named. Every draw is reproducible from its case number."""
import os

import numpy as np
import pytest

from inputs import lj_lattice

pytestmark = pytest.mark.gpu

NCASES = 36


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


@pytest.mark.parametrize("case", range(NCASES))
def test_random_configuration_both_engines(gpu, case, monkeypatch):
    from oracle import oracle as O
    pmc = gpu
    rng = np.random.default_rng(1000 + case)
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
    for var, p in (("PMC_TEAM", 0.6), ("PMC_NO_FILTER", 0.25), ("PMC_NO_ORDER", 0.25)):
        monkeypatch.delenv(var, raising=False)
        if rng.random() < p:
            monkeypatch.setenv(var, str(int(rng.choice([1, 2, 4]))) if var == "PMC_TEAM" else "1")
    mm, mo = _models(pmc, O, rng, S)
    try:
        g = pmc.System(pos, sp, rho, T, mm, seed=seed, chain=chain)
    except (ValueError, RuntimeError) as e:                # geometry the reference itself rejects
        pytest.skip(f"case {case}: {e}")
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
