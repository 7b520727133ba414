"""-m gpu: seeded fuzz of the throughput engine on SMALL boxes (round 5) -- 2 ... 6 cutoffs per side, up to ~1000 particles per
neighbourhood: the single-cell boxes (one cell spanning the periodic box), the big-neighbourhood instances of k_cb_pass (4-5 cells
per side at liquid density) and the ordinary grid next to them, drawn together: dimension, size, density, 1-3 species with random
model matrices of every kind, pools with one or two displacement widths and swaps, forced team widths, the schedule parameter m,
positions shifted by whole boxes, calls that end in a short pass, several chains per shared launch.  Everything the schedule
defines -- accept counts per move, positions, species, the exact energy -- against the oracle after every call.
PMC_SMALLBOX_FUZZ_CASES / PMC_SMALLBOX_FUZZ_FIRST run more than the 32 cases of the suite."""
import os

import numpy as np
import pytest

from test_gpu_fuzz import _pools

pytestmark = pytest.mark.gpu
NCASES = int(os.environ.get("PMC_SMALLBOX_FUZZ_CASES", "32"))
FIRST = int(os.environ.get("PMC_SMALLBOX_FUZZ_FIRST", "0"))


def _draw(pmc, O, case):
    for attempt in range(100):
        rng = np.random.default_rng(77000 + case + 100000 * attempt)
        d = int(rng.choice([2, 3], p=[0.25, 0.75]))
        S = int(rng.choice([1, 2, 3]))
        ratio = float(rng.uniform(2.05, 6.0))                       # box length in (largest) cutoffs
        n = int(rng.integers(14, 32)) if d == 2 else int(rng.integers(6, 15))
        rho = float(rng.uniform(0.3, 1.0))
        if d == 3 and ratio >= 4.0 and rng.random() < 0.6:          # four cells per side, > 1000 candidates per neighbourhood: the big instances
            n, rho = int(rng.integers(12, 15)), float(rng.uniform(0.7, 1.0))
        N = n ** d
        L = (N / rho) ** (1.0 / d)
        mm, mo = _stretched_models(pmc, O, rng, S, L / ratio)
        if mm is None:
            continue
        g = np.stack(np.meshgrid(*[np.arange(n)] * d, indexing="ij"), axis=-1).reshape(-1, d).astype(np.float64)
        pos = (g + 0.5) * (L / n) + rng.uniform(-0.08, 0.08, size=(N, d)) * rho ** (-1.0 / d)
        sp = np.ones(N, dtype=np.int32)
        if S > 1:
            sp = rng.integers(1, S + 1, size=N).astype(np.int32)
            sp[:S] = np.arange(1, S + 1)
        if rng.random() < 0.3:
            pos = pos + rng.integers(-2, 3, size=pos.shape) * L     # periodic images: the instances without a streamed filter
        T = float(rng.uniform(0.6, 2.0))
        seed, chain = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 8))
        env = {}
        if rng.random() < 0.4:
            env["PMC_TEAM"] = str(int(rng.choice([1, 2, 4, 8])))
        if rng.random() < 0.15:
            env["PMC_NO_BIG"] = "1"
        rounds = int(rng.choice([1, 1, 2, 3]))
        try:
            cb = O.Checkerboard(pos, sp, np.full(d, L), T, mo, seed=seed, chain=chain)
        except ValueError:
            continue                     # below two cutoffs, or more than 1024 particles in a single-cell box: tests/test_gpu_single_cell.py
        return rng, d, S, pos, sp, rho, T, seed, chain, env, rounds, mm, mo, cb
    raise AssertionError(f"case {case}: no acceptable geometry")


def _stretched_models(pmc, O, rng, S, maxcut):
    """A random symmetric S x S model matrix (as tests/test_gpu_fuzz.py draws them) whose cutoffs are stretched by a common factor
    so that the largest is `maxcut`; None if that would SHORTEN them (the models stay physical: rcut >= 2 sigma)."""
    kind = rng.choice(["lj", "lj_noshift", "soft_int", "smooth", "mixed"])
    draws = {}
    for i in range(S):
        for j in range(i, S):
            draws[(i, j)] = (float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.8, 1.0)), float(rng.uniform(2.0, 2.6)),
                             kind if kind != "mixed" else str(rng.choice(["lj", "soft_int", "smooth"])), int(rng.choice([6, 12])))
    scale = maxcut / max(sig * f for _, sig, f, _, _ in draws.values())
    if scale < 1.0:
        return None, None
    arr = (O.Model * (S * S))()
    mm = [[None] * S for _ in range(S)]
    for (i, j), (eps, sig, f, k, npow) in draws.items():
        rcut = sig * f * scale
        if k == "lj":
            p, o = pmc.LennardJones(eps, sig, rcut=rcut), O.lennard_jones(eps, sig, rcut)
        elif k == "lj_noshift":
            p, o = pmc.LennardJones(eps, sig, rcut=rcut, shift_potential=False), O.lennard_jones(eps, sig, rcut, False)
        elif k == "soft_int":
            p, o = pmc.SoftSpheres(eps, sig, npow, rcut=rcut), O.soft_spheres(eps, sig, npow, rcut)
        else:
            p, o = pmc.SmoothLennardJones(eps, sig, rcut=rcut), O.smooth_lennard_jones(eps, sig, rcut)
        mm[i][j] = mm[j][i] = p
        arr[i * S + j] = arr[j * S + i] = o
    return mm, (arr, S)


@pytest.mark.parametrize("case", range(FIRST, FIRST + NCASES))
def test_small_box_against_the_oracle(gpu, case, monkeypatch):
    from oracle import oracle as O
    pmc = gpu
    rng, d, S, pos, sp, rho, T, seed, chain, env, rounds, mm, mo, cb = _draw(pmc, O, case)
    for var in ("PMC_TEAM", "PMC_NO_BIG", "PMC_NO_FILTER", "PMC_CAP"):
        monkeypatch.delenv(var, raising=False)
    for var, val in env.items():
        monkeypatch.setenv(var, val)
    g = pmc.System(pos, sp, rho, T, mm, seed=seed, chain=chain)
    assert tuple(int(v) for v in g.grid()[0][:d]) == tuple(int(v) for v in cb.grid()[0][:d])
    assert g.total_energy()[1] == cb.energy_fix
    pm, po = _pools(pmc, rng, S, g)
    pool_o = O.make_pool(po)
    g.rounds = rounds
    done = 0
    for nsw in (2, 1, 3):
        st = pmc.mc_sweep(g, pm, nsweeps=nsw)
        att, acc = cb.sweeps(pool_o, done, nsw, rounds=rounds)
        done += nsw
        tag = (case, d, S, tuple(int(v) for v in g.grid()[0][:d]), env, rounds)
        assert np.array_equal(st.attempted[: len(pm)], att) and np.array_equal(st.accepted[: len(pm)], acc), tag
        assert att.sum() == nsw * len(pos), tag
        assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species()), tag
        assert st.energy_fix == cb.energy_fix, tag
    assert g.total_energy()[1] == cb.energy_fix == cb.total_energy_fix()
    # ... and as three chains of a shared launch (own keys), from the start configuration
    if case % 4 == 0:
        chains = [pmc.System(pos, sp, rho, T, mm, seed=seed, chain=chain + 1 + k) for k in range(3)]
        for c in chains:
            c.rounds = rounds
        sts = pmc.mc_sweep_many(chains, pm, nsweeps=2)
        for k, c in enumerate(chains):
            cbk = O.Checkerboard(pos, sp, np.asarray(c.box), T, mo, seed=seed, chain=chain + 1 + k)
            att, acc = cbk.sweeps(pool_o, 0, 2, rounds=rounds)
            assert np.array_equal(sts[k].accepted[: len(pm)], acc) and sts[k].energy_fix == cbk.energy_fix, (case, k)
            assert np.array_equal(c.position, cbk.positions()) and np.array_equal(c.species, cbk.species()), (case, k)
