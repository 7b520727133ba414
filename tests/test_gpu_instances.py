"""-m gpu: EVERY instance of the sweep kernel the host can dispatch -- dimension x model kind x {one, several species} x
team width (and both register budgets of the one-wavefront kernels) x {single-Displacement pool, multi-move pool} x {no images, images} -- against the oracle, three sweeps each
(m = 2, so a pass of two rounds and a pass of one).  The sweep kernel is a template with several hundred instances, many of
which keep a few dozen bytes of scratch per lane; twice in round 2 and twice in round 3 ONE instance computed wrong energies
after a source change that left every other instance right (a cross-lane read of an inactive lane, harmless
until the operand is spilled -- fixed, and guarded at source level by tests/test_source_lint.py).  The size-driven tests reach
only the instances their systems happen to select; this one reaches all of them."""
import itertools

import numpy as np
import pytest

from inputs import lj_lattice

pytestmark = pytest.mark.gpu

KINDS = ["lj", "soft_int", "soft_odd", "smooth", "mixed"]


def _matrix(pmc, O, kind, S, rcut_over_sigma=2.3):
    arr = (O.Model * (S * S))()
    mm = [[None] * S for _ in range(S)]
    for i in range(S):
        for j in range(i, S):
            eps, sig = 1.0 + 0.2 * (i != j), 1.0 - 0.06 * (i + j)
            rcut = rcut_over_sigma * sig
            k = kind if kind != "mixed" else ["lj", "soft_int", "smooth"][(i + j) % 3]
            if k == "lj":
                p, o = pmc.LennardJones(eps, sig, rcut=rcut), O.lennard_jones(eps, sig, rcut)
            elif k == "soft_int":
                p, o = pmc.SoftSpheres(eps, sig, 12, rcut=rcut), O.soft_spheres(eps, sig, 12, rcut)
            elif k == "soft_odd":
                p, o = pmc.SoftSpheres(eps, sig, 9, rcut=rcut), O.soft_spheres(eps, sig, 9, rcut)
            else:
                p, o = pmc.SmoothLennardJones(eps, sig, rcut=rcut), O.smooth_lennard_jones(eps, sig, rcut)
            mm[i][j] = mm[j][i] = p
            arr[i * S + j] = arr[j * S + i] = o
    return mm, (arr, S)


@pytest.mark.parametrize("d", [3, 2])
@pytest.mark.parametrize("kind", KINDS)
def test_every_kernel_instance_against_the_oracle(gpu, monkeypatch, d, kind):
    from oracle import oracle as O
    pmc = gpu
    n = 12 if d == 3 else 40
    pos0, sp1, rho = lj_lattice(n, jitter=0.06, seed=11, rho=0.9, d=d)
    N = len(pos0)
    L = (N / rho) ** (1.0 / d)
    rng = np.random.default_rng(5)
    sp2 = np.where(rng.random(N) < 0.4, 2, 1).astype(np.int32)
    shift = rng.integers(-2, 3, size=pos0.shape) * L
    checked = 0
    # team "1/3": one wavefront per cell in its 168-VGPR build (the host takes it when LDS leaves room for <= 12-14 workgroups
    # per CU, i.e. for large dense systems; forced here through PMC_MINWAVES)
    for S, team, multi_pool, images in itertools.product((1, 2), ("1", "1/3", "2", "4", "8"), (False, True), (False, True)):
        if kind == "mixed" and S == 1:
            continue                                   # one species has one model: no mixed kinds
        if team == "1/3" and images:
            continue                                   # (the image instances exist in one register budget per team width)
        monkeypatch.setenv("PMC_TEAM", team.split("/")[0])
        if team == "1/3":
            monkeypatch.setenv("PMC_MINWAVES", "3")
        else:
            monkeypatch.delenv("PMC_MINWAVES", raising=False)
        sp = sp1 if S == 1 else sp2
        pos = pos0 + shift if images else pos0
        mm, mo = _matrix(pmc, O, kind, S)
        g = pmc.System(pos, sp, rho, 1.1, mm, seed=77)
        g.rounds = 2
        pm = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.07}, 1.0)]
        po = [{"kind": "Displacement", "sigma": 0.07, "probability": 1.0}]
        if multi_pool:
            pm.append(pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.3}, 0.3))
            po.append({"kind": "Displacement", "sigma": 0.3, "probability": 0.3})
            if S > 1:
                pm.append(pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.4))
                po.append({"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.4})
        cb = O.Checkerboard(pos, sp, np.asarray(g.box), 1.1, mo, seed=77)
        st = pmc.mc_sweep(g, pm, nsweeps=3)
        att, acc = cb.sweeps(O.make_pool(po), 0, 3, rounds=2)
        tag = (d, kind, S, team, multi_pool, images)
        assert list(st.attempted[: len(pm)]) == list(att) and list(st.accepted[: len(pm)]) == list(acc), tag
        assert st.energy_fix == cb.energy_fix, tag
        assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species()), tag
        assert acc[0] > N // 4, tag
        checked += 1
    assert checked == (36 if kind != "mixed" else 18)


@pytest.mark.parametrize("d", [3, 2])
@pytest.mark.parametrize("kind", KINDS)
def test_every_big_neighbourhood_instance_against_the_oracle(gpu, monkeypatch, d, kind):
    """The big-neighbourhood instances (k_cb_pass<..., BIG>, round 5): boxes of four cutoffs per side at liquid density have
    3^d-cell neighbourhoods beyond the 1024 candidates of the default staging and the 512 of the pre-filter's register copy; a
    team of eight then stages the whole neighbourhood and streams ONE float copy of it in LDS (the reference's own validation
    example is such a system: examples/lj-mixture/run-validation.py:26-34, rcut = 4 sigma).  Dimension x model kind x {one,
    two species} x {single-Displacement pool, multi-move pool with swaps}, three sweeps at m = 2 against the oracle, and the same
    bits with the instances switched off (PMC_NO_BIG=1: the direct FP64 loop these systems took until round 4)."""
    from oracle import oracle as O
    pmc = gpu
    monkeypatch.delenv("PMC_TEAM", raising=False)
    monkeypatch.delenv("PMC_MINWAVES", raising=False)
    if d == 3:
        n, rho, rc = 14, 0.622, 4.0          # box 16.4 = 4 cells of 4.1; 43 particles per cell, 1158 per neighbourhood
    else:
        n, rho, rc = 45, 0.84, 12.0          # box 49.1 = 4 cells of 12.3; 127 particles per cell, 1139 per neighbourhood
    pos, sp1, rho = lj_lattice(n, jitter=0.06, seed=11, rho=rho, d=d)
    N = len(pos)
    rng = np.random.default_rng(5)
    sp2 = np.where(rng.random(N) < 0.4, 2, 1).astype(np.int32)
    checked = 0
    for S, multi_pool in itertools.product((1, 2), (False, True)):
        if kind == "mixed" and S == 1:
            continue
        sp = sp1 if S == 1 else sp2
        mm, mo = _matrix(pmc, O, kind, S, rcut_over_sigma=rc)
        pm = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.07}, 1.0)]
        po = [{"kind": "Displacement", "sigma": 0.07, "probability": 1.0}]
        results = []
        for no_big in (False, True):
            if no_big:
                monkeypatch.setenv("PMC_NO_BIG", "1")
            else:
                monkeypatch.delenv("PMC_NO_BIG", raising=False)
            g = pmc.System(pos, sp, rho, 1.1, mm, seed=77)
            assert tuple(int(v) for v in g.grid()[0][:d]) == (4,) * d
            g.rounds = 2
            pmg = list(pm)
            if multi_pool:
                pmg.append(pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.3}, 0.3))
                if S > 1:
                    pmg.append(pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.4))
            st = pmc.mc_sweep(g, pmg, nsweeps=3)
            results.append((st, g.position.copy(), g.species.copy()))
        pof = list(po)
        if multi_pool:
            pof.append({"kind": "Displacement", "sigma": 0.3, "probability": 0.3})
            if S > 1:
                pof.append({"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.4})
        cb = O.Checkerboard(pos, sp, np.asarray(g.box), 1.1, mo, seed=77)
        att, acc = cb.sweeps(O.make_pool(pof), 0, 3, rounds=2)
        tag = (d, kind, S, multi_pool)
        for st, p, spc in results:
            assert list(st.attempted[: len(pof)]) == list(att) and list(st.accepted[: len(pof)]) == list(acc), tag
            assert st.energy_fix == cb.energy_fix, tag
            assert np.array_equal(p, cb.positions()) and np.array_equal(spc, cb.species()), tag
        big, direct = results[0][0], results[1][0]
        # the big instance ran its streamed pre-filter (few FP64 distance evaluations per FP32 test); the default one tested
        # every candidate of every slot in FP64 and does not count them
        assert 0 < big.dense_evals < 0.5 * big.pair_tests, (tag, big.dense_evals, big.pair_tests)
        assert direct.dense_evals == 0, tag
        assert acc[0] > N // 4, tag
        checked += 1
    assert checked == (4 if kind != "mixed" else 2)
