"""-m gpu: periodic images of the throughput engine (DESIGN.md 2.2 "images").  Positions are never wrapped
(src/moves.jl:46-48); an unwrapped input -- a restart from an unwrapped frame, particles hundreds of boxes from the
origin -- is folded through a persistent integer image per particle, fixed at upload, and must give the oracle's chain bit
for bit, keep the running energy equal to a recomputation, leave every particle in its own image, and (the point of the
exercise) take the same fast path as a configuration that sits in the primary box."""
import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice

pytestmark = pytest.mark.gpu


def _disp(pmc, sigma, p=1.0):
    return pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, p)


@pytest.mark.parametrize("team", ["1", "4", "8"])
@pytest.mark.parametrize("span", [3, 400, 100000])
def test_unwrapped_inputs_give_the_oracle_chain(gpu, monkeypatch, team, span):
    """Every particle shifted by random whole boxes within +-span (beyond +-511 boxes a particle stays unfolded: the
    general minimum-image path).  10 sweeps with m = 2: positions, accept count, exact energy == oracle; running energy ==
    recomputation; nobody changed image."""
    from oracle import oracle as O
    pmc = gpu
    monkeypatch.setenv("PMC_TEAM", team)
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=5)
    L = (len(pos) / rho) ** (1.0 / 3.0)
    shift = np.random.default_rng(7).integers(-span, span + 1, size=pos.shape) * L
    p = pos + shift
    g = pmc.System(p, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=3)
    g.rounds = 2
    o = O.System(p, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    cb = O.Checkerboard(p, sp, o.box, 1.0, o.models, seed=3)
    _, fix0 = g.total_energy()
    assert fix0 == cb.energy_fix
    st = pmc.mc_sweep(g, [_disp(pmc, 0.08)], nsweeps=10)
    att, acc = cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.08}]), 0, 10, rounds=2)
    assert st.accepted[0] == acc[0] and st.energy_fix == cb.energy_fix
    out = g.position
    assert np.array_equal(out, cb.positions())
    assert st.energy_fix == g.total_energy()[1] == cb.total_energy_fix()
    assert np.abs(out - p).max() < 1.0                                    # everybody is still where it was, images included
    # the same lattice in the primary box: the folded coordinates differ from it by rounding only, so the energies agree
    # to ~1e-12 relative and the chains accept almost the same moves
    g0 = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=3)
    e0, fix_in = g0.total_energy()
    assert abs(fix0 - fix_in) <= 1e-9 * abs(fix_in)


def test_unwrapped_mixture_with_swaps_and_partial_images(gpu):
    """Kob-Andersen with 30 % swap slots; only every third particle is outside the primary box, so cells mix images."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(4000, frac_b=0.4, rho=1.2, jitter=0.05, seed=5)
    L = (len(pos) / rho) ** (1.0 / 3.0)
    rng = np.random.default_rng(2)
    shift = rng.integers(-5, 6, size=pos.shape) * L * (rng.random((len(pos), 1)) < 0.33)
    p = pos + shift
    g = pmc.System(p, sp, rho, 0.8, pmc.KobAndersen(), seed=21)
    pool = [_disp(pmc, 0.07, 0.7), pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.3)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.07, "probability": 0.7},
                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.3}])
    cb = O.Checkerboard(p, sp, np.asarray(g.box), 0.8, O.kob_andersen(), seed=21)
    st = pmc.mc_sweep(g, pool, nsweeps=6)
    att, acc = cb.sweeps(opool, 0, 6)
    assert list(st.accepted[:2]) == list(acc) and acc[1] > 20 and st.energy_fix == cb.energy_fix
    assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())
    assert st.energy_fix == g.total_energy()[1]
    # pmc_delta_e on out-of-box particles: the energies the sweep kernel would see (exact sums)
    ids = np.array([1, 2, 3, 500, 1999], dtype=np.int64)
    e1, e2 = g.delta_e(ids, np.zeros((len(ids), 3)))
    assert np.array_equal(e1, e2)
    ref = pmc.compute_energy_particles_exact(g)
    assert np.array_equal(e1, ref[ids - 1])


def test_unwrapped_input_takes_the_fast_path(gpu):
    """N = 110592: the work counters of a sweep over the whole-box-shifted configuration equal those of the in-box one
    (no visit falls back to the general minimum image), and the measured rates are within 10 % of each other (round 2:
    -13 % at N = 10^6)."""
    import time
    pmc = gpu
    pos, sp, rho = lj_lattice(48, jitter=0.05, seed=42)
    L = (len(pos) / rho) ** (1.0 / 3.0)
    rates = []
    for shifted in (False, True):
        p = pos + (np.random.default_rng(7).integers(-3, 4, size=pos.shape) * L if shifted else 0.0)
        g = pmc.System(p, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
        g.total_energy()
        pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=20)
        t0 = time.perf_counter()
        st = pmc.mc_sweep(g, [_disp(pmc, 0.05)], nsweeps=100)
        rates.append((100 * len(pos) / (time.perf_counter() - t0), st.pair_tests, st.accepted[0]))
    assert abs(rates[1][1] - rates[0][1]) < 0.01 * rates[0][1]
    assert abs(rates[1][2] - rates[0][2]) < 0.01 * rates[0][2]
    assert rates[1][0] > 0.9 * rates[0][0], rates


def test_checkpoint_restart_with_images_is_bit_identical(gpu):
    """a chain that starts folded keeps k = 0 while its particles diffuse out of the box; download + upload
    re-derives k = floor(x / L) and the restarted chain rounds differently.  With pmc_get_images / pmc_set_images (and the
    counters and the energy) in the checkpoint, the restarted chain is the uninterrupted one bit for bit."""
    pmc = gpu
    pos, sp, rho = lj_lattice(16, jitter=0.05, seed=42)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.15}, 1.0)]
    a = pmc.System(pos, sp, rho, 1.5, pmc.LennardJones(1.0, 1.0), seed=3)
    pmc.mc_sweep(a, pool, nsweeps=400)
    L = a.box[0]
    p1 = a.position
    assert ((p1 < 0) | (p1 >= L)).any(), "no particle has left the box: the test would prove nothing"
    img, ctr, e1 = a.get_images(), a.get_counters(), a.energy[0]
    assert not img.any()                                   # the run started folded: frozen at 0
    sa = pmc.mc_sweep(a, pool, nsweeps=10)
    b = pmc.System(p1, sp, rho, 1.5, pmc.LennardJones(1.0, 1.0), seed=3)     # restart WITH the checkpointed images
    b.set_images(img)
    b.set_counters(*ctr)
    sb = pmc.mc_sweep(b, pool, nsweeps=10)
    assert np.array_equal(a.position, b.position) and sa.accepted[0] == sb.accepted[0]
    c = pmc.System(p1, sp, rho, 1.5, pmc.LennardJones(1.0, 1.0), seed=3)     # ... and without: images re-derived, another chain
    c.set_counters(*ctr)
    assert c.get_images().any()
    pmc.mc_sweep(c, pool, nsweeps=10)
    assert not np.array_equal(a.position, c.position)
