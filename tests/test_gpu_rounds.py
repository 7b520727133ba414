"""-m gpu: the schedule parameter m ("rounds per cell visit", pmc_set_rounds) against its CPU restatement
`orc_cb_sweeps_rounds`: passes of min(m, sweeps left) sweeps share one grid shift and one staged neighbourhood per cell."""
import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice

pytestmark = pytest.mark.gpu


def _pool(pmc, sigma=0.05):
    return [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, 1.0)]


@pytest.mark.parametrize("team", ["1", "2", "4", "8"])
@pytest.mark.parametrize("rounds,calls", [(2, (4, 3)), (3, (7,)), (5, (2, 1, 6))])
def test_rounds_bitexact_against_oracle(gpu, monkeypatch, team, rounds, calls):
    """Single species, every team width; calls whose sweep counts are not multiples of m end in a shorter pass."""
    from oracle import oracle as O
    pmc = gpu
    monkeypatch.setenv("PMC_TEAM", team)
    pos, sp, rho = lj_lattice(14, jitter=0.06, seed=3)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=9)
    g.rounds = rounds
    assert g.rounds == rounds
    o = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    cb = O.Checkerboard(pos, sp, o.box, 1.0, o.models, seed=9)
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    done = 0
    for n in calls:
        st = pmc.mc_sweep(g, _pool(pmc), nsweeps=n)
        att, acc = cb.sweeps(opool, done, n, rounds=rounds)
        done += n
        assert st.attempted[0] == att[0] == n * len(pos) and st.accepted[0] == acc[0]
        assert st.energy_fix == cb.energy_fix
        assert np.array_equal(g.position, cb.positions())
        assert st.sweep_kernel_launches == (n + rounds - 1) // rounds
    assert g.get_counters()[1] == done
    _, fix = g.total_energy()
    assert fix == cb.energy_fix


@pytest.mark.parametrize("team", ["1", "8"])
def test_rounds_with_swaps_bitexact_against_oracle(gpu, monkeypatch, team):
    """Kob-Andersen mixture with 30 % DiscreteSwap slots, m = 3: positions, species, per-move counts, energy."""
    from oracle import oracle as O
    pmc = gpu
    monkeypatch.setenv("PMC_TEAM", team)
    pos, sp, rho = binary_lattice(4000, frac_b=0.4, rho=1.2, jitter=0.05, seed=5)
    g = pmc.System(pos, sp, rho, 0.8, pmc.KobAndersen(), seed=21)
    g.rounds = 3
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.07}, 0.7),
            pmc.Move(pmc.DiscreteSwap([1, 2], g), pmc.DoubleUniform(), [], 0.3)]
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.07, "probability": 0.7},
                         {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.3}])
    cb = O.Checkerboard(pos, sp, np.asarray(g.box), 0.8, O.kob_andersen(), seed=21)
    st = pmc.mc_sweep(g, pool, nsweeps=8)
    att, acc = cb.sweeps(opool, 0, 8, rounds=3)
    assert list(st.attempted[:2]) == list(att) and list(st.accepted[:2]) == list(acc) and acc[1] > 50
    assert st.energy_fix == cb.energy_fix
    assert np.array_equal(g.position, cb.positions()) and np.array_equal(g.species, cb.species())


def test_rounds_many_replicas_equal_single_chains(gpu):
    """pmc_sweep_many with m = 2 on four chains = each chain alone = the oracle."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=8)
    chains = [pmc.System(pos, sp, rho, 1.0 + 0.1 * c, pmc.LennardJones(1.0, 1.0), seed=77, chain=c) for c in range(4)]
    for c in chains:
        c.rounds = 2
    stats = pmc.mc_sweep_many(chains, _pool(pmc), nsweeps=5)
    opool = O.make_pool([{"kind": "Displacement", "sigma": 0.05}])
    o = O.System(pos, sp, rho, 1.0, O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    for c, (ch, st) in enumerate(zip(chains, stats)):
        cb = O.Checkerboard(pos, sp, o.box, 1.0 + 0.1 * c, o.models, seed=77, chain=c)
        att, acc = cb.sweeps(opool, 0, 5, rounds=2)
        assert st.accepted[0] == acc[0] and st.energy_fix == cb.energy_fix
        assert np.array_equal(ch.position, cb.positions())


def test_rounds_argument_checks(gpu):
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=8)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=1)
    assert g.rounds == 1
    for bad in (0, -1, 17):
        with pytest.raises(Exception):
            g.rounds = bad
    g.rounds = 16
    g.rounds = 1
    assert g.rounds == 1
