"""-m gpu: the reference's other list types on the device's reference-order engine, against the
oracle's restatement of src/neighbours.jl:13-44 (EmptyList), :61-217 (CellList), :387-543 (VerletList).
Everything bit-exact: energies are summed in the order each list's iterator yields, moves maintain the
cell / neighbour vectors exactly as update_neighbour_list! does."""
import os

import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _config0(O):
    from test_oracle_golden import _fold, _load
    pos, sp, box = _load("config_0.xyz")
    return _fold(O, pos, box), sp, len(pos) / np.prod(box)


def _lists(pmc, O):
    return [("EmptyList", pmc.EmptyList, O.EMPTYLIST, None, 0.0),
            ("CellList", pmc.CellList, O.CELLLIST, None, 0.0),
            ("VerletList", pmc.VerletList, O.VERLETLIST, {"dr": 0.2}, 0.2)]


def _same_lists(g, o, lt):
    if lt == "EmptyList":
        return                      # no cells, no lists (src/neighbours.jl:13-44)
    assert np.array_equal(g.get_cells(), o.cells())
    if lt in ("CellList", "VerletList"):
        ncell = int(np.prod(g.neighbour_list.ncells))
        for c in range(1, ncell + 1):
            assert np.array_equal(g.get_cell_members(c), o.cell_members(c)), (lt, c)
    if lt == "VerletList":
        for i in range(1, g.N + 1):
            assert np.array_equal(g.get_verlet_neighbours(i), o.neighbours_of(i)), (lt, i)


def test_list_types_golden_energy_and_structures(gpu):
    """test/runtests.jl:22-38 on the device: energy/N = -2.676832 (atol 1e-6) for every list type, and equal
    to the oracle's value bit for bit; per-particle energies, cs, cells[] and neighbours[] identical."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _config0(O)
    for name, cls, olt, lp, dr in _lists(pmc, O):
        g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=cls, list_parameters=lp)
        o = O.System(pos, sp, rho, 0.231, O.jbb(), olt, dr)
        assert abs(g.energy[0] / g.N - (-2.676832)) < 1e-6, name
        assert g.energy[0] == o.energy, name
        assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles()), name
        if name != "EmptyList":
            assert tuple(g.neighbour_list.ncells) == tuple(int(v) for v in o.cell_geometry()[1]), name
        _same_lists(g, o, name)


@pytest.mark.parametrize("swaps", [False, True])
def test_list_types_sequential_chain_bitexact(gpu, swaps):
    """The reference's cross-list simulation test (test/runtests.jl:55-129) as a parity test: the same seeded chain
    on the device and on the oracle, for every list type; large steps so that Verlet lists are rebuilt often."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _config0(O)
    if swaps:
        spec = [{"kind": "Displacement", "sigma": 0.12, "probability": 0.6},
                {"kind": "DiscreteSwap", "species": (1, 3), "probability": 0.2},
                {"kind": "DiscreteSwap", "species": (2, 3), "probability": 0.2}]
    else:
        spec = [{"kind": "Displacement", "sigma": 0.15}]
    traces = {}
    for name, cls, olt, lp, dr in _lists(pmc, O) + [("LinkedList", pmc.LinkedList, O.LINKEDLIST, None, 0.0)]:
        g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=cls, list_parameters=lp, seed=10)
        o = O.System(pos, sp, rho, 0.231, O.jbb(), olt, dr)
        pool = []
        for m in spec:
            if m["kind"] == "Displacement":
                pool.append(pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": m["sigma"]}, m.get("probability", 1.0)))
            else:
                pool.append(pmc.Move(pmc.DiscreteSwap(list(m["species"]), g), pmc.DoubleUniform(), [], m["probability"]))
        st = pmc.mc_sweep(g, pool, nsweeps=2, mode="sequential")
        att, acc = o.mc_steps(O.make_pool(spec), 10, 0, 0, 2 * g.N)
        assert np.array_equal(st.attempted[:len(spec)], att[:len(spec)]) and np.array_equal(st.accepted[:len(spec)], acc[:len(spec)]), name
        assert np.array_equal(g.position, o.positions()), name
        assert np.array_equal(g.species, o.species()), name
        assert g.energy[0] == o.energy, name
        if name != "LinkedList":
            _same_lists(g, o, name)
        assert np.array_equal(pmc.compute_energy_particles(g), o.energy_particles()), name
        traces[name] = (g.energy[0], tuple(int(v) for v in st.accepted[:len(spec)]))
    # the reference's own assertion: the lists agree with each other to 1e-6 (here: same accept counts too)
    e0, a0 = traces["LinkedList"]
    for name, (e, a) in traces.items():
        assert abs(e - e0) < 1e-6 * len(pos) and a == a0, name


def test_list_types_per_move_verbs_and_hand_over(gpu):
    """perform / revert verbs on a 3-d mixture with every list type, then throughput sweeps and back: after
    checkerboard sweeps the list is rebuilt from the current state (build_neighbour_list!)."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = binary_lattice(1728, frac_b=0.2, rho=1.2, jitter=0.05, seed=8)
    rng = np.random.default_rng(1)
    for name, cls, olt, lp, dr in _lists(pmc, O):
        g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), list_type=cls, list_parameters=lp, seed=4)
        o = O.System(pos, sp, rho, 1.0, O.kob_andersen(), olt, dr)
        assert g.energy[0] == o.energy, name
        for k in range(30):
            i = int(rng.integers(1, g.N + 1))
            delta = rng.normal(0, 0.3, 3)
            act = pmc.Displacement(i, delta, 0.0)
            e1, e2 = pmc.perform_action(g, act)
            r1, r2, rde = o.displacement_perform(i, delta)
            assert (e1, e2, act.de) == (r1, r2, rde), (name, k)
            if k % 3:
                pmc.invert_action(act, g)
                pmc.revert_action(g, act)
                o.displacement_revert(i, -delta, rde)
            assert g.energy[0] == o.energy, (name, k)
        _same_lists(g, o, name)
        if name != "EmptyList":
            pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
            pmc.mc_sweep(g, pool, nsweeps=3)
            o2 = O.System(g.position, g.species, rho, 1.0, O.kob_andersen(), olt, dr)
            _same_lists(g, o2, name)
            assert np.array_equal(pmc.compute_energy_particles(g), o2.energy_particles()), name


def test_many_literal_loop_chains_in_one_launch(gpu):
    """the chains of a `parallel = true` run (src/ParticlesMC.jl:164,246) that only the literal loop
    serves -- the reference's CI system with a VerletList dr = 0.2 (test/runtests.jl:41-91), an EmptyList chain forced onto engine
    1, a pool that swaps within one species -- share ONE launch of k_ref_steps_many, one workgroup per chain, any mix of shapes
    and list types.  Every chain must end exactly as the oracle's sequential driver does with the chain's own key: positions,
    species, accept counts, system.energy, cells[] and neighbours[] bit for bit."""
    from oracle import oracle as O
    pmc = gpu
    pos, sp, rho = _config0(O)
    spec = [{"kind": "Displacement", "sigma": 0.12, "probability": 0.7},
            {"kind": "DiscreteSwap", "species": (1, 3), "probability": 0.2},
            {"kind": "DiscreteSwap", "species": (2, 2), "probability": 0.1}]      # i ~ species 2, j ~ species 2: the literal loop only
    chains, oracles, names = [], [], []
    for k in range(5):
        lt = [("VerletList", pmc.VerletList, O.VERLETLIST, {"dr": 0.2}, 0.2), ("CellList", pmc.CellList, O.CELLLIST, None, 0.0),
              ("EmptyList", pmc.EmptyList, O.EMPTYLIST, None, 0.0)][k % 3]
        g = pmc.System(pos, sp, rho, 0.231, pmc.JBB(), list_type=lt[1], list_parameters=lt[3], seed=10, chain=k)
        chains.append(g)
        oracles.append(O.System(pos, sp, rho, 0.231, O.jbb(), lt[2], lt[4]))
        names.append(lt[0])
    pool = []
    for m in spec:
        if m["kind"] == "Displacement":
            pool.append(pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": m["sigma"]}, m["probability"]))
        else:
            pool.append(pmc.Move(pmc.DiscreteSwap(list(m["species"]), chains[0]), pmc.DoubleUniform(), [], m["probability"]))
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=2, mode="sequential")
    assert all(st.sweep_kernel_launches == 1 for st in sts)           # one shared launch
    for k, (g, o) in enumerate(zip(chains, oracles)):
        att, acc = o.mc_steps(O.make_pool(spec), 10, k, 0, 2 * g.N)
        assert np.array_equal(sts[k].attempted[:3], att[:3]) and np.array_equal(sts[k].accepted[:3], acc[:3]), (k, names[k])
        assert np.array_equal(g.position, o.positions()) and np.array_equal(g.species, o.species()), (k, names[k])
        assert g.energy[0] == o.energy, (k, names[k])
        _same_lists(g, o, names[k])
    assert len({tuple(int(v) for v in st.accepted[:3]) for st in sts}) > 1      # the chains differ (own keys)
    # ... and a second call continues every chain where the first stopped
    sts = pmc.mc_sweep_many(chains, pool, nsweeps=1, mode="sequential")
    for k, (g, o) in enumerate(zip(chains, oracles)):
        att, acc = o.mc_steps(O.make_pool(spec), 10, k, 2 * g.N, g.N)
        assert np.array_equal(sts[k].accepted[:3], acc[:3]) and np.array_equal(g.position, o.positions()) and g.energy[0] == o.energy, (k, names[k])
