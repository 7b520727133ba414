"""CPU: host-side mirror of the reference's configuration formats and schedulers."""
import io
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pmc(built):
    import particlesmc_amd
    return particlesmc_amd


def test_three_formats_hold_the_same_configuration(pmc):
    """test/runtests.jl:27-31: positions and species equal across .exyz / .lmp (and .xyz)."""
    a = pmc.load_configuration(os.path.join(GOLD, "config_0.xyz"))
    b = pmc.load_configuration(os.path.join(GOLD, "config_0.exyz"))
    c = pmc.load_configuration(os.path.join(GOLD, "config_0.lmp"))
    for x in (a, b, c):
        assert x["N"] == 1290 and x["d"] == 2
    assert np.array_equal(a["position"], b["position"]) and np.array_equal(a["position"], c["position"])
    assert np.array_equal(a["species"], b["species"]) and np.array_equal(a["species"], c["species"])
    assert np.array_equal(a["box"], b["box"]) and np.allclose(a["box"], c["box"], rtol=0, atol=1e-12)
    assert a["N"] / np.prod(a["box"]) == 1.192060351952634


def test_fold_back_matches_julia_fld(pmc):
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-100, 100, 2000), [-16.4481, 16.4481, 0.0, -32.8962, 32.8962, 65.7924]])
    box = 32.8962
    got = pmc.fold_back(x, box)
    want = np.array([v - O.lib().orc_jl_fld(float(v), box) * box for v in x])
    assert np.array_equal(got, want)
    assert np.all(got >= 0) and np.all(got <= box)


class _FakeSystem:
    def __init__(self, pos, sp, box):
        self.position, self.species, self.box = pos, sp, np.array(box)
        self.N, self.d = pos.shape
        self.density, self.temperature = 1.25, 0.5

    def __len__(self):
        return self.N


@pytest.mark.parametrize("fmt_name,ext", [("XYZ", ".xyz"), ("EXYZ", ".exyz"), ("LAMMPS", ".lammpstrj")])
def test_writers_round_trip_through_readers(pmc, tmp_path, fmt_name, ext):
    from particlesmc_amd import io as pio
    rng = np.random.default_rng(1)
    for d in (2, 3):
        pos = np.round(rng.uniform(0, 7, (40, d)), 6)
        sp = rng.integers(1, 4, 40).astype(np.int32)
        s = _FakeSystem(pos, sp, [7.0] * d)
        path = tmp_path / f"frame{d}{ext}"
        with open(path, "w") as f:
            pio.store_trajectory(f, s, 12, getattr(pio, fmt_name)())
        cfg = pmc.load_configuration(str(path))
        assert cfg["N"] == 40 and cfg["d"] == d
        assert np.array_equal(cfg["species"], sp)
        assert np.allclose(cfg["position"], pos, rtol=0, atol=5e-7)      # digits = 6, IO.jl:365
        assert np.allclose(cfg["box"][:d], 7.0)


def test_build_schedule(pmc):
    assert pmc.build_schedule(10, 0, 5) == [0, 5, 10]
    assert pmc.build_schedule(100, 0, 10)[-1] == 100
    s = pmc.build_schedule(100, 0, [0, 1, 2, 4, 8])
    assert s[:6] == [0, 1, 2, 4, 8, 9] and max(s) <= 100 and s == sorted(set(s))


def test_model_parsing(pmc):
    from particlesmc_amd.io import parse_model
    assert len(parse_model("JBB", 3)) == 3 and len(parse_model("KobAndersen()", 2)) == 2
    spec = {"1-1": {"name": "LennardJones", "epsilon": 1.0, "sigma": 1.0, "rcut": 4.0, "shift_potential": False},
            "1-2": {"name": "LennardJones", "epsilon": 1.1523, "sigma": 1.0339, "rcut": 4.0, "shift_potential": False},
            "2-2": {"name": "LennardJones", "epsilon": 1.3702, "sigma": 1.064, "rcut": 4.0, "shift_potential": False}}
    mm = parse_model(spec, 2)                 # examples/lj-mixture shape, run-validation.py:26-34
    assert mm[0][1].eps == mm[1][0].eps == 1.1523 and mm[1][1].shift == 0.0 and mm[0][0].rcut == 4.0
    with pytest.raises(ValueError):
        parse_model("GeneralKG", 2)


def test_cli_rejects_bad_input(pmc, tmp_path):
    from particlesmc_amd import cli
    with pytest.raises(FileNotFoundError):
        cli.particlesmc(str(tmp_path / "missing.toml"))
    p = tmp_path / "p.toml"
    p.write_text('[system]\nconfig = "nope.xyz"\ntemperature = 1.0\ndensity = 1.0\nmodel = "JBB"\n'
                 '[simulation]\nsteps = 1\nseed = 1\nparallel = false\n')
    with pytest.raises(FileNotFoundError, match="Configuration file"):
        cli.particlesmc(str(p))


def test_run_writes_the_reference_output_layout(pmc, tmp_path, monkeypatch):
    """run!(Simulation) on the host side, with the device sweeps stubbed out: scheduler times, the
    chains/<c>/energy.dat, moves/<m>/acceptance.dat, trajectory and last-frame files of
    src/ParticlesMC.jl:249-291 (no GPU needed)."""
    from particlesmc_amd import simulation as sim
    from particlesmc_amd import io as pio

    class Fake(_FakeSystem):
        def __init__(self):
            rng = np.random.default_rng(0)
            super().__init__(rng.uniform(0, 5, (12, 3)), rng.integers(1, 3, 12).astype(np.int32), [5.0, 5.0, 5.0])
            self.e = -3.0
            self.energy = self

        def __getitem__(self, k):
            return self.e

    calls = []

    def fake_many(systems, pool, nsweeps=1, mode="checkerboard"):
        calls.append(nsweeps)
        out = []
        for s in systems:
            s.e -= 0.1 * nsweeps

            class St:
                attempted = np.array([12 * nsweeps, 4 * nsweeps])
                accepted = np.array([6 * nsweeps, 1 * nsweeps])
            out.append(St())
        return out

    monkeypatch.setattr(sim, "mc_sweep_many", fake_many)
    chains = [Fake(), Fake()]
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 0.75),
            pmc.Move(pmc.DiscreteSwap(1, 1, (1, 2), (6, 6), 0.0), pmc.DoubleUniform(), [], 0.25)]
    algs = [dict(algorithm="Metropolis", pool=pool, seed=1, parallel=True, sweepstep=12),
            dict(algorithm="StoreCallbacks", callbacks=[sim.energy], scheduler=sim.build_schedule(20, 0, 5)),
            dict(algorithm="StoreAcceptance", dependencies=["Metropolis"], scheduler=sim.build_schedule(20, 0, 10)),
            dict(algorithm="StoreTrajectories", scheduler=sim.build_schedule(20, 0, 10), fmt=pio.EXYZ()),
            dict(algorithm="StoreLastFrames", scheduler=[20], fmt=pio.LAMMPS()),
            dict(algorithm="PrintTimeSteps", scheduler=sim.build_schedule(20, 0, 10))]
    s = sim.run(sim.Simulation(chains, algs, 20, path=str(tmp_path / "out")))
    assert calls == [5, 5, 5, 5] and s.attempted.shape == (2, 2) and s.attempted[0, 0] == 240
    for c in (1, 2):
        en = np.loadtxt(tmp_path / "out" / "chains" / str(c) / "energy.dat")
        assert list(en[:, 0]) == [0, 5, 10, 15, 20]
        assert np.allclose(en[:, 1], [(-3.0 - 0.1 * t) / 12 for t in (0, 5, 10, 15, 20)])
        last = pmc.load_configuration(str(tmp_path / "out" / "chains" / str(c) / "lastframe.lammpstrj"))
        assert last["N"] == 12 and last["d"] == 3
        traj = open(tmp_path / "out" / "chains" / str(c) / "trajectory.exyz").read().split("\n")
        assert traj[0] == "12" and traj[1].startswith('Lattice="5.0 0.0 0.0 0.0 5.0 0.0 0.0 0.0 5.0"') and traj[14] == "12"
    a1 = np.loadtxt(tmp_path / "out" / "moves" / "1" / "acceptance.dat")
    a2 = np.loadtxt(tmp_path / "out" / "moves" / "2" / "acceptance.dat")
    assert list(a1[:, 0]) == [0, 10, 20] and a1[-1, 1] == 0.5 and a2[-1, 1] == 0.25


def test_toml_surface_builds_the_same_objects(pmc, tmp_path, monkeypatch):
    """The keys of src/ParticlesMC.jl:137-299 -> pool and output algorithms (device creation stubbed)."""
    from particlesmc_amd import cli

    class FakeChain:
        N = 1290
        box = np.array([32.8962, 32.8962])
        species = np.array([1] * 600 + [2] * 330 + [3] * 360)

        def __len__(self):
            return self.N

    seen = {}

    def fake_load_chains(config, args=None, seed=0, **kw):
        seen.update(args)
        seen["seed"] = seed
        return [FakeChain()]

    monkeypatch.setattr(cli, "load_chains", fake_load_chains)
    (tmp_path / "c.xyz").write_text("0\n")
    params = {
        "system": {"config": "c.xyz", "temperature": 0.231, "density": 1.19, "model": "JBB", "list_type": "LinkedList"},
        "simulation": {"steps": 100, "seed": 10, "parallel": False, "output_path": "data",
                       "move": [{"action": "Displacement", "probability": 0.2, "policy": "SimpleGaussian", "parameters": {"sigma": 0.05}},
                                {"action": "DiscreteSwap", "probability": 0.8, "policy": "DoubleUniform", "parameters": {"species": [1, 3]}}],
                       "output": [{"algorithm": "StoreCallbacks", "callbacks": ["energy"], "scheduler_params": {"linear_interval": 10}},
                                  {"algorithm": "StoreLastFrames", "scheduler_params": {"linear_interval": 100}, "fmt": "EXYZ"}]},
    }
    s = cli.build_simulation(params, base_dir=str(tmp_path))
    assert seen["temperature"] == 0.231 and seen["model"] == "JBB" and seen["seed"] == 10 and seen["list_type"] == "LinkedList"
    metro = s.algorithm_list[0]
    assert metro["sweepstep"] == 1290 and [m.probability for m in metro["pool"]] == [0.2, 0.8]
    assert metro["pool"][0].sigma == 0.05 and metro["pool"][1].action.species == (1, 3)
    assert metro["pool"][1].action.particles_per_species == (600, 360)
    assert s.algorithm_list[1]["scheduler"] == list(range(0, 101, 10)) and s.algorithm_list[2]["fmt"].extension == ".exyz"
    bad = dict(params, simulation=dict(params["simulation"], move=[{"action": "Displacement", "probability": 1.0, "policy": "SimpleGaussian"}]))
    with pytest.raises(ValueError, match="Missing parameter 'sigma'"):
        cli.build_simulation(bad, base_dir=str(tmp_path))
    bad = dict(params, simulation=dict(params["simulation"], move=[{"action": "MoleculeFlip", "probability": 1.0, "policy": "DoubleUniform"}]))
    with pytest.raises(ValueError, match="Unsupported action"):
        cli.build_simulation(bad, base_dir=str(tmp_path))
