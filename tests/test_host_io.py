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
