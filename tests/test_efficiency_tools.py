"""CPU: the estimators of oracle/efficiency.py on series with a known answer, and the two oracle legs on a tiny system."""
import numpy as np

from inputs import lj_lattice


def test_tau_int_of_an_ar1_process():
    from oracle import efficiency as E
    rng = np.random.default_rng(1)
    phi = 0.9                                     # tau_int = (1 + phi) / (2 (1 - phi)) = 9.5
    series = []
    for _ in range(16):
        x = np.zeros(20000)
        eps = rng.normal(size=len(x))
        for k in range(1, len(x)):
            x[k] = phi * x[k - 1] + eps[k]
        series.append(x)
    tau, W, rel = E.tau_int(series)
    assert abs(tau - 9.5) < 3 * rel * 9.5 and rel < 0.05 and W >= 6 * tau - 1
    white = [rng.normal(size=5000) for _ in range(4)]
    assert abs(E.tau_int(white)[0] - 0.5) < 0.1


def test_both_oracle_legs_sample_the_same_liquid(built):
    """N = 1728 (4 cells per side): a short run of each leg from one configuration; same mean energy within noise,
    MSD ratio in a wide band (the precise measurement is the -m gpu test)."""
    from oracle import efficiency as E
    from oracle import oracle as O
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=3)
    box = np.array([(len(pos) / rho) ** (1.0 / 3.0)] * 3)
    cb = O.Checkerboard(pos, sp, box, 1.0, O.single(O.lennard_jones(1.0, 1.0)), seed=5)
    cb.sweeps(O.make_pool([{"kind": "Displacement", "sigma": 0.05}]), 0, 400)
    p0 = cb.positions()
    ss, smsd, sacc = E.sequential_leg(p0, sp, rho, 1.0, 0.05, 2, 60, stride=2, processes=2)
    cs, cmsd = E.oracle_cb_leg(p0, sp, box, 1.0, 0.05, 2, 60, stride=2, rounds=2)
    r = E.efficiency(ss, smsd, cs, cmsd)
    assert len(ss[0]) == len(cs[0]) == 30 and 0.3 < sacc < 0.5
    assert 0.7 < r["msd_ratio"] < 1.15 and abs(r["e_mean_sequential"] - r["e_mean_checkerboard"]) < 0.08
