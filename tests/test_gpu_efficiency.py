"""-m gpu: what one attempted move of the parallel schedule is worth in attempted moves of the reference's sequential
chain (src/moves.jl:57-81, Arianna mc_sweep!).  The headline unit of bench.py is "attempted moves/s": 4.5 % of the slots
are null (proposals that would leave their cell) and particles are confined to their cell between re-shifts, so the
exchange rate with a sequential attempted move is measured here, on the device, against the oracle's sequential chain:
mean-square displacement after K sweeps and the integrated autocorrelation time of e/N, per attempted move, for the
schedule parameter m = 1 (default) and m = 4 (rounds per cell visit)."""
import json
import os

import numpy as np
import pytest

from inputs import binary_lattice, lj_lattice

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _equilibrated_lj(pmc, n, sweeps=3000):
    pos, sp, rho = lj_lattice(n, jitter=0.05, seed=42)
    g = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=4242)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
    st = pmc.mc_sweep(g, pool, nsweeps=sweeps)
    assert st.energy / len(pos) < -5.0          # the liquid, not the lattice it started from
    return g.position.copy(), sp, rho


def test_move_efficiency_vs_sequential_chain_lj_liquid(gpu):
    """N = 4096 LJ liquid (rho = 1, T = 1, sigma = 0.05; 6 cells of width 2.667 per side: close to the cell width of the
    N = 10^6 benchmark, 2.63 -- the null-move fraction scales with 1 / width).  Sequential leg: 64 independent oracle
    chains of 1500 sweeps (one host process each, ~1 minute); checkerboard leg: 64 GPU chains of 1500 sweeps, m = 1 and 4.
    Asserted: the MSD per attempted move of the checkerboard is 0.85 ... 1.03 of the sequential chain's for m = 1
    (expected ~ 1 - null fraction, a few per cent lower through cell confinement) and 0.78 ... 1.03 for m = 4; the
    integrated autocorrelation time of e/N agrees within a factor 0.6 ... 1.6 (its own statistical error at this sample
    size is ~ 15 %); both legs sample the same mean energy."""
    from oracle import efficiency as E
    pmc = gpu
    p0, sp, rho = _equilibrated_lj(pmc, 16)
    nch, nsw = 64, 1500
    seq_series, seq_msd, seq_acc = E.sequential_leg(p0, sp, rho, 1.0, 0.05, nch, nsw, stride=1)
    out = {}
    for m, lo in ((1, 0.85), (4, 0.78)):
        cb_series, cb_msd, cb_acc, cb_null = E.gpu_cb_leg(pmc, p0, sp, rho, 1.0, 0.05, nch, nsw, stride=1, rounds=m)
        ss = [s[m - 1::m] for s in seq_series] if m > 1 else seq_series      # same sampling stride on both legs
        r = E.efficiency(ss, seq_msd, cb_series, cb_msd)
        r.update({"rounds": m, "acceptance_sequential": seq_acc, "acceptance_checkerboard": cb_acc, "null_fraction": cb_null})
        out[m] = r
        assert lo < r["msd_ratio"] < 1.03, r
        assert r["msd_ratio_sigma"] < 0.03, r
        assert 0.6 < r["tau_ratio_seq_over_cb"] < 1.6, r
        assert abs(r["e_mean_sequential"] - r["e_mean_checkerboard"]) < 0.01, r
        # acceptance: the checkerboard's differs from the sequential chain's by its null moves (which are never accepted)
        assert abs(cb_acc - seq_acc * (1.0 - cb_null)) < 0.01, r
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "move_efficiency_test.json"), "w") as f:
        json.dump(out, f, indent=1)


def test_move_efficiency_kob_andersen_mixture(gpu):
    """The same for the Kob-Andersen 80:20 mixture (rho = 1.2, T = 1, N = 10648: 8 cells of width 2.59 per side;
    Displacement only: swaps do not move particles): MSD per attempted move within 0.85 ... 1.03 of the sequential chain's."""
    from oracle import efficiency as E
    pmc = gpu
    pos, sp, rho = binary_lattice(10648, frac_b=0.2, rho=1.2, jitter=0.05, seed=42)
    g = pmc.System(pos, sp, rho, 1.0, pmc.KobAndersen(), seed=17)
    assert tuple(int(v) for v in g.grid()[0]) == (8, 8, 8)
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
    pmc.mc_sweep(g, pool, nsweeps=3000)
    p0 = g.position.copy()
    nsw = 300
    chains = [pmc.System(p0, sp, rho, 1.0, pmc.KobAndersen(), seed=500, chain=c) for c in range(16)]
    for ch in chains:
        ch.total_energy()
    pmc.mc_sweep_many(chains, pool, nsweeps=nsw)
    cb_msd = np.array([((ch.position - p0) ** 2).sum(1).mean() for ch in chains])
    _, seq_msd, _ = E.sequential_leg(p0, sp, rho, 1.0, 0.05, 16, nsw, stride=nsw, model="ka")
    r = E.efficiency(None, seq_msd, None, cb_msd)
    assert 0.85 < r["msd_ratio"] < 1.03 and r["msd_ratio_sigma"] < 0.03, r
