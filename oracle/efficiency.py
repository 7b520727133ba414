"""Test infrastructure (like everything under oracle/): what an attempted move of the parallel (checkerboard) schedule is
worth in attempted moves of the reference's sequential chain (src/moves.jl:57-81; Arianna `mc_sweep!` = N `mc_step!`s).

Two observables per attempted move, both measured on the same equilibrated liquid with the same move pool:
  * the mean-square displacement after K sweeps (positions are never wrapped, src/moves.jl:46-48, so it is the plain
    difference of the downloaded coordinates);
  * the integrated autocorrelation time of e/N sampled every `stride` sweeps (Sokal's automatic window, c = 6),
    ACF averaged over independent chains.
The sequential leg is `orc_mc_steps_sequential` (the oracle's restatement of the reference chain), one chain per host
process; the checkerboard leg is whatever `cb_leg` the caller passes (the GPU engine in tests/ and bench.py, the CPU
restatement `orc_cb_sweeps_rounds` in the CPU-only tests).  Only tests/, bench.py's CPU leg and examples/ call this.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def acf_fft(x):
    """Normalised autocorrelation function of one series (biased estimator, FFT)."""
    x = np.asarray(x, dtype=np.float64)
    x = x - x.mean()
    n = len(x)
    f = np.fft.rfft(x, 2 * n)
    a = np.fft.irfft(f * np.conj(f))[:n]
    return a / a[0]


def tau_int(series, c=6.0):
    """Integrated autocorrelation time (in samples) of a set of equally long series, ACF averaged over the series,
    summed up to Sokal's automatic window W >= c tau.  Returns (tau, W, relative error estimate)."""
    series = [np.asarray(s, dtype=np.float64) for s in series]
    n = min(len(s) for s in series)
    rho = np.mean([acf_fft(s[:n]) for s in series], axis=0)
    tau, W = 0.5, 1
    for W in range(1, n):
        tau += rho[W]
        if W >= c * tau:
            break
    ntot = n * len(series)
    return float(tau), int(W), float(np.sqrt(2.0 * (2 * W + 1) / ntot))


def _seq_worker(args):
    """One sequential chain (own process = own host core): `nsweeps` sweeps from `pos`, e/N every `stride` sweeps."""
    pos, sp, rho, T, sigma, seed, chain, nsweeps, stride, model = args
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from oracle import oracle as O
    s = O.System(pos, sp, rho, T, O.kob_andersen() if model == "ka" else O.single(O.lennard_jones(1.0, 1.0)), O.LINKEDLIST)
    pool = O.make_pool([{"kind": "Displacement", "sigma": sigma}])
    N = len(pos)
    es, att, acc = [], 0, 0
    for k in range(0, nsweeps, stride):
        a, b = s.mc_steps(pool, seed, chain, k * N, stride * N)
        att += int(a[0])
        acc += int(b[0])
        es.append(s.energy / N)
    p1 = s.positions()
    return np.array(es), float(((p1 - pos) ** 2).sum(1).mean()), acc / max(1, att)


def sequential_leg(pos, sp, rho, T, sigma, nchains, nsweeps, stride=1, seed=1000, processes=None, model="lj"):
    """`nchains` independent sequential chains from the same configuration; returns (list of e/N series, MSDs, acceptance).
    model: "lj" (LennardJones(1, 1), one species) or "ka" (KobAndersen())."""
    import multiprocessing as mp
    jobs = [(pos, sp, rho, T, sigma, seed + c, c, nsweeps, stride, model) for c in range(nchains)]
    nproc = processes or max(1, min(nchains, (os.cpu_count() or 2) // 2))
    if nproc == 1:
        res = [_seq_worker(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(nproc) as pool:
            res = pool.map(_seq_worker, jobs)
    return [r[0] for r in res], np.array([r[1] for r in res]), float(np.mean([r[2] for r in res]))


def oracle_cb_leg(pos, sp, box, T, sigma, nchains, nsweeps, stride=1, rounds=1, seed=2000):
    """The checkerboard leg on the CPU restatement (slow: CPU-only tests and cross-checks)."""
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from oracle import oracle as O
    models = O.single(O.lennard_jones(1.0, 1.0))
    pool = O.make_pool([{"kind": "Displacement", "sigma": sigma}])
    N = len(pos)
    series, msd = [], []
    step = max(stride, rounds)
    for c in range(nchains):
        cb = O.Checkerboard(pos, sp, box, T, models, seed=seed + c, chain=c)
        es = []
        for k in range(0, nsweeps, step):
            cb.sweeps(pool, k, step, rounds=rounds)
            es.append(cb.energy_fix / 2.0 ** 32 / N)
        series.append(np.array(es))
        msd.append(float(((cb.positions() - pos) ** 2).sum(1).mean()))
    return series, np.array(msd)


def gpu_cb_leg(pmc, pos, sp, rho, T, sigma, nchains, nsweeps, stride=1, rounds=1, seed=2000):
    """The checkerboard leg on the device: `nchains` chains in shared launches (pmc_sweep_many), e/N every
    max(stride, rounds) sweeps (a pass of `rounds` sweeps never spans calls)."""
    chains = [pmc.System(pos, sp, rho, T, pmc.LennardJones(1.0, 1.0), seed=seed, chain=c) for c in range(nchains)]
    for ch in chains:
        ch.rounds = rounds
        ch.total_energy()
    pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": sigma}, 1.0)]
    step = max(stride, rounds)
    N = len(pos)
    series = [[] for _ in chains]
    att = acc = nul = 0
    for _ in range(0, nsweeps, step):
        stats = pmc.mc_sweep_many(chains, pool, nsweeps=step)
        for c, st in enumerate(stats):
            series[c].append(st.energy / N)
            att += int(st.attempted[0])
            acc += int(st.accepted[0])
            nul += int(st.null_moves)
    msd = np.array([float(((ch.position - pos) ** 2).sum(1).mean()) for ch in chains])
    return [np.array(s) for s in series], msd, acc / max(1, att), nul / max(1, att)


def efficiency(seq_series, seq_msd, cb_series, cb_msd):
    """Ratios checkerboard / sequential per attempted move: MSD (higher = more efficient) and, from the autocorrelation
    times, tau_seq / tau_cb (higher = more efficient), each with a one-sigma estimate."""
    m_seq, m_cb = float(np.mean(seq_msd)), float(np.mean(cb_msd))
    e_seq = float(np.std(seq_msd, ddof=1) / np.sqrt(len(seq_msd))) if len(seq_msd) > 1 else float("nan")
    e_cb = float(np.std(cb_msd, ddof=1) / np.sqrt(len(cb_msd))) if len(cb_msd) > 1 else float("nan")
    r_msd = m_cb / m_seq
    out = {"msd_sequential": m_seq, "msd_checkerboard": m_cb, "msd_ratio": r_msd,
           "msd_ratio_sigma": r_msd * float(np.hypot(e_seq / m_seq, e_cb / m_cb))}
    if seq_series is not None and cb_series is not None:
        t_seq, w_seq, r_seq = tau_int(seq_series)
        t_cb, w_cb, r_cb = tau_int(cb_series)
        out.update({"tau_int_sequential": t_seq, "tau_int_checkerboard": t_cb, "tau_window": [w_seq, w_cb],
                    "tau_ratio_seq_over_cb": t_seq / t_cb, "tau_ratio_sigma": (t_seq / t_cb) * float(np.hypot(r_seq, r_cb)),
                    "e_mean_sequential": float(np.mean([s.mean() for s in seq_series])),
                    "e_mean_checkerboard": float(np.mean([s.mean() for s in cb_series]))})
    return out
