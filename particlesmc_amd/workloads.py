"""Seeded synthetic workloads: shared by bench.py, smoke(), the examples and the parity tests."""
import numpy as np


def lj_lattice(n, jitter=0.05, seed=42, rho=1.0, d=3):
    """n^d simple-cubic sites at density rho, each coordinate jittered by U(-jitter, jitter)."""
    N = n ** d
    L = (N / rho) ** (1.0 / d)
    a = L / n
    g = np.stack(np.meshgrid(*[np.arange(n)] * d, indexing="ij"), axis=-1).reshape(-1, d).astype(np.float64)
    rng = np.random.default_rng(seed)
    pos = (g + 0.5) * a + rng.uniform(-jitter, jitter, size=(N, d))
    return np.ascontiguousarray(pos), np.ones(N, dtype=np.int32), rho


def binary_lattice(N, frac_b=0.2, rho=1.2, jitter=0.05, seed=42, d=3):
    """First N sites of the smallest cubic lattice holding N particles, species shuffled."""
    n = int(np.ceil(N ** (1.0 / d) - 1e-9))
    L = (N / rho) ** (1.0 / d)
    a = L / n
    g = np.stack(np.meshgrid(*[np.arange(n)] * d, indexing="ij"), axis=-1).reshape(-1, d).astype(np.float64)[:N]
    rng = np.random.default_rng(seed)
    pos = (g + 0.5) * a + rng.uniform(-jitter, jitter, size=(N, d))
    nb = int(round(N * frac_b))
    sp = np.ones(N, dtype=np.int32)
    sp[rng.permutation(N)[:nb]] = 2
    return np.ascontiguousarray(pos), sp, rho


def random_gas(N, rho, d, seed=0, min_dist=0.8):
    """Random insertion with a minimum distance (small N only), unwrapped positions shifted
    by whole boxes to exercise the never-wrapped convention."""
    rng = np.random.default_rng(seed)
    L = (N / rho) ** (1.0 / d)
    pts = []
    while len(pts) < N:
        x = rng.uniform(0, L, size=d)
        ok = True
        for y in pts:
            dx = x - y
            dx -= np.round(dx / L) * L
            if np.dot(dx, dx) < min_dist ** 2:
                ok = False
                break
        if ok:
            pts.append(x)
    pos = np.array(pts)
    pos += rng.integers(-2, 3, size=(N, d)) * L
    return np.ascontiguousarray(pos), rho
