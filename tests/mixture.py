"""examples/lj-mixture protocol (run-validation.py:26-61,98-132,155-180), shared by CPU and GPU tests."""
import csv
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EPS = ((1.0, 1.1523), (1.1523, 1.3702))
SIG = ((1.0, 1.0339), (1.0339, 1.0640))
RCUT = 4.0
N = 1000


def reference_rows():
    rows = []
    with open(os.path.join(GOLD, "lj_mixture_calculated_energies.csv")) as f:
        for r in csv.DictReader(f):
            rows.append({k: float(v) for k, v in r.items() if k})
    return rows


def make_config(x, density, seed=0):
    """Rectangular 10^3 lattice centred on the origin, species placed at random
    (run-validation.py:37-61); folded into [0, L) as load_chains does (IO.jl:284)."""
    n2 = round(N * x)
    L = (N / density) ** (1.0 / 3.0)
    a = L / 10
    g = np.stack(np.meshgrid(*[np.arange(10)] * 3, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    pos = (g + 0.5) * a - L / 2
    sp = np.array([1] * (N - n2) + [2] * n2, dtype=np.int32)
    np.random.default_rng(seed).shuffle(sp)
    pos = pos - np.floor_divide(pos, L) * L
    return np.ascontiguousarray(pos), sp, N / L ** 3
