"""Checks of the Julia-Base arithmetic the path depends on against references that do NOT come from this repository's
reading of Julia's source: exact rational arithmetic (fractions.Fraction) and the mathematical definitions in Julia's
documentation (`fld(x, y)`: "largest integer less than or equal to x / y"; `x^n`: the power).  The oracle and the
device restate `fld` / `pow_body` from the same recollection, so a shared misreading would pass every GPU-vs-oracle
test; it cannot pass these."""
import math
from fractions import Fraction

import numpy as np
import pytest

from inputs import lj_lattice


@pytest.fixture(scope="module")
def O(built):
    from oracle import oracle as O
    return O


def _edge_points(c, rng, kmax):
    """x = k c shifted by -3 ... +3 ulps, for cell indices on both sides of the origin (positions are never wrapped)."""
    xs = []
    for k in list(range(-kmax, kmax + 1)) + [int(v) for v in rng.integers(-10 ** 5, 10 ** 5, 40)]:
        x = k * c
        lo = x
        for _ in range(3):
            lo = math.nextafter(lo, -math.inf)
        p = lo
        for _ in range(7):
            xs.append(p)
            p = math.nextafter(p, math.inf)
    return xs


def _exact_floor(x, c):
    return math.floor(Fraction(x) / Fraction(c))


@pytest.mark.parametrize("N,rho,d,rcut", [(10 ** 6, 1.0, 3, 2.5), (32768, 1.0, 3, 2.5), (1290, 1.192060351952634, 2, 2.5), (10 ** 5, 1.2, 3, 2.5)])
def test_float_fld_is_the_exact_floor_of_the_true_quotient_at_cell_edges(O, N, rho, d, rcut):
    L = O.lib()
    box = (N / rho) ** (1.0 / d)
    cell = box / L.orc_jl_fld(box, rcut)
    assert L.orc_jl_fld(box, rcut) == _exact_floor(box, rcut)
    rng = np.random.default_rng(N)
    bad = [(x, L.orc_jl_fld(x, cell), _exact_floor(x, cell)) for x in _edge_points(cell, rng, 45) if L.orc_jl_fld(x, cell) != _exact_floor(x, cell)]
    assert not bad, bad[:5]
    # documented examples of Julia's manual for fld / mod on floats
    assert L.orc_jl_fld(7.3, 5.5) == 1.0 and L.orc_jl_fld(-7.3, 5.5) == -2.0 and L.orc_jl_fld(7.3, -5.5) == -2.0


def test_compensated_integer_power_is_the_correctly_rounded_power(O):
    """Julia's Float64^Int (pow_body) is documented as accurate to < 1 ulp; a compensated power-by-squaring reaches the
    CORRECTLY rounded value almost always, a plain one misses it in tens of percent of the cases.  Exponents: n/2 of
    the soft-sphere models the path uses (n = 12, 14, 18 -> 6, 7, 9), over (sigma/r)^2 in the range inside a cutoff."""
    L = O.lib()
    rng = np.random.default_rng(12)
    xs = np.concatenate([rng.uniform(0.16, 2.5, 4000), np.exp(rng.uniform(np.log(1e-3), np.log(1e3), 2000))])
    for n in (6, 7, 9, 12):
        worst, equal = 0.0, 0
        for x in xs:
            exact = Fraction(float(x)) ** n
            cr = float(exact)                                    # Fraction -> float rounds to nearest: the correctly rounded power
            got = L.orc_pow_int(float(x), n)
            ulp = math.ulp(cr)
            worst = max(worst, abs(Fraction(got) - exact) / Fraction(ulp))
            equal += (got == cr)
        assert worst < 1.0, (n, float(worst))
        assert equal / len(xs) > 0.99, (n, equal / len(xs))
    # the same statistic for plain repeated squaring, to show the test discriminates: it is visibly worse
    def plain(x, n):
        y = 1.0
        while n > 1:
            if n & 1:
                y *= x
            x *= x
            n >>= 1
        return x * y
    eq_plain = sum(plain(float(x), 9) == float(Fraction(float(x)) ** 9) for x in xs[:2000]) / 2000
    assert eq_plain < 0.9


def test_negative_and_small_exponents_follow_julias_special_cases(O):
    """x^3 = x*x*x and x^-2 = (1/x)^2 are literal_pow-compatible shortcuts in Julia (two roundings each, not
    correctly rounded): the restatement must reproduce exactly those roundings."""
    L = O.lib()
    rng = np.random.default_rng(3)
    for x in rng.uniform(0.3, 3.0, 500):
        x = float(x)
        assert L.orc_pow_int(x, 3) == x * x * x
        rx = 1.0 / x
        assert L.orc_pow_int(x, -2) == rx * rx
        assert L.orc_pow_int(x, 1) == x          # (n = 0 never reaches pow_body: `^` returns one(x) first, and no model has n = 0)


@pytest.mark.gpu
def test_device_cell_membership_at_cell_edges_equals_exact_floor(gpu):
    """get_cell on the device (src/neighbours.jl:178-186) for particles sitting within 3 ulps of a cell face, many
    boxes away from the origin on either side: the cell index must be the exact floor of x / cell folded into the
    grid, as rational arithmetic gives it -- not what an oracle sharing the device's reading of `fld` says."""
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.05, seed=5)            # N = 1728, box 12, 4 cells per side
    s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    cell = float(s.neighbour_list.box[0])
    nc = int(s.neighbour_list.ncells[0])
    xs = _edge_points(cell, np.random.default_rng(8), 12)
    rng = np.random.default_rng(9)
    p = pos.copy()
    idx = rng.permutation(len(p))[: len(xs)]
    assert len(idx) == len(xs)
    axis = rng.integers(0, 3, len(xs))
    for i, a, x in zip(idx, axis, xs):
        p[i, a] = x
    s.upload(p)
    s.build_neighbour_list()
    cs = s.get_cells()
    for i in idx:
        mc = [_exact_floor(float(p[i, a]), cell) % nc for a in range(3)]
        assert cs[i] == 1 + mc[0] * nc * nc + mc[1] * nc + mc[2], (i, p[i], cs[i], mc)       # last dimension fastest
