"""Pair-interaction models: host mirror of src/models.jl.

The derived constants (eps4, sigma2, rcut2, shift, ...) are computed by the
host-side constructors of libpmc_hip.so, which restate the reference
constructors operation for operation (include/pmc.h).
"""
import ctypes as C

from . import _lib


class Model:
    """abstract type Model (src/models.jl:8)."""

    def __init__(self, c, name):
        self._c = c
        self.name = name

    # field access in the reference's spelling
    @property
    def rcut(self):
        return self._c.rcut

    @property
    def rcut2(self):
        return self._c.rcut2

    @property
    def shift(self):
        return self._c.shift

    @property
    def eps(self):
        return self._c.eps

    @property
    def sigma(self):
        return self._c.sigma

    def c_struct(self):
        return self._c

    def __repr__(self):
        return f"{type(self).__name__}(eps={self.eps}, sigma={self.sigma}, rcut={self.rcut})"


class LennardJones(Model):
    """LennardJones(eps, sigma; rcut=2.5sigma, shift_potential=true) -- src/models.jl:99-119."""

    def __init__(self, eps, sigma, rcut=None, name="LennardJones", shift_potential=True):
        c = _lib.PairModel()
        _lib.check(_lib.lib().pmc_model_lennard_jones(float(eps), float(sigma),
                                                      2.5 * sigma if rcut is None else float(rcut),
                                                      int(bool(shift_potential)), C.byref(c)))
        super().__init__(c, name)


class SoftSpheres(Model):
    """SoftSpheres(eps, sigma, n; rcut=2.5sigma) -- src/models.jl:52-70."""

    def __init__(self, eps, sigma, n, rcut=None, name="SoftSpheres"):
        c = _lib.PairModel()
        _lib.check(_lib.lib().pmc_model_soft_spheres(float(eps), float(sigma), int(n),
                                                     2.5 * sigma if rcut is None else float(rcut), C.byref(c)))
        super().__init__(c, name)
        self.n = int(n)


class SmoothLennardJones(Model):
    """SmoothLennardJones(eps, sigma; rcut=2.5sigma) -- src/models.jl:137-160."""

    def __init__(self, eps, sigma, rcut=None, name="SmoothLennardJones"):
        c = _lib.PairModel()
        _lib.check(_lib.lib().pmc_model_smooth_lennard_jones(float(eps), float(sigma),
                                                             2.5 * sigma if rcut is None else float(rcut),
                                                             C.byref(c)))
        super().__init__(c, name)


def potential(r2, model):
    """potential(r2, model) -- src/models.jl:72-74,121-123,162-167 (host evaluation)."""
    u = C.c_double()
    _lib.check(_lib.lib().pmc_potential(float(r2), C.byref(model.c_struct()), C.byref(u)))
    return u.value


def cutoff(model):      # src/models.jl:228
    return model.rcut


def cutoff2(model):     # src/models.jl:229
    return model.rcut2


def _matrix(ctor, eps, sigma):
    S = len(eps)
    return [[ctor(eps[i][j], sigma[i][j]) for j in range(S)] for i in range(S)]


def KobAndersen():
    """src/models.jl:125-133."""
    return _matrix(LennardJones, [[1.0, 1.5], [1.5, 0.5]], [[1.0, 0.8], [0.8, 0.88]])


def BHHP():
    """src/models.jl:76-84 (n = 12)."""
    return _matrix(lambda e, s: SoftSpheres(e, s, 12), [[1.0, 1.0], [1.0, 1.0]], [[1.0, 1.2], [1.2, 1.4]])


def JBB():
    """src/models.jl:169-179."""
    return _matrix(SmoothLennardJones,
                   [[1.0, 1.5, 0.75], [1.5, 0.5, 1.5], [0.75, 1.5, 0.75]],
                   [[1.0, 0.8, 0.9], [0.8, 0.88, 0.8], [0.9, 0.8, 0.94]])


PRESETS = {"KobAndersen": KobAndersen, "BHHP": BHHP, "JBB": JBB}


def as_matrix(model_matrix):
    """Accept a single Model, or a square nested list, and return a square nested list."""
    if isinstance(model_matrix, Model):
        return [[model_matrix]]
    S = len(model_matrix)
    rows = [list(r) for r in model_matrix]
    if any(len(r) != S for r in rows):
        raise ValueError("model_matrix must be square")
    return rows


def c_matrix(model_matrix):
    rows = as_matrix(model_matrix)
    S = len(rows)
    arr = (_lib.PairModel * (S * S))()
    for i in range(S):
        for j in range(S):
            arr[i * S + j] = rows[i][j].c_struct()
    return arr, S


def model_from_dict(m):
    """get_model(data, i, j) -- src/IO/IO.jl:129-156 (TOML [model."i-j"] tables)."""
    name = m["name"]
    kw = {}
    if m.get("rcut") is not None:
        kw["rcut"] = m["rcut"]
    if name == "SmoothLennardJones":
        return SmoothLennardJones(m["epsilon"], m["sigma"], **kw)
    if name == "LennardJones":
        return LennardJones(m["epsilon"], m["sigma"], shift_potential=m.get("shift_potential", True), **kw)
    if name == "SoftSpheres":
        return SoftSpheres(m["epsilon"], m["sigma"], m["n"], **kw)
    raise ValueError(f"Model {name} is not implemented")
