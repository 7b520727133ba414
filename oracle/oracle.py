"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``particlesmc_amd``) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpmc_oracle.so")

LJ, SOFT_INT, SOFT_FLOAT, SMOOTH_LJ = 0, 1, 2, 3
EMPTYLIST, LINKEDLIST, CELLLIST, VERLETLIST = 0, 1, 2, 3
MOVE_DISPLACEMENT, MOVE_SWAP = 0, 1


class Model(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n", C.c_int32),
        ("eps", C.c_double), ("eps4", C.c_double), ("sigma", C.c_double), ("sigma2", C.c_double),
        ("rcut", C.c_double), ("rcut2", C.c_double), ("shift", C.c_double),
        ("ndiv2_i", C.c_int64), ("ndiv2_f", C.c_double),
        ("C0", C.c_double), ("C2_s2", C.c_double), ("C4_s4", C.c_double),
    ]


class Move(C.Structure):
    _fields_ = [("kind", C.c_int32), ("probability", C.c_double), ("sigma", C.c_double),
                ("species", C.c_int32 * 2)]


class Stats(C.Structure):
    _fields_ = [("attempted", C.c_int64 * 8), ("accepted", C.c_int64 * 8)]


def build(force=False):
    src = os.path.join(_HERE, "pmc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpmc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    lp = C.POINTER(C.c_int64)
    vp = C.c_void_p
    L.orc_lennard_jones.restype = C.c_double
    L.orc_lennard_jones.argtypes = [C.c_double] * 3
    L.orc_pow_int.restype = C.c_double
    L.orc_pow_int.argtypes = [C.c_double, C.c_int64]
    L.orc_model_lj.argtypes = [C.POINTER(Model), C.c_double, C.c_double, C.c_double, C.c_int]
    L.orc_model_soft.argtypes = [C.POINTER(Model), C.c_double, C.c_double, C.c_int, C.c_double]
    L.orc_model_smooth_lj.argtypes = [C.POINTER(Model), C.c_double, C.c_double, C.c_double]
    L.orc_potential.restype = C.c_double
    L.orc_potential.argtypes = [C.c_double, C.POINTER(Model)]
    L.orc_jl_fld.restype = C.c_double
    L.orc_jl_fld.argtypes = [C.c_double, C.c_double]
    L.orc_jl_fld_i.restype = C.c_int64
    L.orc_jl_fld_i.argtypes = [C.c_int64, C.c_int64]
    L.orc_system_create.restype = vp
    L.orc_system_create.argtypes = [C.c_int64, C.c_int32, dp, ip, C.c_double, C.c_double, C.c_int32,
                                    C.POINTER(Model), C.c_int32, ip]
    L.orc_system_create_list.restype = vp
    L.orc_system_create_list.argtypes = [C.c_int64, C.c_int32, dp, ip, C.c_double, C.c_double, C.c_int32,
                                         C.POINTER(Model), C.c_int32, C.c_double, C.POINTER(C.c_int32)]
    L.orc_system_cell_members.restype = C.c_int64
    L.orc_system_cell_members.argtypes = [vp, C.c_int64, lp, C.c_int64]
    L.orc_system_destroy.argtypes = [vp]
    L.orc_system_energy.restype = C.c_double
    L.orc_system_energy.argtypes = [vp]
    L.orc_system_set_energy.argtypes = [vp, C.c_double]
    L.orc_system_set_temperature.argtypes = [vp, C.c_double]
    L.orc_system_box.argtypes = [vp, dp]
    L.orc_system_cell_geometry.argtypes = [vp, dp, lp]
    L.orc_system_get_cells.argtypes = [vp, lp]
    L.orc_system_get_linked_list.argtypes = [vp, lp, lp]
    L.orc_system_neighbour_cells.restype = C.c_int64
    L.orc_system_neighbour_cells.argtypes = [vp, C.c_int64, lp]
    L.orc_system_get_positions.argtypes = [vp, dp]
    L.orc_system_get_species.argtypes = [vp, ip]
    L.orc_energy_particle.restype = C.c_double
    L.orc_energy_particle.argtypes = [vp, C.c_int64]
    L.orc_energy_particles.argtypes = [vp, dp]
    L.orc_total_energy.restype = C.c_double
    L.orc_total_energy.argtypes = [vp]
    L.orc_neighbours_of.restype = C.c_int64
    L.orc_neighbours_of.argtypes = [vp, C.c_int64, lp, C.c_int64]
    L.orc_displacement_perform.argtypes = [vp, C.c_int64, dp, dp, dp, dp]
    L.orc_displacement_revert.argtypes = [vp, C.c_int64, dp, C.c_double]
    L.orc_swap_perform.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, dp, dp, dp]
    L.orc_swap_revert.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_double]
    L.orc_species_count.restype = C.c_int64
    L.orc_species_count.argtypes = [vp, C.c_int32]
    L.orc_species_ids.argtypes = [vp, C.c_int32, lp]
    L.orc_species_heads.argtypes = [vp, lp]
    L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_det_log.restype = C.c_double
    L.orc_det_log.argtypes = [C.c_double]
    L.orc_det_exp.restype = C.c_double
    L.orc_det_exp.argtypes = [C.c_double]
    L.orc_energy_bias.restype = C.c_int
    L.orc_energy_bias.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                  lp, lp, dp]
    L.orc_det_sincos2pi.argtypes = [C.c_uint32, dp, dp]
    L.orc_box_muller.argtypes = [C.c_uint32, C.c_uint32, dp, dp]
    L.orc_uniform53.restype = C.c_double
    L.orc_uniform53.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_mc_steps_sequential.argtypes = [vp, C.POINTER(Move), C.c_int32, C.c_uint64, C.c_uint32,
                                          C.c_uint64, C.c_int64, C.POINTER(Stats)]
    L.orc_cb_create.restype = vp
    L.orc_cb_create.argtypes = [C.c_int64, C.c_int32, dp, ip, dp, C.c_double, C.c_int32,
                                C.POINTER(Model), C.c_uint64, C.c_uint32, ip]
    L.orc_cb_destroy.argtypes = [vp]
    L.orc_cb_grid.argtypes = [vp, lp, dp]
    L.orc_cb_sweeps.argtypes = [vp, C.POINTER(Move), C.c_int32, C.c_uint64, C.c_int64, C.POINTER(Stats)]
    L.orc_cb_sweeps_rounds.argtypes = [vp, C.POINTER(Move), C.c_int32, C.c_uint64, C.c_int64, C.c_int32, C.POINTER(Stats)]
    L.orc_cb_get_positions.argtypes = [vp, dp]
    L.orc_cb_get_species.argtypes = [vp, ip]
    L.orc_cb_energy_fix.restype = C.c_int64
    L.orc_cb_energy_fix.argtypes = [vp]
    L.orc_cb_total_energy_fix.restype = C.c_int64
    L.orc_cb_total_energy_fix.argtypes = [vp]
    L.orc_cb_set_temperature.argtypes = [vp, C.c_double]
    L.orc_cb_energy_at.restype = C.c_double
    L.orc_cb_energy_at.argtypes = [vp, C.c_int64, dp, C.c_int32]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


# ---- model constructors ------------------------------------------------------
def lennard_jones(eps, sigma, rcut=None, shift_potential=True):
    m = Model()
    lib().orc_model_lj(C.byref(m), eps, sigma, 2.5 * sigma if rcut is None else rcut, int(shift_potential))
    return m


def soft_spheres(eps, sigma, n, rcut=None):
    m = Model()
    lib().orc_model_soft(C.byref(m), eps, sigma, n, 2.5 * sigma if rcut is None else rcut)
    return m


def smooth_lennard_jones(eps, sigma, rcut=None):
    m = Model()
    lib().orc_model_smooth_lj(C.byref(m), eps, sigma, 2.5 * sigma if rcut is None else rcut)
    return m


def model_matrix(ctor, eps, sigma, **kw):
    """Row-major S*S array of models from eps/sigma matrices (models.jl presets)."""
    S = len(eps)
    arr = (Model * (S * S))()
    for i in range(S):
        for j in range(S):
            arr[i * S + j] = ctor(eps[i][j], sigma[i][j], **kw)
    return arr, S


def kob_andersen():   # models.jl:125-133
    return model_matrix(lennard_jones, [[1.0, 1.5], [1.5, 0.5]], [[1.0, 0.8], [0.8, 0.88]])


def bhhp(n=12):       # models.jl:76-84
    return model_matrix(lambda e, s: soft_spheres(e, s, n), [[1.0, 1.0], [1.0, 1.0]], [[1.0, 1.2], [1.2, 1.4]])


def jbb():            # models.jl:169-179
    return model_matrix(smooth_lennard_jones,
                        [[1.0, 1.5, 0.75], [1.5, 0.5, 1.5], [0.75, 1.5, 0.75]],
                        [[1.0, 0.8, 0.9], [0.8, 0.88, 0.8], [0.9, 0.8, 0.94]])


def single(model):
    arr = (Model * 1)()
    arr[0] = model
    return arr, 1


def make_pool(moves):
    """moves: list of dicts {kind, probability, sigma | species}."""
    arr = (Move * len(moves))()
    for k, mv in enumerate(moves):
        arr[k].kind = MOVE_DISPLACEMENT if mv["kind"] == "Displacement" else MOVE_SWAP
        arr[k].probability = mv.get("probability", 1.0)
        arr[k].sigma = mv.get("sigma", 0.0)
        sp = mv.get("species", (0, 0))
        arr[k].species[0], arr[k].species[1] = sp[0], sp[1]
    return arr


class System:
    """orc_system wrapper: the reference-faithful sequential path."""

    def __init__(self, position, species, density, temperature, models, list_type=LINKEDLIST, dr=0.0):
        marr, S = models
        self.pos0 = np.ascontiguousarray(position, dtype=np.float64)
        self.N, self.d = self.pos0.shape
        sp = np.ascontiguousarray(species, dtype=np.int32)
        err = C.c_int32(0)
        self.h = lib().orc_system_create_list(self.N, self.d, _dp(self.pos0), _ip(sp), density, temperature,
                                              S, marr, list_type, dr, C.byref(err))
        self.err = err.value
        if not self.h:
            raise ValueError({1: "Initial configuration has infinite or NaN energy.",
                              2: "bad arguments"}.get(err.value, "error"))
        self.models = models
        self.list_type = list_type

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_system_destroy(self.h)
            self.h = None

    def cell_members(self, c):
        """cells[c] of a CellList / VerletList (1-based c), in the reference's vector order."""
        out = np.zeros(4096, dtype=np.int64)
        n = lib().orc_system_cell_members(self.h, c, out.ctypes.data_as(C.POINTER(C.c_int64)), len(out))
        return out[:n].copy()

    @property
    def energy(self):
        return lib().orc_system_energy(self.h)

    @property
    def box(self):
        b = np.zeros(3)
        lib().orc_system_box(self.h, _dp(b))
        return b[: self.d].copy()

    def cell_geometry(self):
        c = np.zeros(3)
        n = np.zeros(3, dtype=np.int64)
        lib().orc_system_cell_geometry(self.h, _dp(c), _lp(n))
        return c[: self.d].copy(), n[: self.d].copy()

    def cells(self):
        cs = np.zeros(self.N, dtype=np.int64)
        lib().orc_system_get_cells(self.h, _lp(cs))
        return cs

    def linked_list(self):
        _, n = self.cell_geometry()
        head = np.zeros(int(np.prod(n)), dtype=np.int64)
        lst = np.zeros(self.N, dtype=np.int64)
        lib().orc_system_get_linked_list(self.h, _lp(head), _lp(lst))
        return head, lst

    def neighbour_cells(self, c):
        out = np.zeros(27, dtype=np.int64)
        n = lib().orc_system_neighbour_cells(self.h, c, _lp(out))
        return out[:n].copy()

    def positions(self):
        p = np.zeros((self.N, self.d))
        lib().orc_system_get_positions(self.h, _dp(p))
        return p

    def species(self):
        s = np.zeros(self.N, dtype=np.int32)
        lib().orc_system_get_species(self.h, _ip(s))
        return s

    def energy_particle(self, i):
        return lib().orc_energy_particle(self.h, i)

    def energy_particles(self):
        e = np.zeros(self.N)
        lib().orc_energy_particles(self.h, _dp(e))
        return e

    def total_energy(self):
        return lib().orc_total_energy(self.h)

    def neighbours_of(self, i):
        out = np.zeros(self.N, dtype=np.int64)
        n = lib().orc_neighbours_of(self.h, i, _lp(out), self.N)
        return out[:n].copy()

    def displacement_perform(self, i, delta):
        d = np.ascontiguousarray(delta, dtype=np.float64)
        e1, e2, de = C.c_double(), C.c_double(), C.c_double()
        lib().orc_displacement_perform(self.h, i, _dp(d), C.byref(e1), C.byref(e2), C.byref(de))
        return e1.value, e2.value, de.value

    def displacement_revert(self, i, minus_delta, de):
        d = np.ascontiguousarray(minus_delta, dtype=np.float64)
        lib().orc_displacement_revert(self.h, i, _dp(d), de)

    def swap_perform(self, i, j, sp1, sp2):
        e1, e2, de = C.c_double(), C.c_double(), C.c_double()
        lib().orc_swap_perform(self.h, i, j, sp1, sp2, C.byref(e1), C.byref(e2), C.byref(de))
        return e1.value, e2.value, de.value

    def swap_revert(self, i, j, sp1, sp2, de):
        lib().orc_swap_revert(self.h, i, j, sp1, sp2, de)

    def species_ids(self, sp):
        n = lib().orc_species_count(self.h, sp)
        ids = np.zeros(n, dtype=np.int64)
        lib().orc_species_ids(self.h, sp, _lp(ids))
        return ids

    def species_heads(self):
        h = np.zeros(self.N, dtype=np.int64)
        lib().orc_species_heads(self.h, _lp(h))
        return h

    def set_temperature(self, T):
        lib().orc_system_set_temperature(self.h, C.c_double(T))

    def energy_bias(self, species, theta, u=(0.0, 0.0), given=(0, 0)):
        """EnergyBias (moves.jl:246-280): (i, j, log q) sampled with the uniforms u, or log q of a given pair."""
        i, j, lq = C.c_int64(), C.c_int64(), C.c_double()
        rc = lib().orc_energy_bias(self.h, species[0], species[1], theta[0], theta[1], u[0], u[1], given[0], given[1],
                                   C.byref(i), C.byref(j), C.byref(lq))
        if rc:
            raise ValueError("a species of the pair has no particles")
        return i.value, j.value, lq.value

    def mc_steps(self, pool, seed, chain, step0, nsteps):
        st = Stats()
        lib().orc_mc_steps_sequential(self.h, pool, len(pool), seed, chain, step0, nsteps, C.byref(st))
        return np.array(st.attempted[: len(pool)]), np.array(st.accepted[: len(pool)])


class Checkerboard:
    """orc_cb wrapper: restatement of the parallel (checkerboard) schedule."""

    def __init__(self, position, species, box, temperature, models, seed, chain=0):
        marr, S = models
        p = np.ascontiguousarray(position, dtype=np.float64)
        self.N, self.d = p.shape
        sp = np.ascontiguousarray(species, dtype=np.int32)
        b = np.zeros(3)
        b[: self.d] = box
        err = C.c_int32(0)
        self.h = lib().orc_cb_create(self.N, self.d, _dp(p), _ip(sp), _dp(b), temperature, S, marr,
                                     seed, chain, C.byref(err))
        if not self.h:
            raise ValueError("checkerboard grid needs >= 4 cells per side (or a single-cell box: 2 ... 4 cutoffs per side in every dimension, N <= 1024)")
        self.models = models

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cb_destroy(self.h)
            self.h = None

    def grid(self):
        n = np.zeros(3, dtype=np.int64)
        w = np.zeros(3)
        lib().orc_cb_grid(self.h, _lp(n), _dp(w))
        return n[: self.d].copy(), w[: self.d].copy()

    def sweeps(self, pool, sweep0, nsweeps, rounds=1):
        """`rounds` = m of the schedule: passes of m sweeps share one grid shift and one staged neighbourhood per cell."""
        st = Stats()
        lib().orc_cb_sweeps_rounds(self.h, pool, len(pool), sweep0, nsweeps, rounds, C.byref(st))
        return np.array(st.attempted[: len(pool)]), np.array(st.accepted[: len(pool)])

    def positions(self):
        p = np.zeros((self.N, self.d))
        lib().orc_cb_get_positions(self.h, _dp(p))
        return p

    def species(self):
        s = np.zeros(self.N, dtype=np.int32)
        lib().orc_cb_get_species(self.h, _ip(s))
        return s

    @property
    def energy_fix(self):
        return lib().orc_cb_energy_fix(self.h)

    def total_energy_fix(self):
        return lib().orc_cb_total_energy_fix(self.h)

    def set_temperature(self, T):
        lib().orc_cb_set_temperature(self.h, T)

    def energy_at(self, i, x, sp):
        xx = np.ascontiguousarray(x, dtype=np.float64)
        return lib().orc_cb_energy_at(self.h, i, _dp(xx), sp)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return list(o)


def box_muller(w1, w2):
    a, b = C.c_double(), C.c_double()
    lib().orc_box_muller(w1, w2, C.byref(a), C.byref(b))
    return a.value, b.value


def det_sincos2pi(w):
    a, b = C.c_double(), C.c_double()
    lib().orc_det_sincos2pi(w, C.byref(a), C.byref(b))
    return a.value, b.value


TAG_PT = 6


def pt_decide(seed, rnd, energies, ladder, label):
    """CPU restatement of one replica-exchange round (the reference has none: README.md:33).

    label[r] = index of the temperature replica r holds.  Round `rnd` pairs the adjacent temperatures
    (t, t + 1) for t = rnd & 1, +2, ...; the pair swaps iff ln u < (1/T_t - 1/T_{t+1}) (E_a - E_b) with
    u the 53-bit uniform of Philox(ctr = (t, 0, rnd_lo, rnd_hi | TAG_PT << 24), key = seed).
    Returns (new label, number of accepted swaps)."""
    L = lib()
    n = len(label)
    lab = [int(v) for v in label]
    holder = [-1] * n
    for r, t in enumerate(lab):
        holder[t] = r
    acc = 0
    for t in range(int(rnd) & 1, n - 1, 2):
        ra, rb = holder[t], holder[t + 1]
        arg = (1.0 / float(ladder[t]) - 1.0 / float(ladder[t + 1])) * (float(energies[ra]) - float(energies[rb]))
        w = philox((t, 0, int(rnd) & 0xffffffff, ((int(rnd) >> 32) & 0xffffff) | (TAG_PT << 24)),
                   (int(seed) & 0xffffffff, (int(seed) >> 32) & 0xffffffff))
        lnu = L.orc_det_log(L.orc_uniform53(w[0], w[1]))
        if lnu < arg:
            lab[ra], lab[rb] = t + 1, t
            acc += 1
    return lab, acc
