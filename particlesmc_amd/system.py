"""Atoms / System: host mirror of src/atoms.jl over a libpmc_hip.so handle."""
import ctypes as C

import numpy as np

from . import _lib
from .models import as_matrix, c_matrix
from .neighbours import LinkedList


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def fold_back(x, box):
    """fold_back(x, box) = x .- fld.(x, box) .* box -- src/utils.jl:12 (float arguments).

    Julia's float fld is round((x - mod(x, y)) / y); numpy's floor_divide follows
    the same exact-remainder definition (npy_divmod)."""
    x = np.asarray(x, dtype=np.float64)
    box = np.asarray(box, dtype=np.float64)
    return x - np.floor_divide(x, box) * box


class _EnergyRef:
    """system.energy -- a 1-element vector in the reference (src/atoms.jl:22)."""

    def __init__(self, system):
        self._s = system

    def __getitem__(self, k):
        if k not in (0, -1):
            raise IndexError(k)
        e = C.c_double()
        _lib.check(_lib.lib().pmc_get_energy(self._s._h, C.byref(e)), self._s._h)
        return e.value

    def __setitem__(self, k, v):
        if k not in (0, -1):
            raise IndexError(k)
        _lib.check(_lib.lib().pmc_set_energy(self._s._h, float(v)), self._s._h)

    def __len__(self):
        return 1


class SpeciesList:
    """SpeciesList (src/utils.jl:31-49), host view; the device keeps its own copy in step."""

    def __init__(self, species):
        sp = np.asarray(species)
        self.available = sorted(set(int(s) for s in sp))
        self.sp_ids = {s: (np.nonzero(sp == s)[0] + 1) for s in self.available}

    def count(self, s):
        return len(self.sp_ids.get(s, ()))


class Particles:
    """abstract type Particles."""


class Atoms(Particles):
    """struct Atoms (src/atoms.jl:18-30): state lives on one MI355X behind `_h`."""

    def __init__(self, position, species, density, temperature, model_matrix, box, neighbour_list,
                 seed=0, chain=0, device=0):
        pos = np.ascontiguousarray(position, dtype=np.float64)
        if pos.ndim != 2:
            raise ValueError("position must be N x d")
        self.N, self.d = int(pos.shape[0]), int(pos.shape[1])
        sp = np.ascontiguousarray(species, dtype=np.int32)
        if len(sp) != self.N:
            raise AssertionError("length(position) == length(species)")
        self.density = float(density)
        self.model_matrix = as_matrix(model_matrix)
        self.box = np.ascontiguousarray(box, dtype=np.float64)
        self.neighbour_list = neighbour_list
        neighbour_list._system = self
        self.species_list = SpeciesList(sp)
        self.seed, self.chain, self.device = int(seed), int(chain), int(device)
        marr, S = c_matrix(self.model_matrix)
        self._models = marr
        p = _lib.Params()
        p.N, p.d, p.nspecies = self.N, self.d, S
        for a in range(self.d):
            p.box[a] = self.box[a]
            p.cell[a] = neighbour_list.box[a]
            p.ncells[a] = neighbour_list.ncells[a]
        p.temperature, p.density = float(temperature), self.density
        p.models = marr
        p.device, p.chain, p.seed = self.device, self.chain, self.seed
        h = C.c_void_p()
        _lib.check(_lib.lib().pmc_create(C.byref(p), C.byref(h)))
        self._h = h
        if getattr(neighbour_list, "LIST_TYPE", 1) != 1:
            _lib.check(_lib.lib().pmc_set_list_type(self._h, int(neighbour_list.LIST_TYPE), float(getattr(neighbour_list, "dr", 0.0))), self._h)
        _lib.check(_lib.lib().pmc_upload(self._h, _dp(pos), _ip(sp)), self._h)
        self.energy = _EnergyRef(self)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().pmc_destroy(h)
            except Exception:
                pass
            self._h = None

    def __len__(self):                       # Base.length(system) src/ParticlesMC.jl:76
        return self.N

    # --- fields read by IO / callbacks (src/IO/IO.jl:18-23,367) ---------------
    @property
    def position(self):
        pos = np.zeros((self.N, self.d))
        _lib.check(_lib.lib().pmc_download(self._h, _dp(pos), None), self._h)
        return pos

    @position.setter
    def position(self, value):
        pos = np.ascontiguousarray(value, dtype=np.float64)
        _lib.check(_lib.lib().pmc_upload(self._h, _dp(pos), None), self._h)

    def upload(self, position, species=None):
        """Replace positions (and, when given, species: the SpeciesList is then rebuilt as the reference
        constructor builds it, src/utils.jl:36-49) on the device; the neighbour list is rebuilt on next use."""
        pos = np.ascontiguousarray(position, dtype=np.float64)
        if pos.shape != (self.N, self.d):
            raise ValueError("position must be N x d")
        sp = None
        if species is not None:
            sp = np.ascontiguousarray(species, dtype=np.int32)
            if sp.shape != (self.N,):
                raise AssertionError("length(position) == length(species)")
        _lib.check(_lib.lib().pmc_upload(self._h, _dp(pos), None if sp is None else _ip(sp)), self._h)

    @property
    def species(self):
        sp = np.zeros(self.N, dtype=np.int32)
        _lib.check(_lib.lib().pmc_download(self._h, None, _ip(sp)), self._h)
        return sp

    @property
    def temperature(self):
        t = C.c_double()
        _lib.check(_lib.lib().pmc_get_temperature(self._h, C.byref(t)), self._h)
        return t.value

    @temperature.setter
    def temperature(self, T):
        _lib.check(_lib.lib().pmc_set_temperature(self._h, float(T)), self._h)

    # --- LinkedList views -------------------------------------------------------
    def build_neighbour_list(self):
        _lib.check(_lib.lib().pmc_build_cells(self._h), self._h)

    def get_cells(self):
        cs = np.zeros(self.N, dtype=np.int64)
        _lib.check(_lib.lib().pmc_get_cells(self._h, _lp(cs)), self._h)
        return cs

    def get_linked_list(self):
        head = np.zeros(int(np.prod(self.neighbour_list.ncells)), dtype=np.int64)
        lst = np.zeros(self.N, dtype=np.int64)
        _lib.check(_lib.lib().pmc_get_linked_list(self._h, _lp(head), _lp(lst)), self._h)
        return head, lst

    def get_cell_members(self, c):
        """cells[c] of a CellList / VerletList, in the reference's vector order (1-based ids)."""
        out = np.zeros(4096, dtype=np.int64)
        n = C.c_int64()
        _lib.check(_lib.lib().pmc_get_cell_members(self._h, int(c), _lp(out), len(out), C.byref(n)), self._h)
        return out[: n.value].copy()

    def get_verlet_neighbours(self, i):
        """neighbours[i] of a VerletList (1-based ids, vector order)."""
        out = np.zeros(4096, dtype=np.int64)
        n = C.c_int64()
        _lib.check(_lib.lib().pmc_get_verlet_neighbours(self._h, int(i), _lp(out), len(out), C.byref(n)), self._h)
        return out[: n.value].copy()

    def neighbours_of(self, i):
        """collect(neighbour_list(system, i)): the ids the reference's LinkedIterator yields for particle i
        (src/neighbours.jl:316-377; 1-based, includes i itself)."""
        out = np.zeros(1024, dtype=np.int64)
        n = C.c_int64()
        rc = _lib.lib().pmc_neighbours_of(self._h, int(i), _lp(out), len(out), C.byref(n))
        if rc != 0 and n.value > len(out):
            out = np.zeros(n.value, dtype=np.int64)
            rc = _lib.lib().pmc_neighbours_of(self._h, int(i), _lp(out), len(out), C.byref(n))
        _lib.check(rc, self._h)
        return out[: n.value].copy()

    def get_neighbour_cells(self, c):
        out = np.zeros(27, dtype=np.int64)
        n = C.c_int32()
        _lib.check(_lib.lib().pmc_get_neighbour_cells(self._h, int(c), _lp(out), C.byref(n)), self._h)
        return out[: n.value].copy()

    # --- energies -------------------------------------------------------------
    def total_energy(self):
        """Exact fixed-point total (throughput engine); also resets system.energy."""
        e = C.c_double()
        ef = C.c_int64()
        _lib.check(_lib.lib().pmc_total_energy(self._h, C.byref(e), C.byref(ef)), self._h)
        return e.value, ef.value

    def grid(self):
        n = np.zeros(3, dtype=np.int64)
        w = np.zeros(3)
        _lib.check(_lib.lib().pmc_grid(self._h, _lp(n), _dp(w)), self._h)
        return n[: self.d].copy(), w[: self.d].copy()

    def delta_e(self, ids, deltas):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        dl = np.ascontiguousarray(deltas, dtype=np.float64)
        e1 = np.zeros(len(ids))
        e2 = np.zeros(len(ids))
        _lib.check(_lib.lib().pmc_delta_e(self._h, len(ids), _lp(ids), _dp(dl), _dp(e1), _dp(e2)), self._h)
        return e1, e2

    def species_ids(self, sp):
        """species_list.sp_ids[sp] as the device currently holds it (src/utils.jl:31-34)."""
        n = C.c_int64()
        _lib.check(_lib.lib().pmc_species_count(self._h, int(sp), C.byref(n)), self._h)
        ids = np.zeros(n.value, dtype=np.int64)
        if n.value:
            _lib.check(_lib.lib().pmc_get_species_ids(self._h, int(sp), _lp(ids)), self._h)
        return ids

    def _species_ids_device(self, pair):
        return self.species_ids(pair[0]), self.species_ids(pair[1])

    def set_counters(self, step=0, sweep=0):
        _lib.check(_lib.lib().pmc_set_counters(self._h, int(step), int(sweep)), self._h)

    def get_counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().pmc_get_counters(self._h, C.byref(a), C.byref(b)), self._h)
        return a.value, b.value

    # --- schedule parameter of the throughput engine (no reference counterpart) -----------------
    @property
    def rounds(self):
        """m: sweeps per pass = rounds of n_c proposal slots a cell visit runs against its staged neighbourhood."""
        r = C.c_int32()
        _lib.check(_lib.lib().pmc_get_rounds(self._h, C.byref(r)), self._h)
        return r.value

    @rounds.setter
    def rounds(self, m):
        _lib.check(_lib.lib().pmc_set_rounds(self._h, int(m)), self._h)


    # --- checkpoint / restart of a throughput chain (include/pmc.h: pmc_get_images) -----------------
    def get_images(self):
        img = np.zeros((self.N, self.d), dtype=np.int32)
        _lib.check(_lib.lib().pmc_get_images(self._h, _ip(img)), self._h)
        return img

    def set_images(self, img):
        img = np.ascontiguousarray(img, dtype=np.int32).reshape(self.N, self.d)
        _lib.check(_lib.lib().pmc_set_images(self._h, _ip(img)), self._h)

    # --- how PMC_SEQUENTIAL sweeps run (same chain bit for bit; include/pmc.h) --------------------
    def set_sequential_engine(self, engine=0, window=0):
        """engine 0 auto, 1 the one-wavefront literal loop, 2 speculative windows, 3 the chain's state in LDS (small systems;
        any number of chains per launch through mc_sweep_many), 4 a VerletList system's state in LDS (one wavefront resolves, twelve
        evaluate ahead), 5 the same for small EmptyList systems under Displacement-only pools (evaluations repaired term by term);
        window: pending steps per round of engine 2 (0 auto); with engine 5 the wavefronts per chain, 8 or 16 (0 auto)."""
        _lib.check(_lib.lib().pmc_set_sequential_engine(self._h, int(engine), int(window)), self._h)

    def debug_inject(self, what=1):
        """Fault injection for the tests (include/pmc.h: pmc_debug_inject); 1 = the next fused launch is declared starved."""
        _lib.check(_lib.lib().pmc_debug_inject(self._h, int(what)), self._h)

    def verlet_info(self):
        """(batches, steps wavefront 0 ran alone, list rebuilds, steps the engine ran) of the last PMC_SEQUENTIAL call that ran
        engine 4 (VerletList state in LDS)."""
        out = (C.c_int64 * 4)()
        _lib.check(_lib.lib().pmc_verlet_info(self._h, out), self._h)
        return tuple(int(x) for x in out)

    def sequential_info(self):
        """(rounds, step evaluations, window, steps of that call folded into system.energy) of the last PMC_SEQUENTIAL call that
        ran engine 2."""
        out = (C.c_int64 * 4)()
        _lib.check(_lib.lib().pmc_sequential_info(self._h, out), self._h)
        return tuple(int(v) for v in out)


def get_position(system, i):                 # src/ParticlesMC.jl:41
    return system.position[i - 1]


def get_species(system, i):                  # src/ParticlesMC.jl:48
    return int(system.species[i - 1])


def get_model(system, i, j):                 # src/ParticlesMC.jl:55
    sp = system.species
    return system.model_matrix[sp[i - 1] - 1][sp[j - 1] - 1]


def get_box(system):                         # src/ParticlesMC.jl:62
    return system.box


def get_neighbour_list(system):              # src/ParticlesMC.jl:69
    return system.neighbour_list


def compute_energy_particle(system, i, neighbour_list=None):
    """compute_energy_particle(system, i[, nl]) -- src/atoms.jl:81-88 (reference summation order)."""
    if np.ndim(i) > 0:
        return np.array([compute_energy_particle(system, int(k)) for k in i])
    e = C.c_double()
    _lib.check(_lib.lib().pmc_energy_particle(system._h, int(i), C.byref(e)), system._h)
    return e.value


def compute_energy_particles(system):
    e = np.zeros(system.N)
    _lib.check(_lib.lib().pmc_energy_particles(system._h, _dp(e)), system._h)
    return e


def compute_energy_particles_exact(system):
    """All per-particle energies in one launch of the throughput engine (exact fixed-point sums,
    ~1e-11 relative from the reference-order values)."""
    e = np.zeros(system.N)
    _lib.check(_lib.lib().pmc_energy_particles_exact(system._h, _dp(e)), system._h)
    return e


def box_from_density(N, density, d):
    b = np.zeros(3)
    _lib.check(_lib.lib().pmc_box_from_density(int(N), float(density), int(d), _dp(b)))
    return b[:d].copy()


def System(position, species, density, temperature, model_matrix, list_type=LinkedList, list_parameters=None,
           seed=0, chain=0, device=0):
    """System(position, species, density, temperature, model_matrix; list_type) -- src/atoms.jl:40-58."""
    pos = np.ascontiguousarray(position, dtype=np.float64)
    N, d = pos.shape
    box = box_from_density(N, density, d)                                   # atoms.jl:45
    maxcut = max(m.rcut for row in as_matrix(model_matrix) for m in row)    # atoms.jl:46
    neighbour_list = list_type(box, maxcut, N, list_parameters=list_parameters)   # atoms.jl:47
    system = Atoms(pos, species, density, temperature, model_matrix, box, neighbour_list,
                   seed=seed, chain=chain, device=device)
    e = C.c_double()
    rc = _lib.lib().pmc_init_energy(system._h, C.byref(e))                  # atoms.jl:50-56
    if rc == _lib.PMC_ERR_NONFINITE:
        raise ValueError("Initial configuration has infinite or NaN energy.")
    _lib.check(rc, system._h)
    return system
