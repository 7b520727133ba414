"""Neighbour lists: host mirror of the LinkedList plug-in protocol (src/neighbours.jl:224-377)."""
import ctypes as C

import numpy as np

from . import _lib


class NeighbourList:
    """abstract type NeighbourList (src/neighbours.jl:3)."""


class LinkedList(NeighbourList):
    """LinkedList(box, rcut, N) -- src/neighbours.jl:236-244.

    Holds the host-computed geometry (`box` = cell size, `ncells`, exactly as the
    reference names its fields); head/list/cs live on the device and are read
    back on demand through the owning system.
    """

    def __init__(self, box, rcut, N, list_parameters=None):
        box = np.ascontiguousarray(box, dtype=np.float64)
        d = len(box)
        cell = np.zeros(3)
        ncells = np.zeros(3, dtype=np.int64)
        rc = _lib.lib().pmc_linked_list_geometry(d, box.ctypes.data_as(C.POINTER(C.c_double)), float(rcut),
                                                 cell.ctypes.data_as(C.POINTER(C.c_double)),
                                                 ncells.ctypes.data_as(C.POINTER(C.c_int64)))
        _lib.check(rc)
        self.box = cell[:d].copy()          # field name as in the reference struct: cell dimensions
        self.ncells = tuple(int(v) for v in ncells[:d])
        self.rcut = float(rcut)
        self.N = int(N)
        self._system = None

    # device-resident arrays, reference numbering (1-based, -1 = empty head)
    @property
    def cs(self):
        return self._system.get_cells()

    @property
    def head(self):
        return self._system.get_linked_list()[0]

    @property
    def list(self):
        return self._system.get_linked_list()[1]

    def neighbour_cells(self, c):
        return self._system.get_neighbour_cells(c)

    def __call__(self, system, i):
        """`linked_list(system, i)` (src/neighbours.jl:371-377): the neighbours of particle i in LinkedIterator order."""
        return system.neighbours_of(i)


class CellList(LinkedList):
    """CellList(box, rcut, N) -- src/neighbours.jl:61-217: per-cell vectors of ids (`cells`), iterated in vector
    order; moves filter! the old cell and push! into the new one.  Device-resident like the LinkedList."""
    LIST_TYPE = 2

    def __init__(self, box, rcut, N, list_parameters=None):
        box = np.ascontiguousarray(box, dtype=np.float64)
        d = len(box)
        cell = np.zeros(3)
        ncells = np.zeros(3, dtype=np.int64)
        _lib.check(_lib.lib().pmc_cell_list_geometry(d, box.ctypes.data_as(C.POINTER(C.c_double)), float(rcut),
                                                     cell.ctypes.data_as(C.POINTER(C.c_double)),
                                                     ncells.ctypes.data_as(C.POINTER(C.c_int64))))
        self.box = cell[:d].copy()
        self.ncells = tuple(int(v) for v in ncells[:d])
        self.rcut = float(rcut)
        self.N = int(N)
        self._system = None

    @property
    def cells(self):
        return [self._system.get_cell_members(c) for c in range(1, int(np.prod(self.ncells)) + 1)]

    @property
    def head(self):
        raise AttributeError("CellList has no head/list (src/neighbours.jl:61-67)")

    list = head


class VerletList(CellList):
    """VerletList(box, rcut, N; list_parameters = Dict("dr" => ...)) -- src/neighbours.jl:387-543: per-particle
    vectors of the ids within rcut + dr, rebuilt for a particle once it has moved dr / 2 since its last update."""
    LIST_TYPE = 3

    def __init__(self, box, rcut, N, list_parameters=None):
        if list_parameters is None or "dr" not in list_parameters:
            raise ValueError("Verlet list must have dr as a parameter")          # src/neighbours.jl:401-403
        self.dr = float(list_parameters["dr"])
        box = np.ascontiguousarray(box, dtype=np.float64)
        d = len(box)
        cell = np.zeros(3)
        ncells = np.zeros(3, dtype=np.int64)
        _lib.check(_lib.lib().pmc_verlet_list_geometry(d, box.ctypes.data_as(C.POINTER(C.c_double)), float(rcut), self.dr,
                                                       cell.ctypes.data_as(C.POINTER(C.c_double)),
                                                       ncells.ctypes.data_as(C.POINTER(C.c_int64))))
        self.box = cell[:d].copy()
        self.ncells = tuple(int(v) for v in ncells[:d])
        self.rcut = float(rcut)
        self.dr2o4 = (self.dr ** 2) / 4
        self.N = int(N)
        self._system = None

    @property
    def neighbours(self):
        return [self._system.get_verlet_neighbours(i) for i in range(1, self.N + 1)]


class EmptyList(NeighbourList):
    """EmptyList(box, rcut, N) -- src/neighbours.jl:13-44: every particle is a neighbour (j in 1:N).  The device
    sums all N pair terms in that order; meant for small systems and for the reference's cross-list tests."""
    LIST_TYPE = 0

    def __init__(self, box, rcut, N, list_parameters=None):
        box = np.ascontiguousarray(box, dtype=np.float64)
        self.box = box.copy()                 # one cell spanning the box (the device still reports cs = 1)
        self.ncells = tuple(1 for _ in box)
        self.rcut = float(rcut)
        self.N = int(N)
        self._system = None


LIST_TYPES = {"LinkedList": LinkedList, "EmptyList": EmptyList, "CellList": CellList, "VerletList": VerletList}
