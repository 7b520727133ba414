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


class EmptyList(NeighbourList):
    """Out of scope on the device; kept so configs that name it fail loudly."""

    def __init__(self, *a, **k):
        raise NotImplementedError("EmptyList has no HIP implementation; the accelerated path is LinkedList")


LIST_TYPES = {"LinkedList": LinkedList, "EmptyList": EmptyList}
