"""ctypes binding of libpmc_hip.so (include/pmc.h).

There is no CPU fallback: if the shared library is missing this module raises,
and every compute entry point fails loudly when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PMC_LIB") or os.path.join(_HERE, "csrc", "libpmc_hip.so")   # PMC_LIB: debug builds

PMC_OK, PMC_ERR_BADARG, PMC_ERR_HIP, PMC_ERR_NONFINITE, PMC_ERR_UNSUPPORTED, PMC_ERR_STATE = 0, -1, -2, -3, -4, -5
MODEL_LJ, MODEL_SOFT_INT, MODEL_SOFT_FLOAT, MODEL_SMOOTH_LJ = 0, 1, 2, 3
MOVE_DISPLACEMENT, MOVE_SWAP = 0, 1
SEQUENTIAL, CHECKERBOARD = 0, 1
MAX_MOVES = 8


class PairModel(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n", C.c_int32),
        ("eps", C.c_double), ("eps4", C.c_double), ("sigma", C.c_double), ("sigma2", C.c_double),
        ("rcut", C.c_double), ("rcut2", C.c_double), ("shift", C.c_double),
        ("ndiv2_i", C.c_int64), ("ndiv2_f", C.c_double),
        ("C0", C.c_double), ("C2_s2", C.c_double), ("C4_s4", C.c_double),
    ]


class Params(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("d", C.c_int32), ("nspecies", C.c_int32),
        ("box", C.c_double * 3), ("cell", C.c_double * 3), ("ncells", C.c_int64 * 3),
        ("temperature", C.c_double), ("density", C.c_double),
        ("models", C.POINTER(PairModel)),
        ("device", C.c_int32), ("chain", C.c_uint32), ("seed", C.c_uint64),
    ]


class Move(C.Structure):
    _fields_ = [("kind", C.c_int32), ("species", C.c_int32 * 2), ("probability", C.c_double),
                ("sigma", C.c_double)]


class Stats(C.Structure):
    _fields_ = [
        ("attempted", C.c_int64 * MAX_MOVES), ("accepted", C.c_int64 * MAX_MOVES),
        ("energy", C.c_double), ("energy_fix", C.c_int64),
        ("kernel_ms", C.c_double), ("sweep_kernel_ms", C.c_double), ("sweep_kernel_launches", C.c_int64),
        ("pair_tests", C.c_int64), ("pairs_in_range", C.c_int64), ("null_moves", C.c_int64),
        ("dense_evals", C.c_int64), ("staged_candidates", C.c_int64), ("written_back", C.c_int64),
    ]


# every symbol include/pmc.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_up = C.POINTER(C.c_uint32)
_h = C.c_void_p
SYMBOLS = {
    "pmc_create": (C.c_int, [C.POINTER(Params), C.POINTER(_h)]),
    "pmc_destroy": (C.c_int, [_h]),
    "pmc_last_error": (C.c_char_p, [_h]),
    "pmc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pmc_model_lennard_jones": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int32, C.POINTER(PairModel)]),
    "pmc_model_soft_spheres": (C.c_int, [C.c_double, C.c_double, C.c_int32, C.c_double, C.POINTER(PairModel)]),
    "pmc_model_smooth_lennard_jones": (C.c_int, [C.c_double, C.c_double, C.c_double, C.POINTER(PairModel)]),
    "pmc_potential": (C.c_int, [C.c_double, C.POINTER(PairModel), _dp]),
    "pmc_box_from_density": (C.c_int, [C.c_int64, C.c_double, C.c_int32, _dp]),
    "pmc_linked_list_geometry": (C.c_int, [C.c_int32, _dp, C.c_double, _dp, _lp]),
    "pmc_cell_list_geometry": (C.c_int, [C.c_int32, _dp, C.c_double, _dp, _lp]),
    "pmc_verlet_list_geometry": (C.c_int, [C.c_int32, _dp, C.c_double, C.c_double, _dp, _lp]),
    "pmc_set_list_type": (C.c_int, [_h, C.c_int32, C.c_double]),
    "pmc_get_cell_members": (C.c_int, [_h, C.c_int64, _lp, C.c_int64, _lp]),
    "pmc_get_verlet_neighbours": (C.c_int, [_h, C.c_int64, _lp, C.c_int64, _lp]),
    "pmc_upload": (C.c_int, [_h, _dp, _ip]),
    "pmc_download": (C.c_int, [_h, _dp, _ip]),
    "pmc_get_images": (C.c_int, [_h, _ip]),
    "pmc_set_images": (C.c_int, [_h, _ip]),
    "pmc_set_temperature": (C.c_int, [_h, C.c_double]),
    "pmc_get_temperature": (C.c_int, [_h, _dp]),
    "pmc_set_energy": (C.c_int, [_h, C.c_double]),
    "pmc_get_energy": (C.c_int, [_h, _dp]),
    "pmc_set_counters": (C.c_int, [_h, C.c_uint64, C.c_uint64]),
    "pmc_get_counters": (C.c_int, [_h, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pmc_build_cells": (C.c_int, [_h]),
    "pmc_get_cells": (C.c_int, [_h, _lp]),
    "pmc_get_linked_list": (C.c_int, [_h, _lp, _lp]),
    "pmc_get_neighbour_cells": (C.c_int, [_h, C.c_int64, _lp, _ip]),
    "pmc_neighbours_of": (C.c_int, [_h, C.c_int64, _lp, C.c_int64, _lp]),
    "pmc_energy_particle": (C.c_int, [_h, C.c_int64, _dp]),
    "pmc_energy_particles": (C.c_int, [_h, _dp]),
    "pmc_init_energy": (C.c_int, [_h, _dp]),
    "pmc_species_count": (C.c_int, [_h, C.c_int32, _lp]),
    "pmc_get_species_ids": (C.c_int, [_h, C.c_int32, _lp]),
    "pmc_displacement_perform": (C.c_int, [_h, C.c_int64, _dp, _dp, _dp, _dp]),
    "pmc_displacement_revert": (C.c_int, [_h, C.c_int64, _dp, C.c_double]),
    "pmc_swap_perform": (C.c_int, [_h, C.c_int64, C.c_int64, _dp, _dp, _dp]),
    "pmc_swap_revert": (C.c_int, [_h, C.c_int64, C.c_int64, C.c_double]),
    "pmc_energy_bias_sample": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _lp, _lp, _dp]),
    "pmc_energy_bias_logq": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_int64, _dp]),
    "pmc_exp": (C.c_int, [C.c_double, _dp]),
    "pmc_total_energy": (C.c_int, [_h, _dp, _lp]),
    "pmc_energy_particles_exact": (C.c_int, [_h, _dp]),
    "pmc_delta_e": (C.c_int, [_h, C.c_int64, _lp, _dp, _dp, _dp]),
    "pmc_grid": (C.c_int, [_h, _lp, _dp]),
    "pmc_debug_inject": (C.c_int, [_h, C.c_int32]),
    "pmc_set_rounds": (C.c_int, [_h, C.c_int32]),
    "pmc_get_rounds": (C.c_int, [_h, C.POINTER(C.c_int32)]),
    "pmc_set_sequential_engine": (C.c_int, [_h, C.c_int32, C.c_int32]),
    "pmc_sequential_info": (C.c_int, [_h, _lp]),
    "pmc_verlet_info": (C.c_int, [_h, _lp]),
    "pmc_sweep": (C.c_int, [_h, C.c_int32, C.POINTER(Move), C.c_int32, C.c_int64, C.POINTER(Stats)]),
    "pmc_sweep_begin": (C.c_int, [_h, C.c_int32, C.POINTER(Move), C.c_int32, C.c_int64]),
    "pmc_sweep_end": (C.c_int, [_h, C.POINTER(Stats)]),
    "pmc_sweep_many": (C.c_int, [C.POINTER(_h), C.c_int32, C.c_int32, C.POINTER(Move), C.c_int32, C.c_int64, C.POINTER(Stats)]),
    "pmc_pt_decide": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int32, _dp, _dp, _ip, _ip]),
    "pmc_replica_exchange": (C.c_int, [C.POINTER(_h), C.c_int32, _dp, C.c_uint64, C.c_uint64, _ip, _ip]),
    "pmc_replica_exchange_rccl": (C.c_int, [_h, C.c_void_p, C.c_int32, C.c_int32, _dp, C.c_uint64, C.c_uint64, _ip, _ip]),
    "pmc_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "pmc_rccl_comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "pmc_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "pmc_philox4x32": (C.c_int, [_up, _up, _up]),
    "pmc_box_muller": (C.c_int, [C.c_uint32, C.c_uint32, _dp, _dp]),
    "pmc_log_uniform": (C.c_int, [C.c_uint32, C.c_uint32, _dp]),
    "pmc_sequential_words": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, _up, _up]),
    "pmc_debug_rng": (C.c_int, [_h, C.c_int64, _up, _up, _dp, _dp, _dp]),
    "pmc_debug_div": (C.c_int, [_h, C.c_int64, _dp, _dp, _dp, _dp]),
    "pmc_version": (C.c_char_p, []),
}

_lib = None


class PMCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpmc_hip: {msg} (status {code})")
        self.code = code


def lib():
    """Load libpmc_hip.so; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C particlesmc_amd/csrc`. particlesmc_amd has no CPU fallback."
        )
    L = C.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc, handle=None):
    if rc != PMC_OK:
        msg = lib().pmc_last_error(handle)
        raise PMCError(rc, msg.decode() if msg else "error")


def device_count():
    n = C.c_int(0)
    rc = lib().pmc_device_count(C.byref(n))
    return n.value if rc == PMC_OK else 0
