"""CPU: the C-ABI library loads, exports every symbol include/pmc.h declares, its struct layouts
match the header, and its host-side constructors agree bit for bit with the oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pmc.h")


@pytest.fixture(scope="module")
def pmc(built):
    import particlesmc_amd
    return particlesmc_amd


def _declared():
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(pmc_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(pmc):
    from particlesmc_amd import _lib
    L = C.CDLL(_lib.SO_PATH)
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pmc.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes binding"
    assert set(_lib.SYMBOLS) <= set(names)


def test_struct_layouts_match_header(pmc, tmp_path):
    from particlesmc_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pmc.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(pmc_pair_model), sizeof(pmc_params),'
                   ' sizeof(pmc_move), sizeof(pmc_stats), offsetof(pmc_params, models), offsetof(pmc_params, seed),'
                   ' offsetof(pmc_pair_model, ndiv2_i), offsetof(pmc_stats, kernel_ms));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.PairModel), C.sizeof(_lib.Params), C.sizeof(_lib.Move), C.sizeof(_lib.Stats),
            _lib.Params.models.offset, _lib.Params.seed.offset, _lib.PairModel.ndiv2_i.offset,
            _lib.Stats.kernel_ms.offset]
    assert got == want


def test_host_constructors_match_oracle_bitwise(pmc):
    from oracle import oracle as O
    fields = ["kind", "n", "eps", "eps4", "sigma", "sigma2", "rcut", "rcut2", "shift", "ndiv2_i", "ndiv2_f",
              "C0", "C2_s2", "C4_s4"]

    def same(a, b):
        for f in fields:
            assert getattr(a, f) == getattr(b, f), f
    for eps, sig in ((1.0, 1.0), (1.5, 0.8), (0.5, 0.88), (1.1523, 1.0339)):
        same(pmc.LennardJones(eps, sig).c_struct(), O.lennard_jones(eps, sig))
        same(pmc.LennardJones(eps, sig, rcut=4.0, shift_potential=False).c_struct(),
             O.lennard_jones(eps, sig, 4.0, False))
        same(pmc.SmoothLennardJones(eps, sig).c_struct(), O.smooth_lennard_jones(eps, sig))
        for n in (12, 14, 9):
            same(pmc.SoftSpheres(eps, sig, n).c_struct(), O.soft_spheres(eps, sig, n))
    rng = np.random.default_rng(0)
    for m_p, m_o in ((pmc.LennardJones(1.0, 1.0), O.lennard_jones(1.0, 1.0)),
                     (pmc.SoftSpheres(1.0, 1.2, 14), O.soft_spheres(1.0, 1.2, 14)),
                     (pmc.SoftSpheres(1.0, 1.2, 9), O.soft_spheres(1.0, 1.2, 9)),
                     (pmc.SmoothLennardJones(1.5, 0.8), O.smooth_lennard_jones(1.5, 0.8))):
        for r2 in rng.uniform(0.5, 9.0, 200):
            assert pmc.potential(r2, m_p) == O.lib().orc_potential(float(r2), C.byref(m_o))
    assert pmc.KobAndersen()[0][1].rcut == 2.5 * 0.8 and pmc.BHHP()[1][1].n == 12 and len(pmc.JBB()) == 3


def test_geometry_matches_oracle_and_survey_f7(pmc):
    from oracle import oracle as O
    from particlesmc_amd.system import box_from_density
    for N, rho, d, rcut, nc in ((10 ** 6, 1.0, 3, 2.5, 39), (32768, 1.0, 3, 2.5, 12), (10 ** 5, 1.2, 3, 2.5, 17),
                                (1290, 1.192060351952634, 2, 2.5, 13)):
        box = box_from_density(N, rho, d)
        ll = pmc.LinkedList(box, rcut, N)
        assert ll.ncells == (nc,) * d
        assert ll.box[0] == box[0] / O.lib().orc_jl_fld(box[0], rcut)
    assert box_from_density(10 ** 6, 1.0, 3)[0] == 99.99999999999997
    assert box_from_density(32768, 1.0, 3)[0] == 31.999999999999993
    with pytest.raises(pmc.PMCError):
        pmc.LinkedList([1.7, 1.7], 2.5, 3)          # README's 3-particle example: rcut > box


def test_host_random_streams_match_oracle(pmc):
    from oracle import oracle as O
    from particlesmc_amd import _lib
    L = _lib.lib()
    out = (C.c_uint32 * 4)()
    L.pmc_philox4x32((C.c_uint32 * 4)(1, 2, 3, 4), (C.c_uint32 * 2)(5, 6), out)
    assert list(out) == O.philox([1, 2, 3, 4], [5, 6])
    a, b = C.c_double(), C.c_double()
    for w1, w2 in ((0, 0), (0xffffffff, 0xffffffff), (123456789, 987654321)):
        L.pmc_box_muller(w1, w2, C.byref(a), C.byref(b))
        assert (a.value, b.value) == O.box_muller(w1, w2)
        L.pmc_log_uniform(w1, w2, C.byref(a))
        assert a.value == O.lib().orc_det_log(O.lib().orc_uniform53(w1, w2))
    s = pmc.SequentialStream(seed=(7 << 32) | 9, chain=3, step=(1 << 33) + 5)
    assert s.R == O.philox([5, 2, 0, 1], [9, 7 + 3]) and s.Q == O.philox([5, 2, 0, 2], [9, 7 + 3])


def test_no_device_means_a_loud_failure_not_a_fallback(pmc):
    if pmc.device_count() > 0:
        pytest.skip("a GPU is visible")
    from inputs import lj_lattice
    pos, sp, rho = lj_lattice(6)
    with pytest.raises(pmc.PMCError) as ei:
        pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0))
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "particlesmc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and '#include "../../oracle' not in txt
                assert "pmc_oracle.h" not in txt


def test_pt_decide_is_deterministic_and_respects_detailed_balance(pmc):
    from particlesmc_amd.tempering import pt_decide
    ladder = np.array([1.0, 1.2, 1.5, 2.0])
    lab = np.arange(4, dtype=np.int32)
    e = np.array([-100.0, -90.0, -95.0, -60.0])
    l1, n1 = pt_decide(11, 0, e, ladder, lab)
    l2, n2 = pt_decide(11, 0, e, ladder, lab)
    assert np.array_equal(l1, l2) and n1 == n2
    assert sorted(l1) == [0, 1, 2, 3]
    # a colder replica holding a HIGHER energy than its hotter neighbour always swaps
    e = np.array([-50.0, -100.0, 0.0, -10.0])
    l3, n3 = pt_decide(5, 0, e, ladder, lab)           # round 0: pairs (0,1), (2,3)
    assert list(l3) == [1, 0, 3, 2] and n3 == 2
    l4, n4 = pt_decide(5, 1, e, ladder, lab)           # round 1: pair (1,2) only
    assert l4[0] == 0 and l4[3] == 3
    # acceptance frequency matches exp(dbeta * dE) for an unfavourable swap
    ladder2 = np.array([1.0, 2.0])
    e2 = np.array([-1.0, 0.0])                         # (1 - 0.5) * (-1 - 0) = -0.5
    acc = sum(pt_decide(99, 2 * r, e2, ladder2, np.arange(2, dtype=np.int32))[1] for r in range(4000))
    assert abs(acc / 4000 - np.exp(-0.5)) < 0.03


def test_cell_and_verlet_list_geometry_match_oracle(pmc):
    """CellList / VerletList constructors (src/neighbours.jl:126-133, 405-406): the host geometry handed to the
    device equals the oracle's for a sweep of boxes and cutoffs, and the host classes carry the reference's fields."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    checked = 0
    for _ in range(200):
        d = int(rng.integers(2, 4))
        N = int(rng.integers(50, 400))
        rho = float(rng.uniform(0.3, 1.3))
        rcut = float(rng.uniform(1.0, 2.6))
        dr = float(rng.uniform(0.05, 0.5))
        box = pmc.system.box_from_density(N, rho, d)
        if box[0] < 3.2 * (rcut + dr):
            continue
        pos = rng.uniform(0, box[0], (N, d))
        for dx in (0.0, 1.0):
            pos[0] = dx
        sp = np.ones(N, dtype=np.int32)
        m = O.single(O.lennard_jones(1.0, 1.0, rcut))                # random points never coincide: finite energy
        cl = pmc.CellList(box, rcut, N)
        ocl = O.System(pos, sp, rho, 1.0, m, O.CELLLIST)
        pairs = [(cl, ocl)]
        try:
            vl = pmc.VerletList(box, rcut, N, list_parameters={"dr": dr})
        except pmc.PMCError:
            # Int(box / cell) is inexact for this box (src/neighbours.jl:406 throws InexactError): the oracle refuses too
            with pytest.raises(ValueError):
                O.System(pos, sp, rho, 1.0, m, O.VERLETLIST, dr)
        else:
            pairs.append((vl, O.System(pos, sp, rho, 1.0, m, O.VERLETLIST, dr)))
            assert vl.dr == dr and vl.dr2o4 == (dr * dr) / 4
            checked += 1
        for host, orc in pairs:
            cell, nc = orc.cell_geometry()
            assert np.array_equal(host.box, np.asarray(cell)[:d]) and tuple(host.ncells) == tuple(int(v) for v in nc[:d])
    assert checked > 50
    with pytest.raises(ValueError):
        pmc.VerletList(np.array([10.0, 10.0]), 2.5, 10)              # "Verlet list must have dr as a parameter"
