"""julia/PMCHip.jl made falsifiable without a Julia binary (there is none in the build image).

* layout (CPU): the isbits structs the glue declares are parsed out of the .jl text, laid out by the C rules Julia
  uses for isbits structs, and compared with `offsetof` from the C compiler (tests/c/glue_replay.c --layout) and
  with the ctypes mirror the GPU tests go through;
* call sequences (-m gpu): tests/c/glue_replay.c performs every call the glue performs, in the glue's order and
  with its argument shapes, against libpmc_hip.so; the results are compared with the oracle.
"""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "PMCHip.jl")
SRC = os.path.join(ROOT, "tests", "c", "glue_replay.c")
CSRC = os.path.join(ROOT, "particlesmc_amd", "csrc")

JL_TYPES = {"Int32": (4, 4), "UInt32": (4, 4), "Int64": (8, 8), "UInt64": (8, 8), "Float64": (8, 8)}


def _jl_field(t):
    """(size, alignment) of a Julia isbits field type as C lays it out."""
    t = t.strip()
    if t in JL_TYPES:
        return JL_TYPES[t]
    m = re.fullmatch(r"NTuple\{(\d+),\s*(\w+)\}", t)
    if m:
        s, a = JL_TYPES[m.group(2)]
        return int(m.group(1)) * s, a
    if t.startswith("Ptr{"):
        return 8, 8
    raise AssertionError(f"unhandled Julia field type {t}")


def _jl_struct(name):
    txt = open(JL).read()
    body = re.search(r"^struct " + name + r"\b[^\n]*\n(.*?)^end", txt, re.S | re.M).group(1)
    fields = []
    for line in body.split("\n"):
        line = line.split("#")[0]
        for part in line.split(";"):
            part = part.strip()
            if part:
                fname, ftype = part.split("::")
                fields.append((fname.strip(), ftype.strip()))
    off, maxa, out = 0, 1, []
    for fname, ftype in fields:
        size, align = _jl_field(ftype)
        off = (off + align - 1) // align * align
        out.append((fname, off, size))
        off += size
        maxa = max(maxa, align)
    return out, (off + maxa - 1) // maxa * maxa


@pytest.fixture(scope="module")
def harness(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("glue") / "glue_replay")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe, SRC,
                           "-L", CSRC, "-lpmc_hip", "-Wl,-rpath," + CSRC])
    return exe


def test_julia_struct_layouts_equal_the_c_header_and_the_ctypes_mirror(harness):
    from particlesmc_amd import _lib
    lines = subprocess.check_output([harness, "--layout"]).decode().split("\n")
    c = {}
    for ln in lines:
        if ln.strip():
            parts = ln.split()
            c[parts[0]] = tuple(int(v) for v in parts[1:])
    pairs = [("PairModel", "pmc_pair_model", _lib.PairModel), ("Params", "pmc_params", _lib.Params),
             ("CMove", "pmc_move", _lib.Move), ("Stats", "pmc_stats", _lib.Stats)]
    import ctypes as C
    for jl, cname, ct in pairs:
        fields, size = _jl_struct(jl)
        assert (size,) == c[f"{cname}.sizeof"] and size == C.sizeof(ct), (jl, size, c[f"{cname}.sizeof"], C.sizeof(ct))
        cfields = [k.split(".")[1] for k in c if k.startswith(cname + ".") and not k.endswith(".sizeof")]
        assert len(fields) == len(cfields) == len(ct._fields_), (jl, len(fields), len(cfields))
        for (fname, off, fsize), cf, (ctname, *_rest) in zip(fields, cfields, ct._fields_):
            assert (off, fsize) == c[f"{cname}.{cf}"], (jl, fname, cf, off, fsize, c[f"{cname}.{cf}"])
            assert getattr(ct, ctname).offset == off and getattr(ct, ctname).size == fsize, (jl, fname, ctname)
            # same name on all three sides (Greek letters in the Julia constructors aside, the struct fields are ASCII)
            assert fname == cf == ctname, (fname, cf, ctname)


def _jl_functions():
    """{first line: body} of every `function ... end` block and one-line method of the glue (column-0 definitions)."""
    txt = open(JL).read()
    out = {}
    for m in re.finditer(r"^function ([^\n]*)\n(.*?)^end", txt, re.S | re.M):
        out[m.group(1)] = m.group(2)
    for m in re.finditer(r"^([A-Za-z_.!]+\([^\n]*\)(?: where \{[^}]*\})?) = ([^\n]*(?:\n {8,}[^\n]*)*)", txt, re.M):
        out[m.group(1)] = m.group(2)
    return out


def test_glue_never_assigns_to_a_field_of_the_immutable_atoms_struct():
    """`Atoms` is an immutable struct (reference src/atoms.jl:18): `system.<field> = ...` throws.  Elements of its
    arrays (`system.energy[1] = `, `system.species .= `) are fine.  Also: nothing of the glue may read system.energy
    inside build_neighbour_list! (undef at src/atoms.jl:50), and every entry point that depends on the running energy
    hands it over first."""
    txt = open(JL).read()
    code = "\n".join(ln.split("#")[0] for ln in txt.split("\n"))
    fields = "position|species|density|energy|temperature|model_matrix|N|d|box|neighbour_list|species_list"
    bad = re.findall(r"\b(?:system|s|sys)\.(?:" + fields + r")\s*(?:=(?!=)|\+=|-=|\*=)", code)
    assert bad == [], bad
    assert re.search(r"\bs\.temperature\b", code) is None            # (the round-2 text assigned it in two places)
    fns = _jl_functions()
    build = [b for k, b in fns.items() if k.startswith("ParticlesMC.build_neighbour_list!")]
    assert len(build) == 1
    body = "\n".join(ln.split("#")[0] for ln in build[0].split("\n"))
    assert "energy" not in body and ":pmc_set_energy" not in body
    needs = [k for k, b in fns.items() if any(sym in b for sym in (":pmc_displacement_perform", ":pmc_displacement_revert", ":pmc_swap_perform",
                                                                   ":pmc_swap_revert", ":pmc_sweep,", ":pmc_sweep_many", ":pmc_replica_exchange"))]
    assert len(needs) >= 7, needs
    for k in needs:
        b = fns[k]
        first_call = min(b.find(sym) for sym in (":pmc_displacement", ":pmc_swap", ":pmc_sweep", ":pmc_replica") if sym in b)
        assert "handover!" in b[:first_call], k
    # reverts restore the host-side running energy exactly as src/moves.jl:79 and :206 do
    for k, b in fns.items():
        if k.startswith("Arianna.revert_action!"):
            assert "system.energy[1] -= action.δe" in b, k
    # the temperature of a HIP-backed chain is read from the handle by both target-density functions (src/utils.jl:4-10)
    assert "Arianna.delta_log_target_density(e1, e2, system::ParticlesMC.Atoms{D,V,<:HIPLinkedList})" in txt
    assert "Arianna.unnormalised_log_target_density(e, system::ParticlesMC.Atoms{D,V,<:HIPLinkedList})" in txt
    assert "function (L::HIPLinkedList)(system::ParticlesMC.Particles, i::Int)" in txt and ":pmc_neighbours_of" in txt


def test_replay_follows_the_reference_constructor_order():
    """tests/c/glue_replay.c must perform System()'s steps in the order of src/atoms.jl:47-56: create/upload/build, THEN the
    N compute_energy_particle calls and the host sum, and only afterwards the first pmc_set_energy."""
    src = re.sub(r"/\*.*?\*/", "", open(SRC).read(), flags=re.S)      # code only
    main = src[src.index("int main("):]
    i_build = main.index("pmc_build_cells(h)")
    i_loop = main.index("pmc_energy_particle(h, i, &local[i - 1])")
    i_sum = main.index("pairwise_sum(local, 0, N - 1) / 2.0")
    i_first_set = main.index("HANDOVER();")
    assert i_build < i_loop < i_sum < i_first_set
    assert "pmc_set_energy" not in main[:i_sum] and "pmc_init_energy" not in main[:i_first_set]


def test_glue_blocks_are_balanced():
    """No Julia binary parses the file here; at least every block opener has its `end` (comments and strings stripped,
    the `for` of a comprehension -- inside brackets -- does not open a block)."""
    depth = 0
    for n, line in enumerate(open(JL).read().split("\n"), 1):
        code = re.sub(r'"[^"\n]*"', '""', line.split("#")[0])
        flat, prev = code, None
        while flat != prev:                                            # drop bracketed text, innermost first (comprehensions)
            prev, flat = flat, re.sub(r"\[[^\[\]]*\]", "", flat)
        opens = re.findall(r"(?<![\w.!:])(function|struct|if|for|begin|module|while|try|let|do)(?![\w!])", flat)
        ends = re.findall(r"(?<![\w.!:])end(?![\w!])", flat)
        depth += len(opens) - len(ends)
        assert depth >= 0, (n, line)
    assert depth == 0


def test_glue_declares_an_abstract_rng_over_the_sequential_words():
    txt = open(JL).read()
    assert "struct SequentialWords <: Random.AbstractRNG" in txt and ":pmc_sequential_words" in txt
    assert "Random.rand(rng::SequentialWords, ::Random.SamplerType{UInt32})" in txt


@pytest.mark.gpu
def test_glue_call_sequences_replayed_in_c_match_the_oracle(gpu, harness, tmp_path):
    from oracle import oracle as O
    from inputs import lj_lattice
    pmc = gpu
    pos, sp, rho = lj_lattice(12, jitter=0.06, seed=21)
    N, d, S, T, rcut = len(pos), 3, 2, 1.1, 2.4
    sp = np.ones(N, dtype=np.int32)
    sp[np.random.default_rng(4).permutation(N)[: N // 3]] = 2
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.txt")
    with open(fin, "wb") as f:
        f.write(struct.pack("<qiiddd", N, d, S, rho, T, rcut))
        f.write(np.ascontiguousarray(pos).tobytes())
        f.write(sp.tobytes())
    subprocess.check_call([harness, fin, fout], timeout=300)
    out = {}
    for ln in open(fout):
        k, *v = ln.split()
        out.setdefault(k, []).append(v)

    # the harness builds its models with the library's host constructors; the oracle's are the independent restatement
    arr = (O.Model * (S * S))()
    for i in range(S):
        for j in range(S):
            arr[i * S + j] = O.lennard_jones(1.0 + 0.25 * (i != j), 1.0 - 0.1 * (i + j) / 2.0, rcut)
    mo = (arr, S)
    o = O.System(pos, sp, rho, T, mo, O.LINKEDLIST)
    assert float(out["energy0"][0][0]) == o.energy
    assert np.array_equal(np.array(out["cs"][0], dtype=np.int64), o.cells())
    for i, e in out["e_particle"]:
        assert float(e) == o.energy_particle(int(i))
    # (L::HIPLinkedList)(system, i): the LinkedIterator's sequence (src/neighbours.jl:316-377), i itself included
    assert len(out["neighbours"]) == 3
    for row in out["neighbours"]:
        assert np.array_equal(np.array(row[1:], dtype=np.int64), o.neighbours_of(int(row[0]))) and int(row[0]) in [int(v) for v in row[1:]]
    assert out["neighbours_retry"][0][0] == out["neighbours_retry"][0][1] and int(out["neighbours_retry"][0][0]) == len(o.neighbours_of(1))
    delta = np.array([0.031, -0.017, 0.044])
    r = o.displacement_perform(5, delta)
    # (column 4: the library's running energy; column 5: the host-side system.energy[1] the glue keeps: they must stay equal)
    assert tuple(float(v) for v in out["disp_perform"][0][:3]) == r
    assert float(out["disp_perform"][0][3]) == o.energy == float(out["disp_perform"][0][4])
    o.displacement_revert(5, -delta, r[2])
    assert float(out["disp_revert"][0][0]) == o.energy == float(out["disp_revert"][0][1])
    i, j = int(out["swap_perform"][0][0]), int(out["swap_perform"][0][1])
    r = o.swap_perform(i, j, 1, 2)
    assert tuple(float(v) for v in out["swap_perform"][0][2:5]) == r
    assert float(out["swap_perform"][0][5]) == o.energy == float(out["swap_perform"][0][6])
    o.swap_revert(j, i, 1, 2, r[2])
    assert float(out["swap_revert"][0][0]) == o.energy == float(out["swap_revert"][0][1])
    pool = O.make_pool([{"kind": "Displacement", "sigma": 0.05, "probability": 0.8},
                        {"kind": "DiscreteSwap", "species": (1, 2), "probability": 0.2}])
    # make_step! in sequential mode: N steps of the reference chain (seed 77, chain 3, from step 0)
    att, acc = o.mc_steps(pool, 77, 3, 0, N)
    s0 = out["sweep"][0]
    assert [int(s0[1]), int(s0[2]), int(s0[3]), int(s0[4])] == [att[0], acc[0], att[1], acc[1]] and float(s0[5]) == o.energy
    # ... then in checkerboard mode: sweep 1 of the schedule (the sequential sweep advanced the sweep counter to 1)
    cb = O.Checkerboard(o.positions(), o.species(), o.box, T, mo, seed=77, chain=3)
    att, acc = cb.sweeps(pool, 1, 1)
    s1 = out["sweep"][1]
    assert [int(s1[1]), int(s1[2]), int(s1[3]), int(s1[4])] == [att[0], acc[0], att[1], acc[1]] and int(s1[6]) == cb.energy_fix
    # sweep_many!: two rounds of two sweeps
    for k, row in enumerate(out["sweep_many"]):
        att, acc = cb.sweeps(pool, 2 + 2 * k, 2)
        assert [int(row[0]), int(row[1])] == [att[0], acc[0]] and int(row[3]) == cb.energy_fix
    # refresh!: positions, species, energy
    assert np.array_equal(np.array(out["pos"][0], dtype=np.float64).reshape(N, d), cb.positions())
    assert np.array_equal(np.array(out["species"][0], dtype=np.int32), cb.species())
    assert float(out["refresh_energy"][0][0]) == cb.energy_fix / 2.0 ** 32
    assert float(out["temperature"][0][0]) == 1.5 * T
    # EnergyBias overloads: the device's pick and densities equal the oracle's restatement on the same state
    o2 = O.System(cb.positions(), cb.species(), rho, T, mo, O.LINKEDLIST)
    eb = out["energy_bias"][0]
    want = o2.energy_bias((1, 2), (0.3, -0.2), (0.25, 0.75))
    assert (int(eb[0]), int(eb[1]), float(eb[2])) == want and float(eb[3]) == want[2]
    # sweep_many! over two chains of one shape (chain 3 continues at sweep 6, chain 4 starts at sweep 0), then replica_exchange!
    att, acc = cb.sweeps(pool, 6, 2)
    cb2 = O.Checkerboard(o2.positions(), o2.species(), o.box, 1.05 * T, mo, seed=77, chain=4)
    att2, acc2 = cb2.sweeps(pool, 0, 2)
    m0, m1 = out["many"]
    assert [int(m0[1]), int(m0[2]), int(m0[3])] == [att[0], acc[0], cb.energy_fix]
    assert [int(m1[1]), int(m1[2]), int(m1[3])] == [att2[0], acc2[0], cb2.energy_fix]
    lab, nacc = O.pt_decide(99, 0, [cb.energy_fix / 2.0 ** 32, cb2.energy_fix / 2.0 ** 32], [T, 1.05 * T], [0, 1])
    ex = out["exchange"][0]
    assert [int(ex[0]), int(ex[1]), int(ex[2])] == lab + [nacc]
    assert [float(ex[3]), float(ex[4])] == [[T, 1.05 * T][lab[0]], [T, 1.05 * T][lab[1]]]
    # the words the AbstractRNG wrapper serves: Philox blocks of the sequential stream
    # (DESIGN.md "random streams": ctr = (step_lo, step_hi, 0, TAG_SEQ0 | TAG_SEQ1), key = (seed_lo, seed_hi + chain))
    for row in out["words"]:
        step = int(row[0])
        assert [int(v) for v in row[1:5]] == O.philox((step & 0xffffffff, step >> 32, 0, 1), (77, 3))
        assert [int(v) for v in row[5:9]] == O.philox((step & 0xffffffff, step >> 32, 0, 2), (77, 3))
    assert len(out["words"]) == 3
    del pmc


def test_default_tick_is_one_sweep_as_in_the_reference():
    """The reference's Metropolis entry runs one mc_sweep! of N steps per scheduler tick (sweepstep = length(chains[1]),
    src/ParticlesMC.jl:246): the glue's default HIPMetropolis must do the same -- callbacks and schedulers then fire at the
    reference's cadence in both modes -- and longer ticks must be an explicit keyword."""
    txt = open(JL).read()
    m = re.search(r"^HIPMetropolis\(pool, mode; sweeps_per_step=(\d+)\) = HIPMetropolis\(pool, mode, sweeps_per_step\)", txt, re.M)
    assert m and int(m.group(1)) == 1, "the default tick must be one sweep"
    assert "4 * rounds" not in txt
