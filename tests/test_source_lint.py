"""Source-level guard for the HIP kernels (no GPU needed).

A cross-lane read (`readlane`, `__shfl*`, DPP) of a lane that is INACTIVE at the point of the read is undefined: it
returns the right value while the operand sits in a register and garbage once the compiler reloads a spilled operand
inside the divergent branch (only the active lanes are reloaded).  That is what made single instances of the sweep
kernel compute wrong energies "under scratch" in rounds 2 and 3 (`readlane_double(g_tlnu, my)` inside
`if (lane == 0)`).  The kernels keep every such read outside lane-selecting branches; this test reads the sources and
checks it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = [os.path.join(ROOT, "particlesmc_amd", "csrc", f) for f in ("pmc_kernels.h", "pmc_ref.h", "pmc_device.h", "pmc_hip.hip", "pmc_seq.h", "pmc_verlet.h", "pmc_empty.h")]
CROSS = re.compile(r"readlane|__shfl|update_dpp|dpp_add|mov_dpp|ds_bpermute|ds_swizzle|wave_sum_exact|wave_sum\b")
# branches that switch lanes off by their index: `if (lane == 0)`, `if (tid == 0 ...`, `if (lane < n ...`, `if (... && lane == i)`
LANE_BRANCH = re.compile(r"\bif\s*\((?:[^()]|\([^()]*\))*\b(?:lane|tid|threadIdx\.x)\s*(?:==|<|<=|>|>=|!=)(?:[^()]|\([^()]*\))*\)")
# ... and loops whose trip count depends on the lane: `for (int k = lane; k < n; k += 64)`
LANE_LOOP = re.compile(r"\b(?:for|while)\s*\((?:[^()]|\([^()]*\))*\b(?:lane|tid|threadIdx\.x)\b(?:[^()]|\([^()]*\))*\)")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def branch_body(text, start):
    """the statement or block an `if (...)` that ends at `start` controls"""
    i = start
    while i < len(text) and text[i] in " \t\n":
        i += 1
    if i < len(text) and text[i] == "{":
        depth, j = 0, i
        while j < len(text):
            if text[j] == "{":
                depth += 1
            elif text[j] == "}":
                depth -= 1
                if depth == 0:
                    return text[i:j + 1]
            j += 1
        return text[i:]
    j = text.find(";", i)
    return text[i:j + 1]


def test_no_cross_lane_read_inside_a_lane_selecting_branch():
    found = []
    for path in SOURCES:
        text = strip_comments(open(path).read())
        for m in list(LANE_BRANCH.finditer(text)) + list(LANE_LOOP.finditer(text)):
            body = branch_body(text, m.end())
            hit = CROSS.search(body)
            if hit:
                line = text.count("\n", 0, m.start()) + 1
                found.append(f"{os.path.basename(path)}:{line}: `{m.group(0)[:60]}` controls a `{hit.group(0)}`")
    assert not found, "\n".join(found)


def test_the_guard_sees_the_pattern_it_guards_against():
    bad = "if (lane == 0) {\n  double *r = recs;\n  r[7] = readlane_double(g_tlnu, my);\n}\n"
    m = LANE_BRANCH.search(bad)
    assert m and CROSS.search(branch_body(bad, m.end()))
    ok = "const double t = readlane_double(g_tlnu, my);\nif (lane == 0) { r[7] = t; }\n"
    m = LANE_BRANCH.search(ok)
    assert m and not CROSS.search(branch_body(ok, m.end()))


def test_no_experiment_switches_in_the_product_sources():
    """experiments live in LAB.md and in the history, not in the files the product compiles.  No `PMC_EXP_*`
    symbol and no matrix-core reduction switch may come back into the kernel sources or the host code."""
    import glob
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "particlesmc_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")) + [os.path.join(csrc, "Makefile")]):
        txt = open(path).read()
        for sym in ("PMC_EXP_", "PMC_MFMA_REDUCE"):
            assert sym not in txt, f"{os.path.basename(path)} contains the experiment switch {sym}*"
        # round 6: the measured-and-lost two-launch round of the windows is gone, and the library never changes its
        # execution path because a profiler is attached (the profile scripts say PMC_SEQ_NOGRAPH=1 themselves)
        for sym in ("PMC_SEQ_TWO", "ROCP_TOOL_LIBRARIES", "PMC_SEQ_GRAPH\""):
            assert sym not in txt, f"{os.path.basename(path)} contains {sym}"
