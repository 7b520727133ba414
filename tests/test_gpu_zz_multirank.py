"""-m gpu: the path's one collective with MORE THAN ONE RANK.  When the box has >= 2 GPUs this starts 2 (and 8 if present)
fresh processes, one device each, which build an RCCL communicator through the library's own entry points and run 20
exchange rounds (a world of ONE rank runs everywhere and proves the rank script, the id file and the comparison themselves) of BASELINE config 5's system through pmc_replica_exchange_rccl; every rank's labels must equal the
oracle's `pt_decide` on the energies all ranks reported.  On a 1-GPU box the test reports itself as NOT RUN (a skip with
that reason) -- it never passes without having exercised the collective.  (Reference: README.md:33 -- no counterpart;
replica sharding src/IO/IO.jl:319-327.)
(The file name sorts last on purpose: this is the one test whose code path no box of the build rounds could execute, and a
`pytest -x` run on a multi-GPU box should have gone through every other test before it gets here.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NROUNDS = 20


@pytest.mark.parametrize("world", [1, 2, 8])
def test_rccl_replica_exchange_across_ranks(gpu, tmp_path, world):
    from oracle import oracle as O
    ndev = gpu.device_count()
    if ndev < world:
        pytest.skip(f"NOT RUN: pmc_replica_exchange_rccl with {world} ranks needs {world} GPUs, this box has {ndev} "
                    "(the world-size-1 RCCL path and the world-size-2 decision path over gloo are covered elsewhere)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    ranks = _run_ranks(world, tmp_path, env, devices=list(range(world)))
    _check_against_oracle(O, ranks, world, tmp_path)


def _run_ranks(world, tmp_path, env, devices):
    """One fresh process per rank (device devices[r]); every rank's output goes to its own file, so a chatty rank cannot block a
    collective on a full pipe.  Returns what the ranks reported."""
    logs = [open(tmp_path / f"rank{r}.log", "wb") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, "-m", "particlesmc_amd.rccl_check", str(r), str(world), str(devices[r]), str(tmp_path), str(NROUNDS)],
                              cwd=ROOT, env=env, stdout=logs[r], stderr=subprocess.STDOUT) for r in range(world)]
    try:
        for p in procs:
            try:
                p.wait(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail("a rank did not finish within 600 s")
    finally:
        for f in logs:
            f.close()
    outs = [open(tmp_path / f"rank{r}.log", "rb").read().decode(errors="replace") for r in range(world)]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


def _check_against_oracle(O, ranks, world, tmp_path):
    ladder = ranks[0]["ladder"]
    label = list(range(world))
    accepted = 0
    for rnd in range(NROUNDS):
        energies = [ranks[r]["rounds"][rnd]["energy"] for r in range(world)]
        label, nacc = O.pt_decide(9, rnd, energies, ladder, label)
        accepted += nacc
        for r in range(world):
            got = ranks[r]["rounds"][rnd]
            assert got["label"] == label and got["nacc"] == nacc, (rnd, r, got, label)
            assert got["temperature"] == ladder[label[r]]
    assert accepted > 0 or world == 1                                # replicas did exchange temperatures
    assert not os.path.exists(tmp_path / "rccl_id")                  # rank 0 removed the id file after the join


@pytest.mark.parametrize("world", [2, 3, 8])
def test_replica_exchange_across_processes_through_the_stub(gpu, tmp_path, world):
    """the >= 2-rank path of pmc_rccl_comm_create + pmc_replica_exchange_rccl executed on a ONE-GPU box.
    The library opens the library named by PMC_RCCL_LIB instead of librccl.so; tests/c/rccl_stub.c (test infrastructure: the
    four RCCL entry points over files, collective semantics kept) lets `world` processes share device 0.  What runs for real:
    the id file with its launch key, the rendezvous, the all-gather ordered on the handle's stream after the sweep that produced
    the energy, the decisions and the label every rank adopts -- checked against the oracle's pt_decide on the energies all ranks
    reported.  The real-RCCL test above keeps skipping honestly on this box."""
    from oracle import oracle as O
    stub = _build_stub(tmp_path)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PMC_RCCL_LIB"] = str(stub)
    env["TMPDIR"] = str(tmp_path)
    _oversubscribed(env, world)
    ranks = _run_ranks(world, tmp_path, env, devices=[0] * world)
    _check_against_oracle(O, ranks, world, tmp_path)


def _oversubscribed(env, world):
    """Many processes on ONE device time-slice it: a workgroup of a fused sweep launch that waits for a lower ticket of its own
    launch can then sit out its bounded spin while the queue of its process is descheduled (seen with 8 processes; never with
    one process per GPU, which is the contract).  The rank path and the collective are what these tests are about, so the
    oversubscribed cases take the separate colour launches, which have no intra-launch waiting."""
    if world > 3:
        env["PMC_NO_FUSED"] = "1"


def _build_stub(tmp_path):
    stub = tmp_path / "librccl_stub.so"
    subprocess.check_call(["gcc", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-shared", "-fPIC", "-O2", "-o", str(stub),
                           os.path.join(ROOT, "tests", "c", "rccl_stub.c"), "-L/opt/rocm/lib", "-lamdhip64"])
    return stub


@pytest.mark.timeout(900)
def test_bench_eight_ranks_share_one_device_through_the_stub(gpu, tmp_path):
    """the WHOLE rank path of `bench.py --gpus 8` on a one-GPU box.  `--share-devices` puts the eight ranks
    on device 0 (torch.distributed over gloo: barriers, max-over-ranks timing, the one JSON line from rank 0), and with
    PMC_RCCL_LIB naming tests/c/rccl_stub.c's library every rank starts its child process
    (`python -m particlesmc_amd.rccl_check`) exactly as an 8-GPU run does: communicator through pmc_rccl_unique_id /
    pmc_rccl_comm_create with the id in a file, sweeps + pmc_replica_exchange_rccl, labels checked against pmc_pt_decide by rank 0.
    `--strict-rccl` turns a leg that did not verify into exit status 4, `--launch-timeout` a hang of any rank into status 5 with
    the ranks' process group killed (never a re-exec); the line must carry n_ranks_rccl: 8.  What stays unexecuted is RCCL's
    own transport."""
    stub = _build_stub(tmp_path)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PMC_RCCL_LIB"] = str(stub)
    env["TMPDIR"] = str(tmp_path)
    _oversubscribed(env, 8)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-devices", "--strict-rccl",
                        "--launch-timeout", "600", "--lattice", "16", "--steps", "4", "--warmup", "1", "--equilibrate", "4",
                        "--long-steps", "0", "--no-cpu", "--no-others"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["ranks"] == 8 and out["config"]["backend"] == "gloo"
    assert out["n_ranks_rccl"] == 8, out.get("rccl_exchange")
    assert out["rccl_exchange"]["decisions_equal_pt_decide"] is True and out["rccl_exchange"]["transport"] == "librccl_stub.so"
    assert out["running_energy_equals_recomputed"] is True
    # whole-job value: eight ranks' moves over the max-over-ranks time
    assert out["value"] == pytest.approx(8 * 16 ** 3 * 4 / (out["ms_per_step"] * 4e-3), rel=1e-9)
