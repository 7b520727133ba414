"""bench.py's multi-rank contract on a machine without a GPU: `--gpus N` started plainly launches N
ranks itself (torchrun children, gloo here), rank 0 prints ONE JSON line with n_gpus == N, and a rank
count that does not match the visible devices is an error instead of a silent share.  The device work
is stubbed (PMC_BENCH_STUB=1); the GPU path of the same script is what the driver runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
MARK = f"pmc-launch-timeout-test-{os.getpid()}"


def _run(args, **env):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=280)


@pytest.mark.timeout(300)
def test_gpus2_self_launches_two_ranks_and_reports_n_gpus_2():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--lattice", "6", "--no-cpu"], PMC_BENCH_STUB="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["backend"] == "gloo"
    assert out["stub"] is True and out["steps"] == 3 and out["scaling"] == "weak"
    # whole-job value: both ranks' moves over the max-over-ranks time
    assert out["value"] == pytest.approx(2 * 6 ** 3 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-9)


@pytest.mark.timeout(120)
def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--lattice", "6", "--no-cpu"],
             PMC_BENCH_STUB="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.timeout(300)
def test_more_ranks_than_devices_is_refused_not_shared():
    """Without the stub on this GPU-less machine every rank sees 0 devices: exit status 3, no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--lattice", "6", "--no-cpu"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "visible GPU" in r.stderr or "needs a HIP device" in r.stderr


@pytest.mark.timeout(120)
def test_launch_timeout_kills_the_ranks_and_exits_nonzero():
    """a rank that hangs must end `bench.py --gpus N` with a non-zero status -- the parent kills the process
    group of the ranks it started (never a re-exec).  The stub engine sleeps when PMC_BENCH_STUB_HANG names the rank."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--lattice", "6", "--no-cpu", "--launch-timeout", "20"],
             PMC_BENCH_STUB="1", PMC_BENCH_STUB_HANG="1", PMC_BENCH_TEST_MARK=MARK)
    assert r.returncode == 5, (r.returncode, r.stderr[-1500:])
    assert "did not finish within" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # the ranks themselves must be gone (the elastic agent's workers are not in the launcher's process group; the
    # parent finds them in /proc, no optional package).  Every process of the job carries the marker in its environment.
    import time
    time.sleep(0.5)
    left = []
    for e in os.listdir("/proc"):
        if e.isdigit():
            try:
                with open(f"/proc/{e}/environ", "rb") as f:
                    if MARK.encode() in f.read():
                        left.append(int(e))
            except OSError:
                pass
    assert not left, f"processes of the hung job survived the kill: {left}"
