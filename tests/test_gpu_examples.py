"""-m gpu: the example scripts run (each in a child process, small arguments)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_soak_many_chains(gpu):
    out = _run([os.path.join(ROOT, "examples", "soak_many.py")])
    assert "== each chain alone" in out


def test_tempering_local_replicas(gpu):
    out = _run([os.path.join(ROOT, "examples", "tempering.py"), "--local", "4", "--side", "64", "--sweeps", "60", "--rounds", "2"])
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d["replicas"] == 4 and d["moves_per_s"] > 1e6 and 0.0 < d["exchange_acceptance"] <= 1.0
    assert len(d["energy_per_particle"]) == 4


def test_throughput_table_many_chains_rows(gpu):
    out = _run([os.path.join(ROOT, "examples", "throughput_table.py"), "many"])
    rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    by = {r["row"]: r["moves_per_s"] for r in rows}
    shared8 = by["8 chains of N=4096 in shared launches (pmc_sweep_many)"]
    assert shared8 > 3 * by["the same 8 chains one after the other"]
    # the launch-wide team width: many small chains run far above the rate of eight
    assert by["256 chains of N=4096 in shared launches (pmc_sweep_many)"] > 2.5 * shared8
