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
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank.py"), str(r), str(world), str(tmp_path), str(NROUNDS)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank did not finish within 600 s")
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    ranks = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
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
