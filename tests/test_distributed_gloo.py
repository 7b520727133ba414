"""CPU, world_size 2 over gloo: the replica-exchange step and the bench's cross-rank reductions."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from particlesmc_amd import parallel
    from particlesmc_amd.tempering import ReplicaExchange
    d, r, lr, w = parallel.init("gloo")
    assert (r, w) == (rank, world)
    ladder = [1.0, 1.5]
    rex = ReplicaExchange(ladder, seed=17, rank=rank, world=world)
    rng = np.random.default_rng(100 + rank)             # each rank only knows its own energy
    labels = []
    for rnd in range(40):
        e = float(rng.normal(-100.0, 5.0))
        rex.exchange(e)
        labels.append(rex.label.copy())
    tmax = parallel.max_over_ranks(1.0 + rank, d)
    tsum = parallel.sum_over_ranks(1.0 + rank, d)
    q.put((rank, np.array(labels), rex.accepted, rex.attempted, tmax, tsum, rex.temperature))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_replica_exchange_two_ranks_gloo(built):
    import torch.multiprocessing as mp
    from particlesmc_amd.tempering import pt_decide
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lab0, acc0, att0, tmax0, tsum0, t0), (r1, lab1, acc1, att1, tmax1, tsum1, t1) = res
    assert np.array_equal(lab0, lab1) and acc0 == acc1 and att0 == att1     # identical decisions on every rank
    assert tmax0 == tmax1 == 2.0 and tsum0 == tsum1 == 3.0
    assert {t0, t1} == {1.0, 1.5}                                           # the ladder stays a permutation
    # single-process replay of the same all-gathered energies gives the same labels
    rngs = [np.random.default_rng(100), np.random.default_rng(101)]
    label = np.arange(2, dtype=np.int32)
    acc = 0
    for rnd in range(40):
        e = np.array([float(g.normal(-100.0, 5.0)) for g in rngs])
        label, n = pt_decide(17, rnd, e, [1.0, 1.5], label)
        acc += n
        assert np.array_equal(label, lab0[rnd])
    assert acc == acc0 and att0 == 20                                       # with 2 replicas only even rounds pair up
