"""One rank of the multi-GPU replica-exchange test (started by tests/test_gpu_zz_multirank.py, one fresh process per GPU):
`rccl_rank.py RANK WORLD OUTDIR NROUNDS` = particlesmc_amd.rccl_check with device = rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from particlesmc_amd import rccl_check  # noqa: E402

if __name__ == "__main__":
    rank, world, outdir, nrounds = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    rccl_check.run(rank, world, rank, outdir, nrounds)
