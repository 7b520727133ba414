"""One process per GPU: rendezvous helpers shared by bench.py and the tempering driver."""
import os


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (nccl = RCCL on GPUs, gloo on CPU)."""
    rank, local_rank, world = env_ranks()
    if world == 1:
        return None, rank, local_rank, world
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist, rank, local_rank, world


def max_over_ranks(value, dist=None, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
