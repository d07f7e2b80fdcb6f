"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on
CPU for the world_size-2 tests).  Agents are independent, so the data path shards them
contiguously by rank and exchanges nothing; the only collectives are the PPO gradient all-reduce
(one flat fp32 buffer per optimiser step) and timing/statistics reductions."""
import os
from typing import Tuple

import torch as th
import torch.distributed as dist


def init(backend: str = None) -> Tuple[int, int, int]:
    """-> (rank, world_size, local_rank); reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if th.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            th.cuda.set_device(local)
            kw["device_id"] = th.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous agent shard of rank: (first agent id, count); remainders go to the low ranks"""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def allreduce_sum_(t: th.Tensor) -> th.Tensor:
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def max_over_ranks(x: float, device="cpu") -> float:
    if world_size() == 1:
        return float(x)
    t = th.tensor([x], dtype=th.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world_size() > 1:
        dist.barrier()
