"""Multi-GPU layout of a batch of scans: one process per GPU, scans sharded round-robin, no
data-path collective -- scans are independent (one correct_default per file,
packages/app/src-tauri/src/task.rs:19-38); the only exchange is the gather of the per-scan
results (best index / angle), 4-12 bytes per scan.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # one process per GPU: bind the device BEFORE the RCCL communicator is created
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n, rank, world):
    """Scan i belongs to rank i % world."""
    return list(range(rank, n, world))


def gather_results(local, n, rank, world, device=None):
    """All ranks receive the full length-n result vector; local[j] is the result of scan
    shard_indices(n, rank, world)[j].  `local`: 1-D tensor (any dtype)."""
    if world == 1:
        return local.clone()
    per = (n + world - 1) // world
    pad = torch.zeros(per, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.empty(n, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n, r, world)
        out[idx] = parts[r][: len(idx)]
    return out


def barrier_max_seconds(seconds, device):
    """MAX over ranks of a per-rank elapsed time."""
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
