"""Data-parallel sharding of image pairs: one process per GPU, no tensor split.

Every pair is independent in inference (no cross-sample op in ``flower``,
qpwcnet/core/pwcnet.py:28-67), so the only exchange is one all-gather of the
per-level EPE vector -- 6 floats per rank, over RCCL/xGMI on GPUs
(``torch.distributed`` backend "nccl" IS RCCL on ROCm), gloo in the CPU tests.
The reference itself has no multi-device code.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise from the torchrun environment; no-op for a single process.
    -> (world_size, rank, local_rank)."""
    world, rank, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, local_rank


def shard_range(n_items, rank, world):
    """Contiguous shard of ``n_items`` pairs for ``rank``: [lo, hi).  Remainders go
    to the lowest ranks, so shards differ by at most one pair."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world {}/{}".format(rank, world))
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_epe(local_epe, n_local=None):
    """All-gather the per-rank per-level EPE vector.

    local_epe: float32 [L].  n_local: pairs on this rank (weights the mean when
    shards are uneven).  -> (per_rank [world, L], global_mean [L])."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        per_rank = local_epe.unsqueeze(0)
        return per_rank, local_epe.clone()
    world = dist.get_world_size()
    w = torch.tensor([1.0 if n_local is None else float(n_local)], dtype=local_epe.dtype,
                     device=local_epe.device)
    payload = torch.cat([local_epe, w])
    flat = torch.empty(world * payload.numel(), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(flat, payload)  # one ncclAllGather (RCCL) on GPUs
    out = flat.view(world, payload.numel())
    per_rank, weights = out[:, :-1], out[:, -1:]
    return per_rank, (per_rank * weights).sum(dim=0) / weights.sum()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
