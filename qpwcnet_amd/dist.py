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


class EpeGather:
    """The same exchange with the collective of step k in flight behind the compute of step k+1:
    ``submit(local_epe)`` copies the vector into one of two payload buffers and starts an asynchronous
    all-gather (on RCCL's own stream, ordered after the caller's stream up to this point);
    ``collect()`` returns the OLDEST outstanding result -- the caller's stream waits for that collective,
    which by then has had a whole step to finish -- as (per_rank [world, L], global_mean [L]).
    Buffers are allocated once; nothing is copied from the host per step.  Single process: no
    collective, submit/collect are a queue of clones.

    Copy-free form (what bench.py's hipGraph steps use): the producer writes its vector straight into
    ``payload_view(slot)`` in stream order -- the EPE reduction captured inside graph `slot` does -- and
    calls ``submit(slot=slot)``.  Slots alternate 0, 1, 0, ...: slot s is rewritten by step k+2 only after
    ``collect()`` of step k made the caller's stream wait for all-gather k, which read it (at most two
    collectives are ever outstanding, enforced below)."""

    def __init__(self, n_levels, device, n_local=1, dtype=torch.float32, keep_history=None):
        """keep_history: does collect() hand out a copy that stays valid for ever (True), or -- single process, vector
        written in place -- the payload window itself, valid until its slot comes round again two steps later
        (False)?  Default: copies wherever they cost nothing extra (a process group clones after its all-gather
        anyway; CPU tensors), windows for a single process on a GPU, where the copy is one more launch per step on
        the compute stream (tools/epe_copy_ab.py, one process: 1.237 vs 1.230 ms/step; on a side stream ordered by
        events 1.260).  timed_steps() then returns None for the entries that are no longer valid instead of
        aliases of two buffers."""
        # a process group of one rank still runs the collective (that is how the RCCL path is tested on
        # a one-GPU box); no process group = plain single process
        self.collective = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.collective else 1
        self.L = int(n_levels)
        self.pending = []
        self.slot = 0
        if keep_history is None:
            keep_history = self.collective or torch.device(device).type != "cuda"
        self.keep_history = bool(keep_history) or self.collective
        w = self.L + 1
        self.payload = [torch.zeros(w, dtype=dtype, device=device) for _ in range(2)]
        for p in self.payload:
            p[-1] = float(n_local)      # shard weight, constant over the run
        if self.collective:
            self.flat = [torch.empty(self.world * w, dtype=dtype, device=device) for _ in range(2)]

    def payload_view(self, slot):
        """The L-float window of payload buffer `slot` (0 or 1) a producer may write in place."""
        return self.payload[slot][:self.L]

    def next_slot(self):
        return self.slot

    def submit(self, local_epe=None, slot=None):
        """local_epe: the step's vector (copied into the next payload slot).  slot: the vector is
        already in ``payload_view(slot)`` (must be ``next_slot()``)."""
        if len(self.pending) >= 2:
            raise RuntimeError("EpeGather: collect() the oldest result before submitting a third")
        k = self.slot
        if slot is not None:
            if local_epe is not None:
                raise ValueError("EpeGather.submit: pass a vector or a slot, not both")
            if slot != k:
                raise RuntimeError("EpeGather: slot {} submitted, slot {} is next".format(slot, k))
        self.slot ^= 1
        if not self.collective:
            # single process: nothing to exchange.  A slot result is the payload window itself until collect()
            # clones it (the slot comes round again two steps later); a vector is kept as a clone.
            self.pending.append((None, self.payload[k][:self.L] if slot is not None else local_epe.clone()))
            return
        if slot is None:
            self.payload[k][:self.L].copy_(local_epe)
        work = dist.all_gather_into_tensor(self.flat[k], self.payload[k], async_op=True)
        self.pending.append((work, self.flat[k]))

    def collect(self):
        if not self.pending:
            raise RuntimeError("EpeGather: nothing submitted")
        work, buf = self.pending.pop(0)
        if not self.collective:
            # a slot result is a window of the payload buffer that comes round again two steps later
            if self.keep_history:
                buf = buf.clone()
            return buf.unsqueeze(0), buf
        work.wait()
        out = buf.view(self.world, self.L + 1)
        per_rank, weights = out[:, :-1], out[:, -1:]
        return per_rank.clone(), (per_rank * weights).sum(dim=0) / weights.sum()

    def outstanding(self):
        return len(self.pending)


def timed_steps(run_step, gather, steps, warmup, device, sync=None):
    """THE step loop of bench.py (kept here so that the CPU suite can run it under gloo with a stub
    forward): `warmup` untimed steps, a drain, then exactly `steps` steps bracketed by a barrier and a
    device synchronisation on both sides; the last collective is drained INSIDE the timed region.

    run_step(k) runs step k's forward and returns either the step's EPE vector (a tensor, copied into the
    payload) or the payload slot (int) it has already written in stream order.  Every step submits its
    all-gather and, once two are outstanding, collects the older one -- the exchange of step k travels
    while step k+1 computes.  -> (elapsed seconds, MAX over ranks; list of collected (per_rank, mean) in
    step order for the timed steps -- None for steps whose result a history-less gather no longer holds)."""
    if sync is None:
        sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)

    def step(k):
        r = run_step(k)
        if isinstance(r, int):
            gather.submit(slot=r)
        else:
            gather.submit(r)
        return gather.collect() if gather.outstanding() > 1 else None

    def drain():
        out = []
        while gather.outstanding():
            out.append(gather.collect())
        return out

    import time
    for k in range(warmup):
        step(k)
    drain()
    barrier()
    sync()
    t0 = time.perf_counter()
    results = []
    for k in range(warmup, warmup + steps):
        r = step(k)
        if r is not None:
            results.append(r)
    results += drain()
    barrier()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    if not getattr(gather, "keep_history", True):
        # windows of two payload buffers: only the last two steps' results are still what they were
        results = [None] * max(0, len(results) - 2) + results[-2:]
    return elapsed, results


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
