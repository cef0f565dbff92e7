"""CPU suite: the N>1 path with world_size-2 gloo processes (sharding + the
all-gather of the per-level EPE vector)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from qpwcnet_amd import dist as qdist


def test_shard_range_partitions_every_pair_once():
    for n in (64, 8, 7, 1, 0):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                lo, hi = qdist.shard_range(n, r, world)
                got += list(range(lo, hi))
            assert got == list(range(n))
    assert qdist.shard_range(64, 3, 8) == (24, 32)      # config 3: 8 pairs per GPU
    with pytest.raises(ValueError):
        qdist.shard_range(8, 2, 2)


def test_single_process_gather_is_identity():
    v = torch.arange(6, dtype=torch.float32)
    per_rank, mean = qdist.gather_epe(v)
    assert per_rank.shape == (1, 6) and torch.equal(mean, v)
    assert qdist.max_over_ranks(1.5, "cpu") == 1.5


def test_single_process_pipelined_gather():
    eg = qdist.EpeGather(6, "cpu")
    v = torch.arange(6, dtype=torch.float32)
    eg.submit(v)
    eg.submit(v + 1)
    pr, mean = eg.collect()
    assert pr.shape == (1, 6) and torch.equal(mean, v)
    assert torch.equal(eg.collect()[1], v + 1) and eg.outstanding() == 0
    with pytest.raises(RuntimeError):
        eg.collect()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(rank))
    from oracle import np_ref
    from qpwcnet_amd import dist as qd
    w, r, _ = qd.init("gloo")
    assert (w, r) == (world, rank)
    # 5 pairs over 2 ranks -> 3 + 2; every rank scores its own shard with the oracle EPE
    rng = np.random.default_rng(0)
    true = rng.standard_normal((5, 6, 8, 2)).astype(np.float32)
    pred = rng.standard_normal((5, 6, 8, 2)).astype(np.float32)
    lo, hi = qd.shard_range(5, rank, world)
    local = torch.tensor([np_ref.epe_error(true[lo:hi] * s, pred[lo:hi] * s) for s in (1, 2, 3)],
                         dtype=torch.float32)
    per_rank, mean = qd.gather_epe(local, hi - lo)
    full = [np_ref.epe_error(true * s, pred * s) for s in (1, 2, 3)]
    # the pipelined form bench.py uses: two collectives in flight, results come back in order
    eg = qd.EpeGather(3, "cpu", n_local=hi - lo)
    eg.submit(local)
    eg.submit(local * 2)
    pr_a, mean_a = eg.collect()
    eg.submit(local * 3)
    pr_b, mean_b = eg.collect()
    pr_c, mean_c = eg.collect()
    assert eg.outstanding() == 0
    assert torch.equal(pr_a, per_rank) and torch.allclose(mean_a, mean)
    assert torch.allclose(mean_b, 2 * mean) and torch.allclose(mean_c, 3 * mean) and torch.equal(pr_c, 3 * per_rank)
    t = qd.max_over_ranks(float(rank + 1), "cpu")
    qd.barrier()
    q.put((rank, per_rank.numpy(), mean.numpy(), np.asarray(full, np.float32), t))
    dist.destroy_process_group()


def test_world_size_2_gloo_allgather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    (_, pr0, m0, full0, t0), (_, pr1, m1, full1, t1) = res
    assert pr0.shape == (2, 3)
    np.testing.assert_array_equal(pr0, pr1)             # every rank holds every rank's vector
    np.testing.assert_allclose(m0, full0, rtol=1e-5)    # weighted mean == EPE of the whole batch
    np.testing.assert_allclose(m1, full1, rtol=1e-5)
    assert t0 == t1 == 2.0                              # max over ranks


# ---- bench.py's real step / drain loop (qpwcnet_amd.dist.timed_steps) under world_size-2 gloo ----------
def _loop_worker(rank, world, port, q, in_place):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(rank))
    from qpwcnet_amd import dist as qd
    qd.init("gloo")
    n_local = 3 if rank == 0 else 5                      # uneven shards: the mean must be weighted
    gather = qd.EpeGather(6, "cpu", n_local=n_local)
    calls = []

    def stub_forward(k):
        """Stand-in for one forward of step k: a vector that names its step, rank and level."""
        return torch.arange(6, dtype=torch.float32) + 10.0 * k + 1000.0 * rank

    def run_step(k):
        calls.append(k)
        if in_place:    # what the captured EPE reduction does on the GPU: write the payload slot itself
            s = gather.next_slot()
            gather.payload_view(s).copy_(stub_forward(k))
            return s
        return stub_forward(k)

    steps, warmup = 7, 3
    elapsed, results = qd.timed_steps(run_step, gather, steps, warmup, "cpu")
    assert calls == list(range(warmup + steps)) and gather.outstanding() == 0
    q.put((rank, elapsed, [(pr.numpy(), m.numpy()) for pr, m in results]))
    dist.destroy_process_group()


@pytest.mark.parametrize("in_place", [False, True], ids=["copied-payload", "payload-written-in-place"])
def test_bench_step_loop_world_size_2_gloo(in_place):
    """The loop bench.py times (warm-up, drain, barrier, K steps with the all-gather of step k collected
    during step k+1, final drain inside the timed region, max over ranks), with a stub forward: every
    timed step's result comes back once, in step order, holding BOTH ranks' vectors of THAT step, and the
    drained last result is the last step's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q, in_place)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, r0), (_, t1, r1) = res
    assert t0 == t1 and t0 > 0                            # MAX over ranks, same on every rank
    steps, warmup = 7, 3
    assert len(r0) == len(r1) == steps
    base = np.arange(6, dtype=np.float32)
    for i in range(steps):
        k = warmup + i
        expect = np.stack([base + 10.0 * k, base + 10.0 * k + 1000.0])
        for pr, mean in (r0[i], r1[i]):
            np.testing.assert_array_equal(pr, expect)     # ordering: result i belongs to step warmup + i
            np.testing.assert_allclose(mean, (3 * expect[0] + 5 * expect[1]) / 8, rtol=1e-6)


def test_epe_gather_slot_protocol():
    eg = qdist.EpeGather(3, "cpu")
    assert eg.next_slot() == 0
    eg.payload_view(0).copy_(torch.tensor([1.0, 2.0, 3.0]))
    eg.submit(slot=0)
    with pytest.raises(RuntimeError):
        eg.submit(slot=0)                                 # slot 1 is next
    eg.payload_view(1).copy_(torch.tensor([4.0, 5.0, 6.0]))
    eg.submit(slot=1)
    with pytest.raises(RuntimeError):
        eg.submit(torch.zeros(3))                         # at most two outstanding
    assert torch.equal(eg.collect()[1], torch.tensor([1.0, 2.0, 3.0]))
    assert torch.equal(eg.collect()[1], torch.tensor([4.0, 5.0, 6.0]))
    with pytest.raises(ValueError):
        eg.submit(torch.zeros(3), slot=0)


def test_single_process_step_loop_keeps_every_steps_result():
    """with the payload written in place and no process group, collect() used to hand out
    views of the two payload buffers, so only the last two entries of timed_steps' history were valid."""
    gather = qdist.EpeGather(6, "cpu", n_local=4)
    base = torch.arange(6, dtype=torch.float32)

    def run_step(k):
        s = gather.next_slot()
        gather.payload_view(s).copy_(base + 10.0 * k)
        return s

    steps, warmup = 6, 2
    assert gather.keep_history                                   # CPU tensors: copies cost nothing extra
    elapsed, results = qdist.timed_steps(run_step, gather, steps, warmup, "cpu")
    assert elapsed > 0 and len(results) == steps
    for i, (per_rank, mean) in enumerate(results):
        assert torch.equal(mean, base + 10.0 * (warmup + i)) and torch.equal(per_rank[0], mean)
    # the history-less form (the default for a single process on a GPU: no extra launch per step): the entries that
    # no longer hold their step's values come back as None, never as aliases of a later step
    gather = qdist.EpeGather(6, "cpu", n_local=4, keep_history=False)
    elapsed, results = qdist.timed_steps(run_step, gather, steps, warmup, "cpu")
    assert len(results) == steps and results[:-2] == [None] * (steps - 2)
    for i in (steps - 2, steps - 1):
        assert torch.equal(results[i][1], base + 10.0 * (warmup + i))


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no torchrun environment: the parent (which touches no GPU) starts two
    fresh ranks through torch.distributed.run, rank 0's line comes back on stdout, the exit code is the
    children's.  --stub-forward: the real dist.timed_steps loop under gloo with a stub forward (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-forward",
                        "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] is None
    k = 2 + 5 - 1                                         # the drained last result is the last step's
    assert d["last_step_per_rank"] == [[10.0 * k + i for i in range(6)], [10.0 * k + 1000.0 + i for i in range(6)]]
    assert d["allgather"] == {"ranks_in_allgather": 2, "levels_per_rank": 6, "backend": "gloo"}
    assert d["scaling"] == "weak" and d["global_batch"] == 16
    # strong scaling (BASELINE configs[2] is `--gpus 8 --global-batch 64`): 5 pairs over 2 ranks = shards of 3 and 2
    # (dist.shard_range), the mean weighted by the shard sizes, the line says how many ranks the all-gather saw
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-forward",
                        "--global-batch", "5", "--steps", "3", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["global_batch"] == 5 and d["batch_rank0"] == 3
    assert d["allgather"]["ranks_in_allgather"] == 2 and d["allgather"]["levels_per_rank"] == 6
    k = 1 + 3 - 1
    want = [(3 * (10.0 * k + i) + 2 * (10.0 * k + 1000.0 + i)) / 5 for i in range(6)]
    assert all(abs(a - b) < 1e-3 for a, b in zip(d["last_step_mean"], want)), (d["last_step_mean"], want)
    # a failing rank makes the launcher fail: --gpus 2 inside a WORLD_SIZE=3 environment is refused by every rank
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-forward"],
                         env=dict(env, WORLD_SIZE="3", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0
