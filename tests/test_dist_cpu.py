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
