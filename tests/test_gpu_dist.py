"""The RCCL side of the one collective of the path, as far as a one-GPU box can exercise it: a process
group of a single rank on backend "nccl" (= RCCL on ROCm) running the asynchronous all-gather pattern
bench.py uses (two collectives in flight behind compute, results collected one step late).  The
world_size-2 logic runs on gloo in tests/test_dist_cpu.py; the 8-GPU run is the driver's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from qpwcnet_amd import dist as qdist

pytestmark = pytest.mark.gpu


def test_async_epe_gather_on_rccl_single_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    assert not dist.is_initialized()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        eg = qdist.EpeGather(6, dev, n_local=8)
        assert eg.collective and eg.world == 1
        src = torch.arange(6, dtype=torch.float32, device=dev)
        work = torch.zeros(1 << 20, device=dev)
        got = []
        for k in range(5):
            work.add_(1.0)                       # stand-in for the step's compute
            e = src + float(k)                   # the step's EPE vector (overwritten every step)
            eg.submit(e)
            if eg.outstanding() > 1:
                got.append(eg.collect())
        while eg.outstanding():
            got.append(eg.collect())
        torch.cuda.synchronize()
        assert len(got) == 5
        for k, (per_rank, mean) in enumerate(got):
            assert per_rank.shape == (1, 6)
            assert torch.equal(per_rank[0].cpu(), torch.arange(6, dtype=torch.float32) + k)
            assert torch.allclose(mean.cpu(), per_rank[0].cpu())
        with pytest.raises(RuntimeError):
            eg.collect()
    finally:
        dist.destroy_process_group()


def test_two_graphs_one_pool_write_the_payload_in_place_on_rccl():
    """bench.py's N > 1 step on a single-rank RCCL group: two hipGraphs of the same forward over one memory
    pool, replayed alternately, whose captured EPE reductions write all-gather payload slot 0 / 1 directly
    (no per-step copy), driven by the real loop (qpwcnet_amd.dist.timed_steps)."""
    from qpwcnet_amd import metrics, synth
    from qpwcnet_amd.pwcnet import GraphedForward, build_flower
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    assert not dist.is_initialized()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        hw, B = (64, 128), 2
        model = build_flower(True, hw, "channels_last", weights=synth.make_weights(42, hw), device=dev)
        batches = []
        for k in range(3):
            p, g = synth.make_frames(B, hw[0], hw[1], seed=50 + k)
            batches.append((torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev)))
        shapes = [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)]
        gt_pyr = metrics.multiscale_ground_truth(batches[0][1], shapes)
        with torch.no_grad():
            expect = [metrics.per_level_epe(gt_pyr, model(p)).cpu() for p, _ in batches]
        gather = qdist.EpeGather(6, dev, n_local=B)
        assert gather.collective
        g0 = GraphedForward(model, batches[0][0], epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl, out=gather.payload_view(0)))
        g1 = GraphedForward(model, batches[0][0], epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl, out=gather.payload_view(1)),
                            share_with=g0)
        graphs = [g0, g1]

        def run_step(k):
            slot = gather.next_slot()
            graphs[slot].replay(batches[k % 3][0])
            return slot

        steps, warmup = 7, 2
        elapsed, results = qdist.timed_steps(run_step, gather, steps, warmup, dev)
        assert elapsed > 0 and len(results) == steps
        for i, (per_rank, mean) in enumerate(results):
            k = warmup + i
            assert per_rank.shape == (1, 6)
            torch.testing.assert_close(per_rank[0].cpu(), expect[k % 3], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(mean.cpu(), expect[k % 3], rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()
