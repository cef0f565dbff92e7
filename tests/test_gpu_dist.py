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
