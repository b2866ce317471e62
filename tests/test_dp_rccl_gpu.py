"""GradReducer over RCCL (backend "nccl") on the one GPU of the test box: a world of one rank.

The two-rank logic is covered on CPU over gloo (tests/test_dp_gloo.py); this test exists so that
the RCCL code path bench.py takes at --gpus N > 1 -- communicator creation, the parameter
broadcast, bucket all-reduces launched from autograd hooks on the side stream, the stream join in
finish() -- has run on the real device before the driver's multi-GPU run.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_grad_reducer_on_rccl(device):
    from amk.dp import GradReducer
    from amk.models import SoftmaxAttention

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        torch.manual_seed(3)
        net = nn.Sequential(SoftmaxAttention(128, 2, 64), nn.Linear(128, 16)).to(device)
        x = torch.randn(2, 40, 128, device=device)
        y = torch.randn(2, 40, 16, device=device)

        def backward():
            ((net(x) - y) ** 2).mean().backward()

        backward()
        want = [p.grad.clone() for p in net.parameters()]
        net.zero_grad(set_to_none=True)

        red = GradReducer(net.parameters(), bucket_bytes=64 << 10, communicate_when_alone=True)
        assert len(red.buckets) > 1 and not red.alone
        red.broadcast_parameters()
        red.begin(sync=True)            # plain step: all-reduce (sum over one rank) / 1
        backward()
        red.finish()
        for p, w in zip(net.parameters(), want):
            assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-6)
        red.zero_grad()
        red.begin(sync=False)           # accumulation pair: communicate on the second micro-step only
        backward()
        red.finish()
        red.begin(sync=True)
        backward()
        red.finish()
        torch.cuda.synchronize()
        for p, w in zip(net.parameters(), want):
            assert torch.allclose(p.grad, 2 * w, rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()
