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


# ---------------------------------------------------------------------------------------------
# Two ranks on the GPU branch of GradReducer (side stream, ready events, stream join, FlatAdam on the
# reduced buckets).  RCCL refuses two ranks on one device, and the test box has one GPU, so the two
# processes share cuda:0 and talk over gloo -- the same rehearsal bench.py offers
# (AMK_REHEARSE_SHARED_GPU=1); everything but the transport is the code the 8-GPU run executes.
def _two_rank_worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk.dp import GradReducer
    from amk.models import SoftmaxAttention
    from amk.optim import FlatAdam

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(50 + rank)  # different init per rank: the broadcast must fix it
        net = nn.Sequential(SoftmaxAttention(128, 2, 64), nn.Linear(128, 16)).to(dev)
        # overlap on the GPU branch: when the gradient of the FIRST-registered parameter lands (the last one backward
        # produces) -- seen by a hook registered BEFORE the reducer's own, so before the reducer has counted it --
        # bucket 0 (the last-registered parameters) must already have left on the side stream
        seen = {}
        first = next(iter(net.parameters()))

        def _note(p):
            seen.setdefault("launched", list(red.launch_order))   # (a hook must return None)

        first.register_post_accumulate_grad_hook(_note)
        red = GradReducer(net.parameters(), bucket_bytes=64 << 10)
        assert red.on_gpu and not red.alone and len(red.buckets) > 1
        opt = FlatAdam(red, lr=1e-2)
        red.broadcast_parameters()
        g = torch.Generator().manual_seed(9)
        x = torch.randn(4, 40, 128, generator=g)[2 * rank:2 * rank + 2].to(dev)
        y = torch.randn(4, 40, 16, generator=g)[2 * rank:2 * rank + 2].to(dev)
        red.begin(sync=True)
        ((net(x) - y) ** 2).mean().backward()
        seen["before_finish"] = list(red.launch_order)
        red.finish()
        grads = [p.grad.detach().cpu().clone() for p in net.parameters()]
        order = list(red.launch_order)
        assert seen["launched"] and seen["launched"][0] == 0, f"bucket 0 had not left when backward reached its last gradient: {seen}"
        assert len(red.buckets) - 1 not in seen["launched"], "the first-registered parameter's bucket cannot have left before its gradient"
        assert len(seen["before_finish"]) == len(red.buckets), "every bucket leaves inside backward when all parameters get gradients"
        opt.step(max_norm=1.0)
        torch.cuda.synchronize()
        torch.save(dict(grads=grads, order=order, params=[p.detach().cpu().clone() for p in net.parameters()]),
                   os.path.join(out_dir, f"gpu_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_share_the_gpu_over_gloo(device, tmp_path):
    import torch.multiprocessing as mp

    from amk.models import SoftmaxAttention

    mp.spawn(_two_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "gpu_rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "gpu_rank1.pt", weights_only=True)
    assert r0["order"] == r1["order"] == sorted(r0["order"])          # buckets leave in index order on both ranks
    # single-process reference from rank 0's initial weights: mean over the two half batches
    torch.manual_seed(50)
    net = nn.Sequential(SoftmaxAttention(128, 2, 64), nn.Linear(128, 16)).to(device)
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(4, 40, 128, generator=g).to(device), torch.randn(4, 40, 16, generator=g).to(device)
    loss = sum(((net(x[i:i + 2]) - y[i:i + 2]) ** 2).mean() for i in (0, 2)) / 2
    loss.backward()
    for a, b, p in zip(r0["grads"], r1["grads"], net.parameters()):
        assert torch.equal(a, b)                                       # both ranks hold the same averaged gradient
        assert torch.allclose(a, p.grad.cpu(), rtol=2e-5, atol=1e-6)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                                       # and take the same optimizer step


def _two_rank_worker_mixed(rank, world, port, out_dir):
    """The mixed-precision training path on two ranks: bf16 autocast, own bf16 GEMMs, gradients written straight into
    the buckets (direct_grads), bf16 parameter copies refreshed by the optimizer kernel."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk.dp import GradReducer
    from amk.models import SoftmaxAttention
    from amk.models.layers import LayerNorm, Linear
    from amk.optim import FlatAdam

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(70 + rank)

        class Block(nn.Module):
            def __init__(self):
                super().__init__()
                self.norm, self.attn, self.out = LayerNorm(128), SoftmaxAttention(128, 2, 64), Linear(128, 16)

            def forward(self, x):
                return self.out(x + self.attn(self.norm(x)))

        net = Block().to(dev)
        red = GradReducer(net.parameters(), bucket_bytes=64 << 10, direct_grads=True)
        opt = FlatAdam(red, lr=1e-2, bf16_shadow=True)
        red.broadcast_parameters()
        opt.refresh_shadow()          # (the broadcast wrote the parameters behind the optimizer's back)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(4, 40, 128, generator=g)[2 * rank:2 * rank + 2].to(dev)
        y = torch.randn(4, 40, 16, generator=g)[2 * rank:2 * rank + 2].to(dev)
        direct = 0
        for _ in range(3):
            red.begin(sync=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = ((net(x).float() - y) ** 2).mean()
            loss.backward()
            direct += sum(sum(b.direct) for b in red.buckets)
            red.finish()
            opt.step(max_norm=1.0)
        torch.cuda.synchronize()
        assert direct > 0, "no gradient was written in place"
        for p in net.parameters():
            assert torch.isfinite(p).all() and torch.equal(p._amk_bf16, p.detach().to(torch.bfloat16))
        torch.save([p.detach().cpu().clone() for p in net.parameters()], os.path.join(out_dir, f"mixed_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_mixed_precision_direct_gradients(device, tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_two_rank_worker_mixed, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "mixed_rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "mixed_rank1.pt", weights_only=True)
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)      # three optimizer steps later the replicas are still identical


def _bench_line(extra_env, batch=2):
    """bench.py as a child process (a world of one rank that communicates over RCCL anyway); returns (exit code, JSON line)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMK_BENCH_RCCL_ALONE="1", AMK_MIOPEN_FIND="0", AMK_TUNABLEOP="0", **extra_env)
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", str(batch),
                        "--no-cpu-baseline", "--no-kernels", "--no-variants"], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r.returncode, json.loads(lines[0])


def test_bench_times_the_eager_step_first_and_survives_a_hung_capture():
    """Ranks that communicate: bench.py times the eager step, then attempts the captured one under a deadline.  With the
    attempt stuck (test hook) the eager line must still come out, exit code 0; without the hook both modes are in the line."""
    rc, line = _bench_line({"AMK_BENCH_FAKE_HANG": "1", "AMK_DP_GRAPH_DEADLINE": "15", "AMK_DP_GRAPH": "1"})
    assert rc == 0
    assert line["step_launch"] == "eager" and "did not finish" in line["graph_decision"]
    assert line["value"] > 0 and line["dp_allreduce"].startswith("side stream")
    rc, line = _bench_line({"AMK_DP_GRAPH": "1"})
    assert rc == 0
    assert set(line["dp_step_modes"]) == {"eager", "graph"}
    assert line["value"] > 0
