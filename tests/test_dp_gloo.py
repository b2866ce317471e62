"""GradReducer (amk/dp.py) on two CPU processes over gloo: the N>1 path of bench.py.

Checks against a single-process computation of the same global batch: averaged gradients,
parameters with no gradient (the reference's SwitchHead W_d case), gradient accumulation
(sync=False micro-steps) and identical parameters after optimizer steps on both ranks.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.unused = nn.Linear(3, 3)  # never reached: no gradient in any step

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _data(rank_count=2, per_rank=5):
    g = torch.Generator().manual_seed(7)
    return torch.randn(rank_count * per_rank, 8, generator=g), torch.randn(rank_count * per_rank, 4, generator=g)


def _worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk.dp import GradReducer

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # deliberately different init: broadcast must fix it
    net = Net()
    red = GradReducer(net.parameters(), bucket_bytes=256)  # tiny buckets -> several of them
    red.broadcast_parameters()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, y = _data(world)
    xs, ys = x.chunk(world)[rank], y.chunk(world)[rank]

    # step 1: plain synchronous step
    red.begin(sync=True)
    ((net(xs) - ys) ** 2).mean().backward()
    red.finish()
    # a parameter that got no gradient has .grad None after finish(), as in the reference (optimizers skip it)
    assert net.unused.weight.grad is None and net.unused.bias.grad is None
    assert [id(p) for p in red.unused_parameters()] == [id(p) for b in red.buckets for p in b.params
                                                         if p is net.unused.weight or p is net.unused.bias]
    assert red.launch_order == list(range(len(red.buckets)))  # buckets leave in index order on every rank
    g1 = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters()}
    opt.step()
    red.zero_grad()
    # steps 2+3: accumulate two micro-batches, communicate on the second only
    h = xs.shape[0] // 2
    red.begin(sync=False)
    (((net(xs[:h]) - ys[:h]) ** 2).mean() / 2).backward()
    red.finish()
    red.begin(sync=True)
    (((net(xs[h:2 * h]) - ys[h:2 * h]) ** 2).mean() / 2).backward()
    red.finish()
    g2 = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters()}
    opt.step()
    red.zero_grad()
    torch.save(dict(g1=g1, g2=g2, params={n: p.detach().clone() for n, p in net.named_parameters()},
                    nbuckets=len(red.buckets)), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(120)
def test_two_rank_reducer_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert r0["nbuckets"] > 1

    # single-process reference: rank 0's initial weights, mean over ranks of per-rank losses
    torch.manual_seed(100)
    net = Net()
    x, y = _data(world)
    loss = sum(((net(xs) - ys) ** 2).mean() for xs, ys in zip(x.chunk(world), y.chunk(world))) / world
    loss.backward()
    for n, p in net.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(r0["g1"][n], want, atol=1e-6), n
        assert torch.equal(r0["g1"][n], r1["g1"][n]), n
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.step()
    opt.zero_grad()
    h = x.chunk(world)[0].shape[0] // 2
    loss = 0
    for xs, ys in zip(x.chunk(world), y.chunk(world)):
        loss = loss + (((net(xs[:h]) - ys[:h]) ** 2).mean() / 2 + ((net(xs[h:2 * h]) - ys[h:2 * h]) ** 2).mean() / 2)
    (loss / world).backward()
    for n, p in net.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(r0["g2"][n], want, atol=1e-6), n
    opt.step()
    for n, p in net.named_parameters():
        assert torch.allclose(r0["params"][n], p, atol=1e-6), n
        assert torch.equal(r0["params"][n], r1["params"][n]), n


# ---------------------------------------------------------------------------------------------
# The whole GAN train step (amk/train.py VQGANTrainStep: both reducers, the phase toggling of
# requires_grad, the gradient penalty's double backward, clip, Adam) on two ranks.  The generator is a
# CPU stand-in with the ViT-VQGAN call contract (imgs -> (reconstruction, codebook loss)) -- the real one
# runs HIP kernels only -- the discriminator is the real PatchGAN.
class _TinyGenerator(nn.Module):
    def __init__(self):
        super().__init__()
        self.enc = nn.Conv2d(3, 8, 3, padding=1)
        self.dec = nn.Conv2d(8, 3, 3, padding=1)
        self.never_used = nn.Parameter(torch.ones(5))  # no gradient in any step (the reference's W_d case)

    def forward(self, x):
        z = torch.tanh(self.enc(x))
        return self.dec(z), (z ** 2).mean()


def _step_worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5 + rank)  # different init per rank: the trainer's broadcast must fix it
    gen, discr = _TinyGenerator(), NLayerDiscriminator(3, 8, 2)
    if world == 1:  # the single-process reference starts from rank 0's weights
        torch.manual_seed(5)
        gen, discr = _TinyGenerator(), NLayerDiscriminator(3, 8, 2)
    trainer = VQGANTrainStep(gen, discr, lr=1e-2, warmup_steps=2, decay_steps=10, bucket_bytes=1 << 10)
    g = torch.Generator().manual_seed(11)
    imgs = torch.rand(4, 3, 32, 32, generator=g)
    eta = torch.rand(4, 1, 1, 1, generator=g)
    # BatchNorm uses local batch statistics (no SyncBN in the reference): the single-process run evaluates
    # each phase on the two half batches one after the other as accumulation micro-steps (gradients summed,
    # each loss halved), which is the arithmetic of two ranks averaging their gradients.
    logs = []
    halves = (slice(0, 2), slice(2, 4))
    for _ in range(3):
        if world > 1:
            logs.append(trainer.step(imgs[halves[rank]], eta=eta[halves[rank]]))
        else:
            trainer._set_lr()
            trainer.d_phase(imgs[halves[0]], sync=False, accum_steps=2, eta=eta[halves[0]])
            trainer.d_phase(imgs[halves[1]], sync=True, accum_steps=2, eta=eta[halves[1]])
            trainer.g_phase(imgs[halves[0]], sync=False, accum_steps=2)
            out = trainer.g_phase(imgs[halves[1]], sync=True, accum_steps=2)
            trainer.global_step += 1
            logs.append(out)
    torch.save(dict(gen={n: p.detach().clone() for n, p in gen.named_parameters()},
                    discr={n: p.detach().clone() for n, p in discr.named_parameters()},
                    loss=[float(l["loss"]) for l in logs], nb=(len(trainer.g_red.buckets), len(trainer.d_red.buckets))),
               os.path.join(out_dir, f"step_w{world}_r{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_train_step_matches_accumulated_single_process(tmp_path):
    mp.spawn(_step_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    _step_worker(0, 1, 0, str(tmp_path))
    r0 = torch.load(tmp_path / "step_w2_r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "step_w2_r1.pt", weights_only=True)
    one = torch.load(tmp_path / "step_w1_r0.pt", weights_only=True)
    assert r0["nb"][0] > 1 and r0["nb"][1] > 1  # several buckets per reducer
    for part in ("gen", "discr"):
        for n in r0[part]:
            assert torch.equal(r0[part][n], r1[part][n]), (part, n)  # ranks stay in lockstep
            assert torch.allclose(r0[part][n], one[part][n], rtol=2e-4, atol=2e-6), (part, n)  # summation order only
    assert torch.equal(r0["gen"]["never_used"], torch.ones(5))


# ---------------------------------------------------------------------------------------------
class GatedNet(nn.Module):
    """`head` (registered LAST, so it sits in bucket 0) never gets a gradient; `skip` is bypassed on rank 1 only."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.skip = nn.Linear(16, 16)
        self.b = nn.Linear(16, 4)
        self.head = nn.Linear(3, 3)

    def forward(self, x, use_skip):
        h = torch.tanh(self.a(x))
        if use_skip:
            h = h + self.skip(h)
        return self.b(h)


def _worker_static(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk.dp import GradReducer

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    net = GatedNet()
    red = GradReducer(net.parameters(), bucket_bytes=256)
    assert any(p is net.head.bias for p in red.buckets[0].params)   # a never-used parameter is in the first bucket
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, y = _data(world)
    xs, ys = x.chunk(world)[rank], y.chunk(world)[rank]
    early = []
    for step in range(3):
        red.begin(sync=True)
        ((net(xs, use_skip=rank == 0) - ys) ** 2).mean().backward()
        early.append(list(red.launch_order))        # buckets that left BEFORE finish()
        red.finish()
        # `skip` fired on rank 0 only: it must count as used on BOTH ranks (rank 1 applies the averaged gradient)
        assert net.skip.weight.grad is not None and net.head.weight.grad is None
        opt.step()
        red.zero_grad()
    # step 0 records the static set (nothing can leave before bucket 0 is complete, and `head` never completes it);
    # from step 1 on the buckets whose used parameters all fired leave during backward
    assert early[0] == [] and len(early[1]) > 0 and early[1][0] == 0, early
    torch.save({n: p.detach().clone() for n, p in net.named_parameters()}, os.path.join(out_dir, f"static{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_static_unused_parameters_do_not_block_and_are_decided_globally(tmp_path):
    world = 2
    mp.spawn(_worker_static, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "static0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "static1.pt", weights_only=True)
    for n in r0:
        assert torch.equal(r0[n], r1[n]), f"{n}: the replicas drifted apart"
    torch.manual_seed(5)
    fresh = GatedNet()
    assert not torch.equal(r0["skip.weight"], fresh.skip.weight)   # updated (on both ranks) ...
    assert torch.equal(r0["head.weight"], fresh.head.weight)       # ... and the never-used one untouched


# ---------------------------------------------------------------------------------------------
# The classifier train step (amk/train.py ClassifierTrainStep = trainers/vit.py:66-76: CE, AdamW, clip, cosine schedule)
# on two ranks.  The network is a CPU stand-in with ViTMoE's distinguishing property under data parallelism: a gate
# `W_d` that only SELECTS (top-k indices, no values used: models/switchhead_attention.py:80-87), so it never receives a
# gradient -- and it is registered last, i.e. it sits in the FIRST bucket to leave.  (The real ViTMoE runs HIP kernels
# only: tests/test_dp_rccl_gpu.py runs this same check on it, two ranks sharing the GPU.)
class _RoutedNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Linear(12, 16)
        self.experts = nn.Parameter(torch.randn(4, 16, 16) * 0.2)
        self.head = nn.Linear(16, 5)
        self.W_d = nn.Linear(16, 4, bias=False)

    def forward(self, x):
        h = torch.tanh(self.embed(x.flatten(1)))
        sel = self.W_d(h).topk(2, dim=-1).indices            # values discarded: W_d gets no gradient
        out = torch.zeros_like(h)
        for e in range(4):                                     # un-weighted sum over the selected experts
            hit = (sel == e).any(-1)
            out = out + hit.unsqueeze(-1) * (h @ self.experts[e])
        return self.head(out)


def _cls_worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk.train import ClassifierTrainStep

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(21 + rank)   # different init per rank: the step's broadcast must fix it
    net = _RoutedNet()
    # world 1: the same global batch as two accumulation micro-steps (mean of the two half-batch losses = the average of
    # the two ranks' gradients)
    ts = ClassifierTrainStep(net, lr=1e-2, warmup_steps=1, total_steps=8, max_grad_norm=1.0, bucket_bytes=512,
                             accum_steps=1 if world > 1 else 2)
    assert len(ts.red.buckets) > 2 and any(p is net.W_d.weight for p in ts.red.buckets[0].params)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 8, 3, 2, 2, generator=g)
    y = torch.randint(0, 5, (3, 8), generator=g)
    early, w0 = [], net.W_d.weight.detach().clone()
    finish = ts.red.finish

    def spying_finish(*a, **k):
        if ts.red.sync_step:
            early.append(list(ts.red.launch_order))   # buckets that left during backward, before finish()
        out = finish(*a, **k)
        if ts.red.sync_step:
            assert net.W_d.weight.grad is None         # what the optimizer sees: skipped, as in the reference
            assert net.head.weight.grad is not None
        return out

    ts.red.finish = spying_finish
    for s in range(3):
        if world > 1:
            ts.step(x[s, 4 * rank: 4 * rank + 4], y[s, 4 * rank: 4 * rank + 4])
        else:
            ts.step(x[s, :4], y[s, :4])
            ts.step(x[s, 4:], y[s, 4:])
            ts._lr_arg = s            # (the two-rank run's scheduler index counts optimizer steps, this one micro-steps)
    assert torch.equal(net.W_d.weight, w0)   # skipped like .grad None: no weight decay either
    torch.save(dict(params={n: p.detach().clone() for n, p in net.named_parameters()}, early=early),
               os.path.join(out_dir, f"cls_w{world}_r{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_classifier_step_with_a_gradient_free_gate(tmp_path):
    mp.spawn(_cls_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    _cls_worker(0, 1, 0, str(tmp_path))
    r0 = torch.load(tmp_path / "cls_w2_r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "cls_w2_r1.pt", weights_only=True)
    one = torch.load(tmp_path / "cls_w1_r0.pt", weights_only=True)
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n                                    # replicas in lockstep
        assert torch.allclose(r0["params"][n], one["params"][n], rtol=2e-4, atol=2e-6), n          # = accumulated run
    # step 0 records which parameters never fire (nothing can leave before bucket 0, and W_d never completes it);
    # from step 1 on bucket 0 leaves during backward
    assert r0["early"][0] == [] and len(r0["early"][1]) >= 1 and r0["early"][1][0] == 0, r0["early"]
    assert len(r0["early"][2]) >= 1
