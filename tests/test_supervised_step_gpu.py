"""The single-model train steps of amk/train.py on the GPU: ClassifierTrainStep (trainers/vit.py:29-46,66-75: ViT and
ViTMoE, BASELINE.json configs[1] / configs[3]) and MaskedTokenTrainStep (trainers/muse.py:48-97: the Muse decoder over a
frozen vq, configs[4]).

  * FlatAdam as AdamW with per-parameter weight decay and a device-resident learning rate (capturable) against torch.optim.AdamW;
  * a reduced ViTMoE through the step: `W_d` (models/switchhead_attention.py:80-87: no gradient) skipped, the rest trained;
  * the step CAPTURED WITH ITS RCCL COLLECTIVES INSIDE (a world of one rank that communicates anyway) replays to the
    bits of the eager step -- the data-parallel step as one HIP-graph launch;
  * two ranks sharing the GPU over gloo on the reduced ViTMoE: identical replicas after three AdamW steps, equal to an
    accumulated single-process run, bucket 0 (which holds a W_d) leaves during backward from the second step on;
  * configs[3] at its own size through reducer + FlatAdam (3.9 GB of flat state).
"""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

from util import assert_close

pytestmark = pytest.mark.gpu

SMALL_MOE = dict(dim=128, image_size=64, patch_size=16, n_heads=2, d_head=64, depth=2, n_experts=4, sel_experts=2,
                 dropout=0.0, num_classes=10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("capturable", [False, True])
def test_flat_adamw_param_groups_match_torch(device, capturable):
    """AdamW with the two parameter groups of trainers/muse.py:48-58 (weight decay / none on biases) and a learning rate
    that changes every step; capturable=True keeps lr, the step counts and lr * wd on the device."""
    from amk.dp import GradReducer
    from amk.optim import FlatAdam

    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(37, 53), nn.Tanh(), nn.Linear(53, 300), nn.Tanh(), nn.Linear(300, 7)).to(device)
    ref = copy.deepcopy(net)
    red = GradReducer(net.parameters(), bucket_bytes=32 << 10)
    biases = [p for n, p in net.named_parameters() if "bias" in n]
    opt = FlatAdam(red, lr=3e-3, betas=(0.9, 0.96), weight_decay=0.1, decoupled=True, capturable=capturable, no_decay=biases)
    ropt = torch.optim.AdamW([dict(params=[p for n, p in ref.named_parameters() if "bias" not in n], weight_decay=0.1),
                              dict(params=[p for n, p in ref.named_parameters() if "bias" in n], weight_decay=0.0)],
                             lr=3e-3, betas=(0.9, 0.96))
    g = torch.Generator().manual_seed(1)
    for step in range(5):
        lr = 3e-3 * (step + 1) / 5
        x = torch.randn(16, 37, generator=g).to(device)
        red.begin(True)
        net(x).pow(2).mean().backward()
        red.finish()
        ref(x).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.7)
        for grp in ropt.param_groups:
            grp["lr"] = lr
        ropt.step()
        ropt.zero_grad()
        opt.step(max_norm=0.7, lr=lr)
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert_close(p, q, 2e-6, n)


def _moe_batch(device, n=8, seed=5):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 3, 64, 64, generator=g).to(device), torch.randint(0, 10, (n,), generator=g).to(device))


def test_classifier_step_reduced_vit_moe(device):
    """Three AdamW steps on a reduced ViTMoE: every W_d stays exactly as initialised (no gradient -> no update, no weight
    decay), everything the reference trains moves, the loss is finite, the bucket holding a W_d does not wait for it."""
    from amk.models import ViTMoE
    from amk.train import ClassifierTrainStep

    torch.manual_seed(0)
    model = ViTMoE(**SMALL_MOE).to(device)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    ts = ClassifierTrainStep(model, lr=1e-3, warmup_steps=1, total_steps=10, bucket_bytes=256 << 10)
    assert len(ts.red.buckets) > 2
    imgs, labels = _moe_batch(device)
    for _ in range(3):
        loss = ts.step(imgs, labels)
    assert torch.isfinite(loss)
    moved = 0
    for n, p in model.named_parameters():
        if "W_d" in n:
            assert torch.equal(p, before[n]), n
        else:
            moved += int(not torch.equal(p, before[n]))
    assert moved >= sum(1 for n in before if "W_d" not in n) - 1     # (LambdaLR's first step has lr 0: two real updates follow)
    unused = {id(p) for p in ts.red.static_unused_parameters()}
    assert unused == {id(p) for n, p in model.named_parameters() if "W_d" in n}


def _rccl_world_of_one(device):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", [True, False])
def test_captured_dp_step_with_collectives_replays_like_eager(device, overlap):
    """The data-parallel step as ONE HIP-graph replay: RCCL all-reduces (issued from the autograd hooks on the reducer's side
    stream) are captured with the forward, backward, clip and AdamW.  World of one rank with communicate_when_alone=True:
    the collectives are real RCCL launches.  The replayed steps must equal the eager steps to the bits (reproducible
    attention backward), on the reduced ViTMoE.  overlap=False: the all-reduces sit on the compute stream (a single-stream
    graph: what bench.py captures for the ViT-VQGAN step, whose 90 MB of gradients do not need the overlap)."""
    from amk import ops
    from amk.models import ViTMoE
    from amk.train import ClassifierTrainStep

    _rccl_world_of_one(device)
    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        torch.manual_seed(0)
        base = ViTMoE(**SMALL_MOE).to(device)
        imgs, labels = _moe_batch(device)
        runs = []
        for graphed in (False, True):
            model = copy.deepcopy(base)
            ts = ClassifierTrainStep(model, lr=1e-3, warmup_steps=2, total_steps=20, bucket_bytes=256 << 10, capturable=True,
                                     communicate_when_alone=True, overlap=overlap)
            assert not ts.red.alone and ts.red.avg_in_collective and len(ts.red.buckets) > 2 and ts.red.overlap == overlap
            losses = []
            if graphed:
                ts.capture(imgs, labels, warmup=2)         # two real steps, then the capture
            else:
                losses += [ts.step(imgs, labels), ts.step(imgs, labels)]
            for _ in range(4):
                losses.append(ts.step(imgs, labels).clone())
            if graphed:
                assert ts._graph is not None
            torch.cuda.synchronize()
            runs.append((losses[-4:], [p.detach().clone() for p in model.parameters()], ts.global_step))
        (l0, p0, s0), (l1, p1, s1) = runs
        assert s0 == s1 == 6
        for a, b in zip(l0, l1):
            assert torch.equal(a, b), (a, b)
        for a, b in zip(p0, p1):
            assert torch.equal(a, b)
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_captured_gan_step_with_collectives_replays_like_eager(device):
    """The same for VQGANTrainStep (two reducers, two optimizers, the gradient penalty's double backward)."""
    from amk import ops
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    _rccl_world_of_one(device)
    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        torch.manual_seed(0)
        cfg = dict(dim=64, img_size=32, patch_size=4, n_heads=2, d_head=64, depth=2, mlp_dim=96, dropout=0.0)
        gen0 = ViTVQGAN(cfg, dict(codebook_size=256, codebook_dim=32)).to(device)
        dis0 = NLayerDiscriminator(3, 8, 2).to(device)
        imgs = torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(2)).to(device)
        runs = []
        for graphed in (False, True):
            gen, dis = copy.deepcopy(gen0), copy.deepcopy(dis0)
            tr = VQGANTrainStep(gen, dis, lr=1e-3, warmup_steps=2, decay_steps=50, bucket_bytes=64 << 10, capturable=True,
                                communicate_when_alone=True)
            torch.manual_seed(123)                     # the gradient penalty's eta: same draws in both runs ...
            if graphed:
                tr.capture(imgs, warmup=2)
            else:
                tr.step(imgs), tr.step(imgs)
            logs = [tr.step(imgs) for _ in range(3)]
            torch.cuda.synchronize()
            runs.append(([float(l["loss"]) for l in logs], [float(l["d_loss"]) for l in logs], tr.global_step))
        (g0, d0, s0), (g1, d1, s1) = runs
        assert s0 == s1 == 5
        # ... but a replayed graph re-uses the eta it captured, so only the step structure is compared here: finite,
        # decreasing-or-equal magnitude, and the generator losses within the spread the eta draws allow
        assert all(map(lambda v: v == v and abs(v) < 1e4, g0 + g1 + d0 + d1))
        for a, b in zip(g0, g1):
            assert abs(a - b) < 0.2 * max(1.0, abs(a)), (g0, g1)
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old
        dist.destroy_process_group()


def _moe_two_rank_worker(rank, world, port, out_dir, overlap=True):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "attention-models_amd"))
    from amk import ops
    from amk.models import ViTMoE
    from amk.train import ClassifierTrainStep

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(30 + rank)           # different init per rank: the broadcast must fix it
        model = ViTMoE(**SMALL_MOE).to(dev)
        ts = ClassifierTrainStep(model, lr=1e-3, warmup_steps=1, total_steps=10, bucket_bytes=256 << 10,
                                 accum_steps=1 if world > 1 else 2, overlap=overlap)
        g = torch.Generator().manual_seed(6)
        imgs = torch.randn(3, 8, 3, 64, 64, generator=g).to(dev)
        labels = torch.randint(0, 10, (3, 8), generator=g).to(dev)
        early = []
        finish = ts.red.finish

        def spying_finish(*a, **k):
            if ts.red.sync_step:
                early.append(list(ts.red.launch_order))
            return finish(*a, **k)

        ts.red.finish = spying_finish
        for s in range(3):
            if world > 1:
                ts.step(imgs[s, 4 * rank:4 * rank + 4], labels[s, 4 * rank:4 * rank + 4])
            else:
                ts.step(imgs[s, :4], labels[s, :4])
                ts.step(imgs[s, 4:], labels[s, 4:])
                ts._lr_arg = s
        torch.cuda.synchronize()
        torch.save(dict(params={n: p.detach().cpu().clone() for n, p in model.named_parameters()}, early=early,
                        unused=[n for n, p in model.named_parameters() if any(p is q for q in ts.red.static_unused_parameters())]),
                   os.path.join(out_dir, f"moe_w{world}_r{rank}.pt"))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_classifier_step_on_reduced_vit_moe(device, tmp_path, overlap):
    """overlap=False: the all-reduces on the compute stream at bucket completion (the form bench.py captures for models with
    little gradient traffic) -- same buckets, same order, same results."""
    import torch.multiprocessing as mp

    mp.spawn(_moe_two_rank_worker, args=(2, _free_port(), str(tmp_path), overlap), nprocs=2, join=True)
    mp.spawn(_moe_two_rank_worker, args=(1, 0, str(tmp_path), overlap), nprocs=1, join=True)   # (fresh process: same start state)
    r0 = torch.load(tmp_path / "moe_w2_r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "moe_w2_r1.pt", weights_only=True)
    one = torch.load(tmp_path / "moe_w1_r0.pt", weights_only=True)
    assert r0["unused"] and all("W_d" in n for n in r0["unused"]) and r0["unused"] == r1["unused"]
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n
        assert_close(r0["params"][n], one["params"][n], 2e-4, n)
    # step 0 records the static-unused set; afterwards the first bucket leaves inside backward
    assert len(r0["early"][1]) >= 1 and r0["early"][1][0] == 0 and len(r0["early"][2]) >= 1, r0["early"]


@pytest.mark.timeout(900)
def test_vit_moe_config3_full_size_through_reducer_and_flat_adamw(device):
    """configs[3] (240.6 M parameters) through GradReducer + FlatAdam(AdamW): 962 MB of gradients in 32-MiB buckets,
    3.9 GB of flat state; two steps, finite, W_d untouched, every other parameter updated by the second step."""
    from amk.models import ViTMoE
    from amk.train import ClassifierTrainStep

    torch.manual_seed(0)
    model = ViTMoE(dim=1024, image_size=256, patch_size=32, n_heads=8, d_head=64, depth=6, n_experts=32, sel_experts=2,
                   dropout=0.0, num_classes=1000).to(device)
    ts = ClassifierTrainStep(model, lr=1e-4, warmup_steps=1, total_steps=100)
    assert ts.red.grads_nbytes() > 960e6 and len(ts.red.buckets) >= 12
    wd0 = {n: p.detach().clone() for n, p in model.named_parameters() if "W_d" in n}
    g = torch.Generator().manual_seed(2)
    imgs = torch.randn(8, 3, 256, 256, generator=g).to(device)
    labels = torch.randint(0, 1000, (8,), generator=g).to(device)
    first = [p.detach().clone() for p in list(model.parameters())[:4]]
    losses = [float(ts.step(imgs, labels)) for _ in range(3)]
    assert all(l == l and l < 20 for l in losses), losses
    for n, p in model.named_parameters():
        if "W_d" in n:
            assert torch.equal(p, wd0[n]), n
    assert any(not torch.equal(a, b) for a, b in zip(first, list(model.parameters())[:4]))
    assert {id(p) for p in ts.red.static_unused_parameters()} == {id(p) for n, p in model.named_parameters() if "W_d" in n}


def test_masked_token_step_frozen_vq(device):
    """trainers/muse.py:48-97 on a small MUSE: the frozen vq is in no bucket and does not move; biases / embeddings get
    no weight decay; three steps run and the loss is finite."""
    from amk.models import MUSE, ViTVQGAN
    from amk.train import MaskedTokenTrainStep

    torch.manual_seed(0)
    vq = ViTVQGAN(dict(dim=64, img_size=32, patch_size=4, n_heads=2, d_head=64, depth=1, mlp_dim=96, dropout=0.0),
                  dict(codebook_size=128, codebook_dim=32)).to(device)
    model = MUSE(dim=128, vq=vq, text_dim=48, n_heads=2, d_head=64, depth=2, mult=2).to(device)
    vq0 = {n: p.detach().clone() for n, p in vq.named_parameters()}
    ts = MaskedTokenTrainStep(model, lr=1e-3, weight_decay=0.05, warmup_steps=1, bucket_bytes=128 << 10)
    managed = {id(p) for p in ts.red.params}
    assert not any(id(p) in managed for p in vq.parameters())
    nd = {id(p) for p, w in zip(ts.optim.params, ts.optim.wd) if w == 0.0}
    assert nd == {id(p) for n, p in model.named_parameters() if p.requires_grad and any(f in n for f in ts.NO_DECAY)}
    text = torch.randn(4, 7, 48, device=device)
    imgs = torch.rand(4, 3, 32, 32, device=device)
    for _ in range(3):
        loss = ts.step(text, imgs)
    assert torch.isfinite(loss)
    for n, p in vq.named_parameters():
        assert torch.equal(p, vq0[n]), n


def test_accumulated_step_captured_as_one_graph(device):
    """The shipped Muse schedule in miniature (cfg/muse.yaml:51,80: tiny batches, many accumulation steps): accum_steps
    micro-batches and the optimizer step as ONE graph replay equal the same iterations run one by one -- parameters and
    schedule position -- with FlatAdam as plain Adam-with-L2 (cfg/maskgit.yaml:59) and no clipping (max_grad_norm null)."""
    from amk import ops
    from amk.models import ViTMoE
    from amk.train import ClassifierTrainStep

    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        torch.manual_seed(0)
        base = ViTMoE(**SMALL_MOE).to(device)
        g = torch.Generator().manual_seed(8)
        mbs = [(torch.randn(2, 3, 64, 64, generator=g).to(device), torch.randint(0, 10, (2,), generator=g).to(device)) for _ in range(4)]
        runs = []
        for graphed in (False, True):
            model = copy.deepcopy(base)
            ts = ClassifierTrainStep(model, lr=1e-3, weight_decay=0.01, decoupled=False, warmup_steps=2, total_steps=40,
                                     max_grad_norm=None, accum_steps=4, bucket_bytes=256 << 10, capturable=True)
            if graphed:
                ts.capture_accumulated(mbs, warmup=1)
                for _ in range(2):
                    ts.step_accumulated(mbs)
            else:
                for _ in range(3):
                    for mb in mbs:
                        ts.step(*mb)
            torch.cuda.synchronize()
            runs.append(([p.detach().clone() for p in model.parameters()], ts.global_step, ts.last_lr))
        (p0, s0, lr0), (p1, s1, lr1) = runs
        assert s0 == s1 == 12 and lr0 == lr1
        for a, b in zip(p0, p1):
            assert torch.equal(a, b)
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old


def test_classifier_step_learns(device):
    """Forty AdamW steps of ClassifierTrainStep on eight fixed images: the reduced ViTMoE memorises them (the cross-entropy
    falls from ~ln 10 to well under 1) -- router, expert GEMMs, SwitchHead, schedule and optimizer working together."""
    from amk.models import ViTMoE
    from amk.train import ClassifierTrainStep

    torch.manual_seed(0)
    model = ViTMoE(**SMALL_MOE).to(device)
    ts = ClassifierTrainStep(model, lr=2e-3, warmup_steps=3, total_steps=200, max_grad_norm=1.0, bucket_bytes=256 << 10)
    imgs, labels = _moe_batch(device)
    losses = [float(ts.step(imgs, labels)) for _ in range(40)]
    assert all(l == l for l in losses)
    assert losses[0] > 1.8 and losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
