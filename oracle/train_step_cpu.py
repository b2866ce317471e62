"""CPU baseline for bench.py: the same ViT-VQGAN training step (trainers/vitgqgan.py:139-206 of
the reference, per_loss_weight 0) evaluated with the CPU oracle (oracle/ref_cpu.py) for the
generator and plain PyTorch CPU ops for the PatchGAN discriminator.  Test infrastructure: it is
the thing TIMED as ``cpu_baseline`` ("port"), never a product path."""
import os
import time

import torch
import torch.nn.functional as F

from . import ref_cpu


def _step(w, names, cfg, discr, g_opt, d_opt, img, adv_w=0.1, laplace_w=1.0, max_norm=1.0, eta=None):
    """One step of trainers/vitgqgan.py:139-189 (per_loss_weight 0).  `eta`: the gradient penalty's
    interpolation weights (reference :116 draws them uniformly; the parity test passes the same ones to
    both sides).  Returns the logged scalars (reference :201-204)."""
    params = [w[n] for n in names]
    # discriminator phase
    for p in params:
        p.requires_grad_(False)
    for p in discr.parameters():
        p.requires_grad_(True)
    rec, _, _ = ref_cpu.vitvqgan_forward(img, w, cfg)
    fake, real = discr(rec), discr(img)
    if eta is None:
        eta = torch.rand(img.shape[0], 1, 1, 1)
    mixed = (eta * img + (1 - eta) * rec).detach().requires_grad_(True)
    pred = discr(mixed)
    (g,) = torch.autograd.grad(pred, mixed, torch.ones_like(pred), create_graph=True, retain_graph=True)
    gp = ((g.norm(2, dim=1) - 1) ** 2).mean() * 10.0
    d_loss = 0.5 * (F.relu(1 - real).mean() + F.relu(1 + fake).mean()) + gp
    d_loss.backward()
    torch.nn.utils.clip_grad_norm_(discr.parameters(), max_norm)
    d_opt.step()
    d_opt.zero_grad()
    # generator phase
    for p in params:
        p.requires_grad_(True)
    for p in discr.parameters():
        p.requires_grad_(False)
    rec, cb_loss, _ = ref_cpu.vitvqgan_forward(img, w, cfg)
    g_loss, l1, l2 = F.softplus(-discr(rec)).mean(), F.l1_loss(rec, img), F.mse_loss(rec, img)
    loss = cb_loss + adv_w * g_loss + laplace_w * l1 + l2
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, max_norm)
    g_opt.step()
    g_opt.zero_grad()
    return dict(d_loss=float(d_loss.detach()), g_loss=float(g_loss.detach()), l1=float(l1.detach()),
                l2=float(l2.detach()), codebook_loss=float(cb_loss.detach()), loss=float(loss.detach()))


def host_cores():
    """Cores this process may really use: cgroup quota, then affinity, then cpu_count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def time_train_step(state_dict, cfg, discr, batch=2, steps=8, warmup=1, threads=None, budget_s=15.0):
    """Returns (images_per_sec, threads_used, description of the sample).  The sample is
    bounded to about `budget_s` seconds of CPU work: the warm-up step sets how many of `steps` are timed."""
    threads = threads or min(host_cores(), 32)
    torch.set_num_threads(threads)
    w = {k: v.detach().clone().float().cpu() for k, v in state_dict.items()}
    names = sorted(w)
    for n in names:
        w[n].requires_grad_(True)
    discr = discr.cpu().float()
    g_opt = torch.optim.Adam([w[n] for n in names], lr=1e-4)
    d_opt = torch.optim.Adam(discr.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(batch, 3, cfg["img_size"], cfg["img_size"], generator=g)
    t0 = time.perf_counter()
    for _ in range(warmup):
        _step(w, names, cfg, discr, g_opt, d_opt, img)
    if warmup:
        per = (time.perf_counter() - t0) / warmup
        steps = max(1, min(steps, int(budget_s / max(per, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        _step(w, names, cfg, discr, g_opt, d_opt, img)
    dt = time.perf_counter() - t0
    sample = f"{steps} train steps of batch {batch} at {cfg['img_size']}px after {warmup} warm-up, fp32, torch CPU"
    return batch * steps / dt, threads, sample
