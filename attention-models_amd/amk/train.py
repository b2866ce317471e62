"""The ViT-VQGAN training step of the reference (trainers/vitgqgan.py:139-206), one process per
GPU, gradients reduced by amk.dp.GradReducer instead of accelerate's DDP wrap.

Per step and image: 2 generator forwards + 1 generator backward (24 attention-core forwards,
12 backwards, 2 VQ lookups, 1 VQ backward), the PatchGAN discriminator three times plus its
gradient penalty (double backward), two Adam updates.  The LPIPS term of the reference needs
downloaded VGG weights (unavailable offline): ``per_loss_weight`` is therefore 0 here -- the
config key exists in the reference (cfg/vitvqgan.yaml:67) -- everything else follows the
recipe, including the quirks (gradient-penalty norm over the channel axis only).
"""
import math

import torch
import torch.nn.functional as F

from .dp import GradReducer
from .models.discriminator import input_grad_only


def set_requires_grad(module, flag):
    for p in module.parameters():
        p.requires_grad_(flag)


def hinge_d_loss(fake, real):
    return 0.5 * (F.relu(1.0 - real).mean() + F.relu(1.0 + fake).mean())


def g_nonsaturating_loss(fake):
    return F.softplus(-fake).mean()


def cosine_warmup_lr(step, base_lr, t_initial, warmup_t, warmup_lr_init=1e-6, lr_min=5e-5):
    """timm CosineLRScheduler(t_initial, warmup_t, warmup_lr_init, lr_min) as the reference
    configures it (trainers/vitgqgan.py:75-76), single cycle."""
    if step < warmup_t:
        return warmup_lr_init + step * (base_lr - warmup_lr_init) / warmup_t
    t = min(step, t_initial)
    return lr_min + 0.5 * (base_lr - lr_min) * (1.0 + math.cos(math.pi * t / t_initial))


class VQGANTrainStep:
    def __init__(self, model, discr, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0,
                 adv_loss_weight=0.1, logit_laplace_weight=1.0, max_grad_norm=1.0,
                 warmup_steps=50000, decay_steps=100000, gp_lambda=10.0, bucket_bytes=32 << 20,
                 share_forward=False, capturable=False):
        self.model, self.discr = model, discr
        self.adv_w, self.laplace_w = adv_loss_weight, logit_laplace_weight
        self.max_grad_norm, self.gp_lambda = max_grad_norm, gp_lambda
        self.base_lr, self.warmup_steps, self.decay_steps = lr, warmup_steps, decay_steps
        fused = next(model.parameters()).is_cuda
        # capturable=True: the learning rate lives in a device tensor and the optimizer never reads
        # device state on the host, so step_body() can be captured into a HIP graph (amk/graphs.py)
        okw = dict(betas=betas, weight_decay=weight_decay, fused=fused)
        if capturable:
            dev = next(model.parameters()).device
            okw.update(capturable=True)
            self.g_optim = torch.optim.Adam(model.parameters(), lr=torch.tensor(float(lr), device=dev), **okw)
            self.d_optim = torch.optim.Adam(discr.parameters(), lr=torch.tensor(float(lr), device=dev), **okw)
        else:
            self.g_optim = torch.optim.Adam(model.parameters(), lr=lr, **okw)
            self.d_optim = torch.optim.Adam(discr.parameters(), lr=lr, **okw)
        self.g_red = GradReducer(model.parameters(), bucket_bytes)
        self.d_red = GradReducer(discr.parameters(), bucket_bytes)
        self.g_red.broadcast_parameters()
        self.d_red.broadcast_parameters()
        self.global_step = 0
        # The reference runs the generator forward twice per step on the same images with unchanged
        # generator weights (trainers/vitgqgan.py:149 and :169; only the discriminator steps in
        # between).  share_forward=True runs it once, with the autograd graph, and hands the
        # discriminator phase a detached reconstruction: the same numbers while dropout is 0 (every
        # shipped config), one generator forward less.  Off by default: the step then mirrors the
        # reference call for call.
        self.share_forward = bool(share_forward)

    def gradient_penalty(self, real, fake):
        """trainers/vitgqgan.py:115-131."""
        eta = torch.rand(real.shape[0], 1, 1, 1, device=real.device)
        mixed = (eta * real + (1.0 - eta) * fake).detach().requires_grad_(True)
        pred = self.discr(mixed)
        with input_grad_only():   # only d pred / d mixed is asked for: skip the weight gradients
            (grad,) = torch.autograd.grad(pred, mixed, grad_outputs=torch.ones_like(pred),
                                          create_graph=True, retain_graph=True)
        return ((grad.norm(2, dim=1) - 1.0) ** 2).mean() * self.gp_lambda

    def _set_lr(self):
        lr = cosine_warmup_lr(self.global_step, self.base_lr, self.decay_steps, self.warmup_steps)
        for opt in (self.g_optim, self.d_optim):
            for g in opt.param_groups:
                if torch.is_tensor(g["lr"]):
                    g["lr"].fill_(lr)
                else:
                    g["lr"] = lr

    def step(self, img, sync=True):
        self._set_lr()
        logs = self.step_body(img, sync)
        self.global_step += 1
        return logs

    def step_body(self, img, sync=True):
        """Everything of a step that runs on the device (no host-side schedule): what a HIP graph captures."""
        model, discr = self.model, self.discr
        # ---- discriminator phase (reference :146-163)
        shared = None
        if self.share_forward:
            set_requires_grad(model, True)
            shared = model(img)
            rec = shared[0].detach()
        else:
            set_requires_grad(model, False)
        set_requires_grad(discr, True)
        self.d_red.begin(sync)
        if shared is None:
            rec, _ = model(img)
        d_loss = hinge_d_loss(discr(rec), discr(img)) + self.gradient_penalty(img, rec)
        d_loss.backward()
        self.d_red.finish()
        if sync:
            if self.max_grad_norm:
                torch.nn.utils.clip_grad_norm_(discr.parameters(), self.max_grad_norm)
            self.d_optim.step()
            self.d_red.zero_grad()
        # ---- generator phase (reference :167-189)
        set_requires_grad(model, True)
        set_requires_grad(discr, False)
        self.g_red.begin(sync)
        rec, codebook_loss = shared if shared is not None else model(img)
        l1 = F.l1_loss(rec, img)
        l2 = F.mse_loss(rec, img)
        g_loss = g_nonsaturating_loss(discr(rec))
        loss = codebook_loss + self.adv_w * g_loss + self.laplace_w * l1 + l2
        loss.backward()
        self.g_red.finish()
        if sync:
            if self.max_grad_norm:
                torch.nn.utils.clip_grad_norm_(model.parameters(), self.max_grad_norm)
            self.g_optim.step()
            self.g_red.zero_grad()
        return dict(d_loss=d_loss.detach(), g_loss=g_loss.detach(), l1=l1.detach(), l2=l2.detach(),
                    codebook_loss=codebook_loss.detach(), loss=loss.detach())

    # ---- checkpoints in the reference's format (trainers/utils/base_trainer.py:92-115)
    def save_ckpt(self, path, config=None):
        """{'step', 'state_dict', 'config'} with the generator's state_dict under the reference's key
        names, so either side can load the other's file (models/model_factory.py:14-17)."""
        ckpt = {"step": self.global_step,
                "state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                "config": config}
        torch.save(ckpt, path)

    def resume_from_checkpoint(self, path):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        self.global_step = int(ckpt["step"])
        self.model.load_state_dict(ckpt["state_dict"])
        self.g_red.broadcast_parameters()
        return ckpt.get("config")
