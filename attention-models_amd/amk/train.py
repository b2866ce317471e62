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
from .optim import FlatAdam


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


def cosine_with_warmup_factor(step, num_warmup_steps, num_training_steps, num_cycles=0.5):
    """The lr lambda of transformers.get_cosine_schedule_with_warmup (trainers/vit.py:33,
    trainers/utils/scheduler.py:12-13); tests/test_host_logic.py checks it against the library."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def constant_with_warmup_factor(step, num_warmup_steps):
    """transformers.get_constant_schedule_with_warmup (trainers/utils/scheduler.py:10-11)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1.0, num_warmup_steps))
    return 1.0


class SupervisedTrainStep:
    """The single-model train loops of the reference -- trainers/vit.py:66-76 (ViT / ViTMoE classifier),
    trainers/muse.py:88-97, trainers/maskgit.py (masked-token decoder over a FROZEN vq) -- one process per GPU:

        with accelerator.accumulate(model):            sync = every accum_steps-th call; else no communication
            with accelerator.autocast(): out = model(...)
            loss = ...
            accelerator.backward(loss)                 loss / accum_steps; gradients all-reduced by amk.dp.GradReducer
            if sync and max_grad_norm: clip_grad_norm_       (bucketed RCCL all-reduce on a side stream under backward)
            optim.step(); scheduler.step(global_step); optim.zero_grad()       (all three only on sync iterations)

    Subclasses give ``_loss(*batch)``.  Parameters with requires_grad False (the frozen vq of MUSE) are in no bucket;
    parameters that never receive a gradient (SwitchHeadAttention.W_d, models/switchhead_attention.py:80-87) travel as
    zeros, are recorded as static-unused by the first synchronised step and are skipped by the optimizer, as torch
    skips ``.grad is None``.  The optimizer is torch.optim.AdamW's arithmetic on amk.optim.FlatAdam (clip + update +
    zeroing in two passes over the flat buckets; torch.optim on CPU tensors); `no_decay`: name fragments whose parameters
    get weight decay 0 (trainers/muse.py:48-58).

    Schedule: the reference calls ``scheduler.step(self.global_step)`` after the optimizer, on sync iterations only
    (accelerate skips it otherwise), and LambdaLR starts the optimizer at factor(0): the optimizer step of iteration g
    runs with base_lr * factor(g'), g' the PREVIOUS sync iteration's index (0 for the first).

    capture(): the device side of one optimizer step -- forward, backward, the buckets' all-reduces on the side stream,
    clip, AdamW -- as ONE HIP-graph replay.  With more than one rank the RCCL collectives are captured with it (torch
    captures ProcessGroupNCCL collectives; the side stream joins the capture through the reducer's ready events), so a
    data-parallel step is not host-paced; over gloo (CPU tests, the shared-GPU rehearsal) capture() refuses."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, no_decay=(), decoupled=True,
                 schedule="cosine_with_warmup", warmup_steps=0, total_steps=1, max_grad_norm=1.0, accum_steps=1,
                 autocast=None, capturable=False, bucket_bytes=32 << 20, fused_optimizer=None, direct_grads=None,
                 communicate_when_alone=False, overlap=True):
        self.model = model
        self.base_lr, self.schedule = float(lr), schedule
        self.warmup_steps, self.total_steps = int(warmup_steps), int(total_steps)
        self.max_grad_norm, self.accum_steps, self.autocast = max_grad_norm, int(accum_steps), autocast
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        on_gpu = named[0][1].is_cuda
        # communicate_when_alone: a world of one rank issues its (identity) RCCL all-reduces anyway -- how the one-GPU test
        # box exercises the collective path, eager and captured
        self.red = GradReducer([p for _, p in named], bucket_bytes, communicate_when_alone=communicate_when_alone,
                               overlap=overlap, direct_grads=on_gpu if direct_grads is None else bool(direct_grads))
        skip = [p for n, p in named if any(frag in n for frag in no_decay)]
        self.fused_optimizer = on_gpu if fused_optimizer is None else bool(fused_optimizer)
        if self.fused_optimizer:
            self.optim = FlatAdam(self.red, lr=lr, betas=betas, weight_decay=weight_decay, decoupled=decoupled,
                                  capturable=capturable, bf16_shadow=autocast == torch.bfloat16, no_decay=skip)
        else:
            ids = {id(p) for p in skip}
            groups = [dict(params=[p for _, p in named if id(p) not in ids], weight_decay=weight_decay),
                      dict(params=skip, weight_decay=0.0)]
            cls = torch.optim.AdamW if decoupled else torch.optim.Adam
            kw = dict(capturable=True) if capturable else {}
            self.optim = cls([g for g in groups if g["params"]], betas=betas,
                             lr=torch.tensor(float(lr), device=named[0][1].device) if capturable else lr, **kw)
        self.red.broadcast_parameters()
        if self.fused_optimizer:
            self.optim.refresh_shadow()
        self.global_step = 0
        self._micro = 0          # accelerator.step: calls of accumulate() so far
        self._lr_arg = 0         # the index the scheduler was last stepped with
        self._graph = None
        self._accum_graph = None
        self.last_lr = None

    # ---- what a subclass provides
    def _loss(self, *batch):
        raise NotImplementedError

    # ---- schedule
    def lr_factor(self, step):
        if self.schedule == "cosine_with_warmup":
            return cosine_with_warmup_factor(step, self.warmup_steps, self.total_steps)
        if self.schedule == "constant_with_warmup":
            return constant_with_warmup_factor(step, self.warmup_steps)
        if self.schedule in (None, "constant"):
            return 1.0
        raise ValueError(f"unknown schedule {self.schedule!r}")

    def _set_lr(self):
        lr = self.last_lr = self.base_lr * self.lr_factor(self._lr_arg)
        if isinstance(self.optim, FlatAdam):
            self.optim.lr = lr
            return
        for g in self.optim.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    def _amp(self):
        import contextlib

        if self.autocast is None:
            return contextlib.nullcontext()
        dev = "cuda" if self.red.on_gpu else "cpu"
        return torch.autocast(dev, dtype=self.autocast)

    # ---- the step
    def step_body(self, sync, *batch):
        """Everything of one iteration that runs on the device: what capture() records (sync=True)."""
        self.red.begin(sync)
        loss = self._loss(*batch)
        (loss if self.accum_steps == 1 else loss / self.accum_steps).backward()
        self.red.finish()
        if sync:
            if isinstance(self.optim, FlatAdam):
                self.optim.step(max_norm=self.max_grad_norm)
            else:
                if self.max_grad_norm:
                    torch.nn.utils.clip_grad_norm_(self.red.params, self.max_grad_norm)
                self.optim.step()
                self.red.zero_grad()
        return loss.detach()

    def step(self, *batch):
        """One iteration of the reference loop on one (micro-)batch.  Returns the loss (device scalar, undivided)."""
        self._micro += 1
        sync = self._micro % self.accum_steps == 0
        if sync:
            self._set_lr()
        g = self._graph
        if g is not None and sync and self.accum_steps == 1 and len(batch) == len(g.static_inputs) and all(
                a.shape == b.shape for a, b in zip(batch, g.static_inputs)):
            loss = g.replay(*batch)
        else:
            loss = self.step_body(sync, *batch)
        if sync:
            self._lr_arg = self.global_step      # scheduler.step(self.global_step)
        self.global_step += 1
        return loss

    def _tick(self):
        self._micro += 1
        self._set_lr()
        self._lr_arg = self.global_step
        self.global_step += 1

    def capture(self, *batch, warmup=2):
        """Capture step_body(True, *batch) into a HIP graph (accum_steps 1); step() then replays it for batches of these
        shapes.  `warmup` eager steps run first: real optimizer steps, with their schedule ticks."""
        from .graphs import GraphedStep

        if self.accum_steps != 1:
            raise RuntimeError("capture(): gradient accumulation steps run eagerly")
        if not self.red.alone and not self.red.avg_in_collective:
            raise RuntimeError("capture(): only RCCL (backend 'nccl') collectives can be captured into a HIP graph")
        self._graph = None
        self._graph = GraphedStep(lambda *b: self.step_body(True, *b), list(batch), warmup=warmup, before_each=self._tick)

    def release_graph(self):
        self._graph = None
        self._accum_graph = None

    # ---- one optimizer step over accum_steps micro-batches (the shipped Muse / MaskGit schedules: batch 1-8 x 16-32
    # accumulation steps, cfg/muse.yaml:51,80, cfg/maskgit.yaml:45,75 -- host-bound launch by launch)
    def _accum_body(self, k):
        n = self.accum_steps

        def body(*xs):
            for i in range(n - 1):
                self.step_body(False, *xs[i * k:(i + 1) * k])
            return self.step_body(True, *xs[(n - 1) * k:])
        return body

    def _tick_accum(self):
        n = self.accum_steps
        self._micro += n
        self._set_lr()
        self._lr_arg = self.global_step + n - 1
        self.global_step += n

    def capture_accumulated(self, micro_batches, warmup=2):
        """Capture a whole optimizer step -- accum_steps micro-batches (each a tuple of tensors), the last one synchronising
        -- into ONE HIP graph; step_accumulated() then replays it.  Same conditions as capture()."""
        from .graphs import GraphedStep

        if len(micro_batches) != self.accum_steps:
            raise ValueError(f"capture_accumulated: {len(micro_batches)} micro-batches for accum_steps={self.accum_steps}")
        if not self.red.alone and not self.red.avg_in_collective:
            raise RuntimeError("capture_accumulated(): only RCCL (backend 'nccl') collectives can be captured into a HIP graph")
        k = len(micro_batches[0])
        self._accum_graph = None
        self._accum_graph = GraphedStep(self._accum_body(k), [t for mb in micro_batches for t in mb], warmup=warmup,
                                        before_each=self._tick_accum)

    def step_accumulated(self, micro_batches):
        """accum_steps iterations of the reference loop at once (call it on optimizer-step boundaries): a replay of
        capture_accumulated()'s graph when the shapes match, else eager micro-steps.  Returns the last micro-batch's loss."""
        n = self.accum_steps
        if len(micro_batches) != n or self._micro % n != 0:
            raise ValueError("step_accumulated: pass accum_steps micro-batches, starting on an optimizer-step boundary")
        flat = [t for mb in micro_batches for t in mb]
        self._micro += n
        self._set_lr()
        g = getattr(self, "_accum_graph", None)
        if g is not None and len(flat) == len(g.static_inputs) and all(a.shape == b.shape for a, b in zip(flat, g.static_inputs)):
            loss = g.replay(*flat)
        else:
            loss = self._accum_body(len(micro_batches[0]))(*flat)
        self._lr_arg = self.global_step + n - 1      # the synchronising iteration's scheduler.step(self.global_step)
        self.global_step += n
        return loss

    def save_ckpt(self, path, config=None):
        """trainers/utils/base_trainer.py:92-107: {'step', 'state_dict', 'config'}, main process only."""
        import torch.distributed as dist

        if dist.is_initialized() and dist.get_rank() != 0:
            return
        torch.save({"step": self.global_step, "config": config,
                    "state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}, path)


    def resume_from_checkpoint(self, path):
        """trainers/utils/base_trainer.py:110-115: step + model weights (no optimizer / scheduler state, as the reference)."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        self.global_step = int(ckpt["step"])
        self.model.load_state_dict(ckpt["state_dict"])
        self.red.broadcast_parameters()
        if self.fused_optimizer:
            self.optim.refresh_shadow()
        return ckpt.get("config")


class ClassifierTrainStep(SupervisedTrainStep):
    """trainers/vit.py: AdamW(lr, betas) -- torch's default weight decay 0.01 (:29) --, CrossEntropyLoss (:31) on the
    logits computed under autocast (:67-71: the loss itself sits outside the autocast block), cosine schedule with warm-up
    (:33).  BASELINE.json configs[1] (ViT) and configs[3] (ViTMoE)."""

    def _loss(self, imgs, target):
        with self._amp():
            out = self.model(imgs)
        return F.cross_entropy(out, target)


class MaskedTokenTrainStep(SupervisedTrainStep):
    """trainers/muse.py:45-97 / trainers/maskgit.py: ``loss = model(text, img)`` under autocast, the model's vq frozen
    (models/model_factory.py freeze_model: not in any bucket), AdamW with weight decay 0 on biases / LayerNorm gains /
    embeddings (:48-58).  `text` here is the text tower's output (amk.models.muse.MUSE takes CLIP's hidden states: the tower
    itself is a download); for a MaskGit-style model without text pass a 1-tuple batch."""

    NO_DECAY = ("bias", "layer_norm.weight", "embeddings.weight")

    def __init__(self, model, weight_decay=0.0, no_decay=NO_DECAY, schedule="constant_with_warmup", **kw):
        super().__init__(model, weight_decay=weight_decay, no_decay=no_decay, schedule=schedule, **kw)

    def _loss(self, *batch):
        with self._amp():
            return self.model(*batch)


class VQGANTrainStep:
    def __init__(self, model, discr, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0,
                 adv_loss_weight=0.1, logit_laplace_weight=1.0, max_grad_norm=1.0,
                 warmup_steps=50000, decay_steps=100000, gp_lambda=10.0, bucket_bytes=32 << 20,
                 share_forward=False, capturable=False, fused_optimizer=None, autocast=None, communicate_when_alone=False,
                 overlap=True):
        self.model, self.discr = model, discr
        # autocast: None (f32, the parity mode) or a dtype (torch.bfloat16): the forward passes and losses of both phases
        # run inside torch.autocast as the reference's do (trainers/vitgqgan.py:149,168 `accelerator.autocast()`);
        # parameters, gradients and optimizer state stay f32
        self._autocast = autocast
        self.adv_w, self.laplace_w = adv_loss_weight, logit_laplace_weight
        self.max_grad_norm, self.gp_lambda = max_grad_norm, gp_lambda
        self.base_lr, self.warmup_steps, self.decay_steps = lr, warmup_steps, decay_steps
        fused = next(model.parameters()).is_cuda
        # direct_grads: the Linear layers' weight-gradient kernels write straight into the generator's gradient buckets (no
        # weight is shared between two such layers in ViTVQGAN); the discriminator's convolutions go through autograd
        # overlap=False: all-reduces on the compute stream (see amk.dp.GradReducer: what a captured step with little
        # gradient traffic wants)
        self.g_red = GradReducer(model.parameters(), bucket_bytes, direct_grads=fused, communicate_when_alone=communicate_when_alone,
                                 overlap=overlap)
        self.d_red = GradReducer(discr.parameters(), bucket_bytes, communicate_when_alone=communicate_when_alone, overlap=overlap)
        # fused_optimizer (default on the GPU): amk.optim.FlatAdam -- clip + Adam + zeroing in two passes over the
        # reducer's flat buckets (csrc/optim.hip) instead of clip_grad_norm_ + Adam.step + zero_grad.
        # capturable=True: step counts and learning rate live on the device (FlatAdam(capturable=True), or torch's
        # Adam with a tensor lr), so that step_body() can be captured into a HIP graph: see capture().
        self.fused_optimizer = fused if fused_optimizer is None else bool(fused_optimizer)
        okw = dict(betas=betas, weight_decay=weight_decay, fused=fused)
        self._graph = None
        self._accum_graph = None
        if self.fused_optimizer:
            # under bf16 autocast the generator's update kernel also refreshes the bf16 copies its GEMMs read
            self.g_optim = FlatAdam(self.g_red, lr=lr, betas=betas, weight_decay=weight_decay, capturable=capturable,
                                    bf16_shadow=autocast == torch.bfloat16)
            self.d_optim = FlatAdam(self.d_red, lr=lr, betas=betas, weight_decay=weight_decay, capturable=capturable)
        elif capturable:
            dev = next(model.parameters()).device
            okw.update(capturable=True)
            self.g_optim = torch.optim.Adam(model.parameters(), lr=torch.tensor(float(lr), device=dev), **okw)
            self.d_optim = torch.optim.Adam(discr.parameters(), lr=torch.tensor(float(lr), device=dev), **okw)
        else:
            self.g_optim = torch.optim.Adam(model.parameters(), lr=lr, **okw)
            self.d_optim = torch.optim.Adam(discr.parameters(), lr=lr, **okw)
        self.g_red.broadcast_parameters()
        self.d_red.broadcast_parameters()
        if self.fused_optimizer:
            # the bf16 copies were derived from this rank's initial weights; the broadcast has just replaced those
            self.g_optim.refresh_shadow()
        self.global_step = 0
        # The reference runs the generator forward twice per step on the same images with unchanged
        # generator weights (trainers/vitgqgan.py:149 and :169; only the discriminator steps in
        # between).  share_forward=True runs it once, with the autograd graph, and hands the
        # discriminator phase a detached reconstruction: the same numbers while dropout is 0 (every
        # shipped config), one generator forward less.  Off by default: the step then mirrors the
        # reference call for call.
        self.share_forward = bool(share_forward)

    @property
    def autocast(self):
        return self._autocast

    @autocast.setter
    def autocast(self, dtype):
        """Switch precision modes between steps (not inside a captured graph: release_graph() first)."""
        self._autocast = dtype
        if self.fused_optimizer:
            if dtype == torch.bfloat16:
                self.g_optim.enable_shadow()
            else:
                self.g_optim.disable_shadow()

    def _amp(self):
        import contextlib

        if self.autocast is None:
            return contextlib.nullcontext()
        return torch.autocast("cuda", dtype=self.autocast)

    def gradient_penalty(self, real, fake, eta=None):
        """trainers/vitgqgan.py:115-131.  eta: the interpolation weights (drawn here unless given)."""
        if eta is None:
            eta = torch.rand(real.shape[0], 1, 1, 1, device=real.device)
        mixed = (eta * real + (1.0 - eta) * fake).detach().requires_grad_(True)
        pred = self.discr(mixed)
        with input_grad_only():   # only d pred / d mixed is asked for: skip the weight gradients
            (grad,) = torch.autograd.grad(pred, mixed, grad_outputs=torch.ones_like(pred),
                                          create_graph=True, retain_graph=True)
        return ((grad.norm(2, dim=1) - 1.0) ** 2).mean() * self.gp_lambda

    def _set_lr(self):
        # The reference steps its schedulers AFTER the optimizer, with the step index before the increment
        # (trainers/vitgqgan.py:161-162,187-188,206), and timm's scheduler starts the optimizer at
        # warmup_lr_init: step n therefore runs with schedule(n - 1), steps 0 and 1 both with schedule(0).
        lr = cosine_warmup_lr(max(self.global_step - 1, 0), self.base_lr, self.decay_steps, self.warmup_steps)
        for opt in (self.g_optim, self.d_optim):
            if isinstance(opt, FlatAdam):
                opt.lr = lr
                continue
            for g in opt.param_groups:
                if torch.is_tensor(g["lr"]):
                    g["lr"].fill_(lr)
                else:
                    g["lr"] = lr

    def step(self, img, sync=True, accum_steps=1, eta=None):
        """sync=False: a gradient-accumulation micro-step (no communication, no optimizer step).
        accum_steps: the losses are divided by it before backward, as accelerator.backward does with
        gradient_accumulation_steps (the logged values stay undivided)."""
        self._set_lr()
        if self._graph is not None and sync and accum_steps == 1 and eta is None and img.shape == self._graph.static_inputs[0].shape:
            logs = self._graph.replay(img)  # the whole device side of the step as one HIP-graph launch
        else:
            logs = self.step_body(img, sync, accum_steps, eta)
        self.global_step += 1
        return logs

    def capture(self, img, warmup=2):
        """Capture step_body() for batches shaped like `img` into a HIP graph; step() then replays it (the host
        enqueues one launch instead of ~2500).  Needs capturable=True.  With several ranks the reducers' RCCL collectives
        on their side stream are captured WITH the step (the side stream joins the capture through the ready events), so
        the data-parallel step is one replay per rank too; over gloo it refuses.  `warmup` eager steps run first on the
        capture stream -- they are real optimizer steps.  release_graph() returns to eager steps."""
        from .graphs import GraphedStep

        self._check_capturable("capture")
        self._graph = None
        self._graph = GraphedStep(lambda x: self.step_body(x), [img], warmup=warmup, before_each=self._tick)

    def _check_capturable(self, what):
        for red in (self.g_red, self.d_red):
            if not red.alone and not red.avg_in_collective:
                raise RuntimeError(f"VQGANTrainStep.{what}: only RCCL (backend 'nccl') collectives can be captured into a "
                                   "HIP graph; over gloo the data-parallel step runs eagerly")

    def release_graph(self):
        self._graph = None
        self._accum_graph = None

    def capture_accumulated(self, micro_batches, warmup=2):
        """Capture ONE optimizer step over several micro-batches (gradient accumulation: every micro-batch but the last
        with sync=False, losses divided by their number, as accelerator.accumulate runs the reference's shipped
        `gradient_accumulation_steps`) into a HIP graph; step_accumulated() then replays it.  Same conditions as capture()."""
        from .graphs import GraphedStep

        self._check_capturable("capture_accumulated")
        n = len(micro_batches)

        def body(*xs):
            for x in xs[:-1]:
                self.step_body(x, False, n, None)
            return self.step_body(xs[-1], True, n, None)

        self._accum_graph = None
        self._accum_graph = GraphedStep(body, list(micro_batches), warmup=warmup, before_each=lambda: self._tick(n))

    def _tick(self, n=1):
        """A warm-up step of capture() is a real optimizer step: it gets its learning rate and advances global_step."""
        self.global_step += n - 1
        self._set_lr()
        self.global_step += 1

    def step_accumulated(self, micro_batches):
        """One optimizer step over the micro-batches: a replay of capture_accumulated()'s graph when the shapes match, else
        eager micro-steps.  The reference ticks its schedulers every iteration (trainers/vitgqgan.py:161,187) and the
        optimizers step at the last micro-batch: the learning rate is the one of that iteration."""
        self.global_step += len(micro_batches) - 1
        self._set_lr()
        self.global_step -= len(micro_batches) - 1
        g = getattr(self, "_accum_graph", None)
        if g is not None and len(micro_batches) == len(g.static_inputs) and all(
                a.shape == b.shape for a, b in zip(micro_batches, g.static_inputs)):
            logs = g.replay(*micro_batches)
        else:
            n = len(micro_batches)
            for x in micro_batches[:-1]:
                self.step_body(x, False, n, None)
            logs = self.step_body(micro_batches[-1], True, n, None)
        self.global_step += len(micro_batches)
        return logs

    def _optim_step(self, opt, red, module):
        if isinstance(opt, FlatAdam):
            opt.step(max_norm=self.max_grad_norm)
            return
        if self.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(module.parameters(), self.max_grad_norm)
        opt.step()
        red.zero_grad()

    def d_phase(self, img, sync=True, accum_steps=1, eta=None, rec=None):
        """Discriminator phase (reference :146-163).  rec: a detached reconstruction to reuse (share_forward)."""
        model, discr = self.model, self.discr
        if rec is None:
            set_requires_grad(model, False)
        set_requires_grad(discr, True)
        self.d_red.begin(sync)
        with self._amp():
            if rec is None:
                rec, _ = model(img)
            d_loss = hinge_d_loss(discr(rec), discr(img)) + self.gradient_penalty(img, rec, eta)
        (d_loss if accum_steps == 1 else d_loss / accum_steps).backward()
        self.d_red.finish()
        if sync:
            self._optim_step(self.d_optim, self.d_red, discr)
        return d_loss.detach()

    def g_phase(self, img, sync=True, accum_steps=1, shared=None):
        """Generator phase (reference :167-189).  shared: (rec, codebook_loss) with its graph (share_forward)."""
        model, discr = self.model, self.discr
        set_requires_grad(model, True)
        set_requires_grad(discr, False)
        self.g_red.begin(sync)
        with self._amp():
            rec, codebook_loss = shared if shared is not None else model(img)
            l1 = F.l1_loss(rec, img)
            l2 = F.mse_loss(rec, img)
            g_loss = g_nonsaturating_loss(discr(rec))
            loss = codebook_loss + self.adv_w * g_loss + self.laplace_w * l1 + l2
        (loss if accum_steps == 1 else loss / accum_steps).backward()
        self.g_red.finish()
        if sync:
            self._optim_step(self.g_optim, self.g_red, model)
        return dict(g_loss=g_loss.detach(), l1=l1.detach(), l2=l2.detach(),
                    codebook_loss=codebook_loss.detach(), loss=loss.detach())

    def step_body(self, img, sync=True, accum_steps=1, eta=None):
        """Everything of a step that runs on the device (no host-side schedule): what a HIP graph captures."""
        shared = None
        if self.share_forward:
            set_requires_grad(self.model, True)
            with self._amp():
                shared = self.model(img)
        d_loss = self.d_phase(img, sync, accum_steps, eta, rec=shared[0].detach() if shared is not None else None)
        logs = self.g_phase(img, sync, accum_steps, shared)
        logs["d_loss"] = d_loss
        return logs

    # ---- checkpoints in the reference's format (trainers/utils/base_trainer.py:92-115)
    def save_ckpt(self, path, config=None):
        """{'step', 'state_dict', 'config'} with the generator's state_dict under the reference's key
        names, so either side can load the other's file (models/model_factory.py:14-17)."""
        import torch.distributed as dist

        if dist.is_initialized() and dist.get_rank() != 0:
            return  # one writer (the reference's accelerator.save is main-process only as well)
        ckpt = {"step": self.global_step,
                "state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                "config": config}
        torch.save(ckpt, path)

    def resume_from_checkpoint(self, path):
        import pickle

        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            # a checkpoint written BY THE REFERENCE stores its OmegaConf DictConfig under 'config'
            # (trainers/utils/base_trainer.py:92-107); the safe loader refuses such objects.
            raise RuntimeError(
                f"{path}: not loadable with weights_only=True ({e}). A reference-written checkpoint must have its "
                "'config' entry converted to a plain dict (or dropped) first; only tensors and plain containers "
                "are loaded here.") from e
        self.global_step = int(ckpt["step"])
        self.model.load_state_dict(ckpt["state_dict"])
        self.g_red.broadcast_parameters()
        if self.fused_optimizer:
            self.g_optim.refresh_shadow()   # (the bf16 copies the mixed-precision GEMMs read: written behind the optimizer's back)
        return ckpt.get("config")
