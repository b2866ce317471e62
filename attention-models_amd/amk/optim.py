"""Adam / AdamW over the flat gradient buckets of amk.dp.GradReducer: global-norm clip, moment and
parameter update and the zeroing of the gradients in two HBM passes (csrc/optim.hip), instead of
``clip_grad_norm_`` + ``Adam.step`` + ``zero_grad`` (reference: trainers/vitgqgan.py:67-68,159-163,
185-189; trainers/vit.py:29-31,74-76).

The parameters themselves move into flat buffers laid out like the gradient buckets (``p.data``
becomes a view), so one launch per bucket updates every parameter in it.  Parameters that received
no gradient since the last step are skipped, as torch optimizers skip ``.grad is None``; step counts
are kept per parameter for the bias corrections.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib
from .dp import ALIGN


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class FlatAdam:
    # the decay factor travels in the per-parameter table (needed for no_decay groups and for a device-resident lr);
    # False: the launch-argument form of the kernel (one weight decay for all, lr from the host)
    TABLE_WD = os.environ.get("AMK_OPT_TABLE_WD", "1") == "1"

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False,
                 capturable=False, bf16_shadow=False, no_decay=()):
        """decoupled=True: AdamW (torch.optim.AdamW; trainers/vit.py:29, trainers/utils/optimizer.py:14-15).  no_decay: the
        parameters of the second group of trainers/muse.py:48-58 / trainers/maskgit.py (weight decay 0); every other
        parameter decays with `weight_decay`.  The decay factor travels in the per-parameter table (lr * wd for AdamW), so
        it follows the learning rate wherever that lives -- host or device.

        capturable=True: the per-parameter table {active, lr / bias-correction-1, sqrt(bias-correction-2)} is
        computed on the device from device-resident step counts and learning rate (a few tiny launches) instead of on
        the host, so that ``step`` can be captured into a HIP graph and replayed (amk/graphs.py): nothing in it then
        reads host memory.  The set of parameters that receive gradients must not change between replays.

        bf16_shadow=True: a bf16 copy of every parameter (``p._amk_bf16``, a view of one flat buffer per bucket) is kept
        current by the update kernel itself; the mixed-precision ops (amk.ops.linear / swiglu_ffn under bf16 autocast)
        read it instead of casting the fp32 weight on every call.  Code that writes parameters behind the optimizer's
        back (load_state_dict, manual init) must call ``refresh_shadow()``."""
        if not reducer.on_gpu:
            raise RuntimeError("FlatAdam runs on MI355X (HIP) parameters only; use torch.optim on CPU")
        self.red = reducer
        self.capturable = bool(capturable)
        self._lr = float(lr)
        self.betas, self.eps, self.weight_decay, self.decoupled = betas, eps, weight_decay, decoupled
        self.L = _lib.load()
        self.npart = self.L.amk_opt_num_partials()
        dev = reducer.buckets[0].flat.device
        self.params = [p for b in reducer.buckets for p in b.params]
        self.steps = np.zeros(len(self.params), dtype=np.int64)
        skip = {id(p) for p in no_decay}
        self.wd = np.array([0.0 if id(p) in skip else float(weight_decay) for p in self.params], dtype=np.float64)
        self.flat_p, self.m, self.v, self.seg, self.flat_p16 = [], [], [], [], []
        pid = 0
        for b in reducer.buckets:
            fp = torch.zeros_like(b.flat)
            seg = torch.empty(b.flat.numel() // ALIGN, dtype=torch.int32)
            for p, off in zip(b.params, b.offsets):
                view = fp[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view  # the parameter now lives in the flat buffer
                nseg = -(-p.numel() // ALIGN)
                seg[off // ALIGN: off // ALIGN + nseg] = pid
                pid += 1
            self.flat_p.append(fp)
            self.m.append(torch.zeros_like(b.flat))
            self.v.append(torch.zeros_like(b.flat))
            self.seg.append(seg.to(dev))
        if bf16_shadow:
            self.enable_shadow()
        else:
            self.disable_shadow()   # (copies a deepcopy of a model may have carried over from another optimizer: stale)
        self.partials = torch.zeros(len(reducer.buckets) * self.npart, device=dev, dtype=torch.float32)
        self.norm = torch.zeros(1, device=dev, dtype=torch.float32)
        # per-step table {active, lr / bc1, sqrt(bc2), 0} per parameter: pinned host staging, async copy
        self._tab_host = [torch.zeros(len(self.params), 4).pin_memory() for _ in range(2)]
        self._tab_np = [t.numpy() for t in self._tab_host]  # views of the pinned buffers
        self._tab_dev = [torch.zeros(len(self.params), 4, device=dev) for _ in range(2)]
        self._tab_done = [None, None]
        self._flip = 0
        if self.capturable:
            self.steps_dev = torch.zeros(len(self.params), device=dev, dtype=torch.float64)
            self.lr_dev = torch.full((), float(lr), device=dev, dtype=torch.float64)
            self._fired_host, self._fired_dev = None, None
            self.wd_dev = torch.tensor(self.wd, device=dev)

    def enable_shadow(self):
        """Start keeping bf16 copies of the parameters (see ``bf16_shadow``); idempotent."""
        if self.flat_p16:
            return self.refresh_shadow()
        for b, fp in zip(self.red.buckets, self.flat_p):
            f16 = fp.to(torch.bfloat16)
            for p, off in zip(b.params, b.offsets):
                p._amk_bf16 = f16[off:off + p.numel()].view_as(p)
                p._amk_bf16_version = p._version   # (the update kernel writes behind torch's back: the version stays)
            self.flat_p16.append(f16)

    def disable_shadow(self):
        for p in self.params:
            p.__dict__.pop("_amk_bf16", None)
            p.__dict__.pop("_amk_bf16_version", None)
        self.flat_p16 = []

    def refresh_shadow(self):
        """Re-derive the bf16 copies from the fp32 parameters (after anything but ``step`` wrote them)."""
        for fp, f16 in zip(self.flat_p, self.flat_p16):
            f16.copy_(fp)
        if self.flat_p16:
            for p in self.params:
                p._amk_bf16_version = p._version

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if self.capturable:
            self.lr_dev.fill_(self._lr)  # (a launch with the value as its argument: no host read at replay time)

    def _device_table(self, fired):
        """The step's table, computed on the device (capturable mode)."""
        if self._fired_host is None or not np.array_equal(fired, self._fired_host):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdam: the set of parameters with gradients changed inside a HIP-graph capture; "
                                   "run the step eagerly once (warm-up) before capturing")
            self._fired_host = fired.copy()
            self._fired_dev = torch.tensor(fired.astype(np.float64), device=self.steps_dev.device)
        b1, b2 = self.betas
        self.steps_dev += self._fired_dev
        t = self.steps_dev.clamp(min=1.0)
        tab = self._tab_dev[0]
        tab[:, 0] = self._fired_dev
        tab[:, 1] = self.lr_dev / (1.0 - torch.pow(torch.full_like(t, b1), t))
        tab[:, 2] = torch.sqrt(1.0 - torch.pow(torch.full_like(t, b2), t))
        tab[:, 3] = self.wd_dev * self.lr_dev if self.decoupled else self.wd_dev
        return tab

    def step(self, max_norm=None, lr=None):
        """One optimizer step on the (already reduced) gradients; leaves every gradient zeroed and
        re-attached.  Returns the global gradient norm (device scalar, before clipping)."""
        if lr is not None:
            self.lr = lr
        red, L = self.red, self.L
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        b1, b2 = self.betas
        fired = np.fromiter((f for b in red.buckets for f in b.fired), dtype=bool, count=len(self.params))
        if self.capturable:
            tab_d = self._device_table(fired)
        else:
            # double-buffered table: the copy of step n may still be in flight when step n+1 fills its table
            tab_h, tab_d = self._tab_host[self._flip], self._tab_dev[self._flip]
            if self._tab_done[self._flip] is not None:
                self._tab_done[self._flip].synchronize()  # the copy issued two steps ago has read this host buffer
            done = self._tab_done[self._flip] = torch.cuda.Event()
            self._flip ^= 1
            self.steps += fired
            t = np.maximum(self.steps, 1).astype(np.float64)
            tab = self._tab_np[self._flip ^ 1]
            tab[:, 0] = fired
            tab[:, 1] = self.lr / (1.0 - b1 ** t)
            tab[:, 2] = np.sqrt(1.0 - b2 ** t)
            tab[:, 3] = self.wd * self.lr if self.decoupled else self.wd
            tab_d.copy_(tab_h, non_blocking=True)
            done.record()
        clip = float(max_norm) if max_norm else 0.0
        for k, b in enumerate(red.buckets):
            _lib.check(L.amk_sumsq_partials(_ptr(b.flat), b.flat.numel(), _ptr(self.partials[k * self.npart:]), stream),
                       "amk_sumsq_partials")
        for k, b in enumerate(red.buckets):
            rc = L.amk_adam_flat_step_shadow(
                _ptr(self.flat_p[k]), _ptr(b.flat), _ptr(self.m[k]), _ptr(self.v[k]), b.flat.numel(),
                _ptr(self.seg[k]), _ptr(tab_d), _ptr(self.partials), self.partials.numel(),
                clip, float(self.lr), float(b1), float(b2), float(self.eps), float(self.weight_decay),
                (2 if self.TABLE_WD else 0) | (1 if self.decoupled else 0), _ptr(self.norm) if k == 0 else _ptr(None),
                _ptr(self.flat_p16[k]) if self.flat_p16 else _ptr(None), stream)
            _lib.check(rc, "amk_adam_flat_step")
        red.mark_zeroed()
        return self.norm

    # state in torch.optim.Adam's layout, for checkpoints / comparisons
    def state_of(self, p):
        k = next(i for i, q in enumerate(self.params) if q is p)
        bi = 0
        for b, m, v in zip(self.red.buckets, self.m, self.v):
            if k < bi + len(b.params):
                off = b.offsets[k - bi]
                step = int(self.steps_dev[k].item()) if self.capturable else int(self.steps[k])
                return dict(step=step, exp_avg=m[off:off + p.numel()].view_as(p),
                            exp_avg_sq=v[off:off + p.numel()].view_as(p))
            bi += len(b.params)
        raise KeyError("parameter not managed by this optimizer")
