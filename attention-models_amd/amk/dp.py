"""Data-parallel gradient reducer: one process per GPU, bucketed all-reduce over RCCL/xGMI on a
side HIP stream, overlapped with the rest of backward.

This replaces what ``accelerate.Accelerator.prepare(model)`` injects in the reference
(DistributedDataParallel; trainers/vitgqgan.py:99-109, trainers/utils/base_trainer.py:29-33):
the forward needs no communication, the only exchange is one all-reduce of the fp32
gradients per optimizer step, averaged over ranks.

Design (MI355X: 7 xGMI links x ~153 GB/s per GPU, ring collectives are per-link bound):
  * gradients live in a few large contiguous fp32 buckets (default 32 MiB) laid out in REVERSE
    registration order, so the first bucket to complete in backward is the first one sent;
    ``param.grad`` is a view into its bucket (no flatten / unflatten copies); every parameter
    starts on a 1-KiB boundary of its bucket, which is what lets amk.optim.FlatAdam walk a bucket
    in 256-element segments that each belong to one parameter;
  * a post-accumulate hook per parameter counts a bucket down; when its last gradient lands and
    every EARLIER bucket has been sent, the compute stream records an event, the side stream waits
    on it and issues the all-reduce there (RCCL: ``ReduceOp.AVG``, no separate 1/world pass) --
    compute never blocks.  Buckets always leave in index order, on every rank, whatever the order
    in which autograd finishes them: the collectives of the ranks pair up by construction;
  * ``finish()`` sends buckets whose parameters produced no gradient this step (the
    reference has such parameters: SwitchHeadAttention.W_d, frozen sub-models -- a stock DDP
    wrap would raise), makes the compute stream wait for the side stream, and detaches ``.grad``
    of the parameters that received none since the last ``zero_grad()`` -- torch optimizers then
    skip them as they do in the reference (no weight decay, no moment update);
  * the set of parameters that receive no gradient is taken to be static (``static_unused=True``, DDP's
    ``static_graph``): the first synchronised step records it -- as the UNION over ranks of the parameters that fired,
    one small blocking MAX all-reduce, so that "unused" is decided globally and every rank skips or updates the same
    parameters -- and later steps do not wait for those parameters before sending their buckets (without this, one
    W_d in the first bucket keeps every all-reduce until ``finish()``: no overlap for SwitchHead / Agent models).  A
    recorded-unused parameter that does fire after its bucket left raises; ``reset_static()`` re-records;
  * ``begin(sync=False)`` skips communication for gradient-accumulation micro-steps
    (``accelerator.accumulate`` / ``no_sync`` semantics).
Works unchanged on CPU tensors with the gloo backend (used by the world_size-2 tests).
"""
import os
import weakref

import torch
import torch.distributed as dist

ALIGN = 256  # elements: a parameter's slice of its bucket starts on a 1-KiB boundary


class _Bucket:
    __slots__ = ("flat", "params", "views", "offsets", "pending", "ready", "launched", "work", "fired", "static_unused", "direct")

    def __init__(self, flat, params, views, offsets):
        self.flat, self.params, self.views, self.offsets = flat, params, views, offsets
        self.pending, self.ready, self.launched, self.work = len(params), False, False, None
        self.fired = [False] * len(params)
        self.static_unused = None  # per parameter: never receives a gradient on any rank (recorded by the first sync step)
        self.direct = [False] * len(params)  # per parameter: this step's gradient was written into its view by a kernel


class GradReducer:
    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, communicate_when_alone=False, static_unused=True,
                 direct_grads=False, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradReducer: no trainable parameters")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a world of one needs no collectives; communicate_when_alone=True issues them anyway (RCCL smoke test)
        self.alone = self.world == 1 and not (communicate_when_alone and dist.is_initialized())
        dev = self.params[0].device
        self.on_gpu = dev.type == "cuda"
        # RCCL averages inside the collective; gloo (CPU tests, shared-GPU rehearsal) has no AVG
        self.avg_in_collective = (not self.alone) and dist.get_backend(process_group) == "nccl"
        # overlap=False: the all-reduces are issued on the COMPUTE stream at the point where the bucket completes (no side
        # stream, no events).  For steps captured into a HIP graph whose gradients are small against the step: a graph
        # that forks onto a second stream is launched node by node from the host on this runtime (measured on MI355X:
        # 83 ms of host time per replay of the 2500-launch GAN step against 0.13 ms for the single-stream graph), and a
        # 90 MB all-reduce left un-overlapped costs less than that.
        self.overlap = bool(overlap) and os.environ.get("AMK_DP_OVERLAP", "1") == "1"
        self.side = torch.cuda.Stream(device=dev) if self.on_gpu else None   # (`overlap` may be switched between steps)
        self.sync_step = True
        self.active = False
        self._warned_idle = False
        self.static_unused = bool(static_unused)
        # direct_grads: the weight-gradient kernels of amk.ops write a parameter's FIRST gradient of a step straight into its
        # (zeroed) bucket view -- claim() / wrote() below -- instead of returning a tensor for autograd to add to it: one
        # element-wise launch less per parameter and step.  For models in which a parameter meets one custom backward per
        # step (no weight sharing between such layers): a second weight-gradient kernel for the same parameter after its
        # bucket left raises; contributions from plain autograd ops to such a parameter (weight tying) are not detected.
        self.direct_grads = bool(direct_grads) and os.environ.get("AMK_DIRECT_GRADS", "1") == "1"
        self.buckets = []
        self._bucket_of = {}
        self._next = 0  # index of the next bucket to send
        self.launch_order = []  # bucket indices in the order they left this step (tests read it)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = -(-p.numel() // ALIGN) * ALIGN * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur, dev)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur, dev)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        me = weakref.ref(self)
        for p in self.params:
            p._amk_reducer = me

    # ------------------------------------------------------------------ construction
    def _close(self, params, dev):
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("GradReducer reduces fp32 gradients")
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += -(-p.numel() // ALIGN) * ALIGN
        flat = torch.zeros(off, device=dev, dtype=torch.float32)
        views = []
        for p, o in zip(params, offsets):
            v = flat[o:o + p.numel()].view_as(p)
            p.grad = v
            views.append(v)
        b = _Bucket(flat, params, views, offsets)
        for i, p in enumerate(params):
            self._bucket_of[p] = (b, i)
        self.buckets.append(b)

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s parameters (what DDP does at wrap time)."""
        if not self.alone:
            for p in self.params:
                # through p.detach(), not p.data: the write bumps p._version, which is what version-checked copies of a
                # parameter (the bf16 copies of amk.optim.FlatAdam(bf16_shadow=True)) key on
                dist.broadcast(p.detach(), src=src, group=self.group)

    # ------------------------------------------------------------------ per step
    def begin(self, sync=True):
        """Call before backward.  sync=False: accumulate locally, no communication this step."""
        self.sync_step = sync
        self.active = True   # a backward owned by this reducer is (about to be) running: claim() hands out views
        self._next = 0
        self.launch_order = []
        for b in self.buckets:
            b.pending, b.ready, b.launched, b.work = len(b.params), False, False, None
            b.direct = [False] * len(b.params)
            if b.static_unused is not None:
                b.pending -= sum(b.static_unused)  # not waited for
                b.ready = b.pending == 0           # only such parameters: its zeros leave with the first bucket that completes
            for p, v in zip(b.params, b.views):
                if p.grad is None:
                    p.grad = v  # detached by finish() of an earlier step: accumulate into the bucket again

    def zero_grad(self):
        for b in self.buckets:
            b.flat.zero_()
            b.fired = [False] * len(b.params)
            for p, v in zip(b.params, b.views):
                p.grad = v  # re-attach if finish() or an optimizer set it to None

    def mark_zeroed(self):
        """The buckets were zeroed by someone else (amk.optim.FlatAdam does it inside its update pass)."""
        for b in self.buckets:
            b.fired = [False] * len(b.params)
            for p, v in zip(b.params, b.views):
                p.grad = v

    def claim(self, p):
        """The bucket view to write p's gradient into, if this is the first gradient p receives since the bucket was zeroed
        (else None: the caller returns a tensor and autograd accumulates).  The caller must call wrote(p) after launching
        the kernel that fills the view."""
        if not self.direct_grads or not self.active:
            # outside begin() ... finish() a backward is not this reducer's (torch.autograd.grad on the model, a
            # probe): the caller returns tensors and autograd does what it always does
            return None
        ent = self._bucket_of.get(p)
        if ent is None:
            return None
        b, i = ent
        if b.direct[i] and b.launched and not self.alone:
            # a second weight-gradient kernel for a parameter whose first one was written in place and whose bucket has
            # been sent already: that contribution would miss the all-reduce
            raise RuntimeError("GradReducer(direct_grads=True): a parameter received a second gradient after its bucket was "
                               "sent (a weight shared between two layers?); construct with direct_grads=False")
        if b.fired[i] or b.launched or p.grad is not b.views[i]:
            return None
        return b.views[i]

    def wrote(self, p):
        """p's gradient now sits in its bucket view (written on the current stream): what the post-accumulate hook does."""
        b, i = self._bucket_of[p]
        b.direct[i] = True
        self._count(b, i)

    def _on_grad(self, p):
        b, i = self._bucket_of[p]
        if b.direct[i]:
            # counted already by wrote().  (The post-accumulate hook also runs when the backward returned None for the
            # parameter; a genuine further contribution through plain autograd ops has been added to the view by now --
            # in time only if the bucket has not left: weights tied to non-amk ops need direct_grads=False.)
            return
        view = b.views[i]
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)  # autograd (create_graph) or an optimizer replaced .grad: fold it back
            p.grad = view
        self._count(b, i)

    def _count(self, b, i):
        b.fired[i] = True
        if b.static_unused is not None and b.static_unused[i]:
            if b.launched and not self.alone:
                raise RuntimeError("GradReducer: a parameter recorded as never receiving a gradient received one after its "
                                   "bucket was sent; call reset_static() when the set of used parameters changes "
                                   "(or construct with static_unused=False)")
            return  # not counted in pending
        b.pending -= 1
        if b.pending == 0:
            b.ready = True
            if self.sync_step:
                self._launch_ready()

    def _launch_ready(self):
        while self._next < len(self.buckets) and self.buckets[self._next].ready:
            self._launch(self._next)
            self._next += 1

    def _launch(self, idx):
        b = self.buckets[idx]
        b.launched = True
        self.launch_order.append(idx)
        if self.alone:
            return
        if self.on_gpu and not self.overlap:
            if self.avg_in_collective:
                dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
                b.flat.mul_(1.0 / self.world)
        elif self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                if self.avg_in_collective:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
                    b.flat.mul_(1.0 / self.world)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, detach_unused=True):
        """Call after backward, before clip / optimizer.step()."""
        self.active = False
        if not self.sync_step:
            return
        for b in self.buckets:
            b.ready = True  # parameters without a gradient this step: zeros travel
        self._launch_ready()
        if not self.alone:
            if self.on_gpu:
                if self.overlap:
                    torch.cuda.current_stream().wait_stream(self.side)
            else:
                for b in self.buckets:
                    if b.work is not None:
                        b.work.wait()
                        b.flat.mul_(1.0 / self.world)
                        b.work = None
        self._settle_unused()
        if detach_unused:
            for b in self.buckets:
                for p, f in zip(b.params, b.fired):
                    if not f:
                        p.grad = None  # the reference leaves .grad None here: optimizers skip the parameter

    def reset_static(self):
        """Forget which parameters never receive gradients; the next synchronised step records the set again."""
        for b in self.buckets:
            b.static_unused = None

    def _settle_unused(self):
        """Decide "unused" globally: a parameter is skipped only if it fired on NO rank (otherwise a rank where it did not
        fire would skip an update the others apply, and the replicas drift apart).  Static mode: one blocking MAX
        all-reduce of the fired flags at the first synchronised step, reused afterwards; else one per step."""
        recorded = all(b.static_unused is not None for b in self.buckets)
        if self.static_unused and recorded and not self.alone:
            # DDP's static_graph contract: the recorded set stands.  A recorded-USED parameter that did not fire here may
            # have fired on another rank (its averaged gradient is in the bucket), so it counts as fired; if it fired on
            # NO rank this step the optimizer applies a zero-gradient update (moment decay, weight decay) where the
            # reference would skip the parameter -- said once, loudly, since no collective is spent on finding out.
            idle = [p for b in self.buckets for p, f, u in zip(b.params, b.fired, b.static_unused) if not f and not u]
            if idle and not self._warned_idle:
                import warnings

                self._warned_idle = True
                warnings.warn(f"GradReducer(static_unused=True): {len(idle)} parameter(s) recorded as used received no local "
                              "gradient this step; they are treated as used (a peer may have produced one).  Call "
                              "reset_static() if the set of used parameters changed.")
            for b in self.buckets:
                b.fired = [f or not u for f, u in zip(b.fired, b.static_unused)]
            return
        flags = [f for b in self.buckets for f in b.fired]
        if not self.alone:
            dev = self.buckets[0].flat.device
            t = torch.tensor(flags, dtype=torch.float32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            flags = [bool(x) for x in t.cpu().tolist()]
        k = 0
        for b in self.buckets:
            b.fired = flags[k:k + len(b.params)]
            k += len(b.params)
            if self.static_unused:
                b.static_unused = [not f for f in b.fired]

    def unused_parameters(self):
        """Parameters that received no gradient since the last zero_grad()."""
        return [p for b in self.buckets for p, f in zip(b.params, b.fired) if not f]

    def static_unused_parameters(self):
        """The parameters recorded (by the first synchronised step, globally) as never receiving a gradient."""
        return [p for b in self.buckets if b.static_unused is not None for p, u in zip(b.params, b.static_unused) if u]

    def grads_nbytes(self):
        return sum(b.flat.numel() for b in self.buckets) * 4
