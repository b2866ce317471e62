"""torch.autograd wrappers over the libamk.so C ABI.

torch is plumbing here: it owns device memory, streams and the autograd graph; every FLOP of
the hot path runs in the HIP kernels of ``csrc/``.  Inputs must be fp32 HIP tensors -- there
is deliberately no CPU / eager path ("no fallback").
"""
import ctypes
import os
import weakref

import torch
from torch.autograd.function import once_differentiable

from . import lib as _lib

_NULL = ctypes.c_void_p(0)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else _NULL


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional in-situ kernel timing: set to a dict and every hot-path ABI call is bracketed by two HIP
# events on the launch stream (bench.py does this during its timed steps); None = no events.
KERNEL_EVENTS = None


class _timed:
    """with _timed("name"): <one ABI call>  -- records (start, end) events when KERNEL_EVENTS is a dict."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.b.record()
            KERNEL_EVENTS.setdefault(self.name, []).append((self.a, self.b))
        return False


def kernel_event_summary(events):
    """{name: (launches, average ms)} after a device synchronize."""
    out = {}
    for name, pairs in events.items():
        ms = [a.elapsed_time(b) for a, b in pairs]
        out[name] = (len(ms), sum(ms) / len(ms))
    return out


def _require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "amk ops run only on MI355X (HIP) tensors; got a CPU tensor. There is no CPU fallback."
            )
        if t.is_floating_point() and t.dtype != torch.float32:
            raise RuntimeError(f"amk kernels compute in fp32; got {t.dtype}")


# ---------------------------------------------------------------------------- attention
def _strides4(t):
    """(sb, st, sh) element strides of a (B, H, T, D) view whose last axis is contiguous."""
    return (t.stride(0), t.stride(2), t.stride(1))


def _kernel_view_ok(t):
    return (
        t.stride(3) == 1
        and t.data_ptr() % 16 == 0
        and all(s % 4 == 0 for s in (t.stride(0), t.stride(1), t.stride(2)))
    )


def _as_kernel_view(t):
    """Return a (B,H,T,D) view the kernels can address; copy only if the layout forces it."""
    if _kernel_view_ok(t):
        return t
    B, H, T, D = t.shape
    return t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)  # (B,T,H,D) storage


def _new_bthd(B, H, T, D, like):
    """(B,H,T,D) view over fresh (B,T,H*D) storage: the layout the W_o projection consumes."""
    return torch.empty((B, T, H, D), device=like.device, dtype=torch.float32).permute(0, 2, 1, 3)


def _mask_u8(mask, shape, what):
    if mask is None:
        return None
    m = mask
    while m.dim() > len(shape) and m.shape[0] == 1:
        m = m[0]
    if m.dim() != len(shape):
        raise RuntimeError(f"{what} must have {len(shape)} dims (got shape {tuple(mask.shape)})")
    m = m.expand(*shape)
    return m.to(torch.uint8).contiguous()


# Forward path of the attention core: "f32" = exact-f32 MFMA (v_mfma_f32_32x32x2_f32); "bf16x6" = the same
# kernel with every product formed from three-way bf16 splits of its f32 operands (six exact partial
# products, f32 accumulation): f32-level error, 2.6x the matrix rate.  See csrc/attn_fwd_x6.hip.
ATTENTION_FORWARD = os.environ.get("AMK_ATTENTION_FORWARD", "f32")

# Training: the forward leaves the raw scores S (4 bytes per (b, h, i, j), 32x32 tiles) in HBM and the fused
# backward reads them back instead of recomputing S = QK^T -- four matrix products instead of five, bit for
# bit the same results.  The reference keeps the same tensor alive for autograd; 288 GB of HBM make it
# affordable here (1.07 GB per ViT-VQGAN layer at batch 32, 12.9 GB for the twelve layers).  A call whose
# scores would exceed ATTENTION_KEEP_SCORES_MAX_BYTES, or that would take the scores alive across all layers beyond
# ATTENTION_KEEP_SCORES_BUDGET_BYTES (default: a quarter of the device's memory), recomputes instead -- the same
# results from five products, so a deeper model or a larger batch degrades in speed, not into an out-of-memory error.
ATTENTION_KEEP_SCORES = os.environ.get("AMK_ATTN_KEEP_SCORES", "1") == "1"
ATTENTION_KEEP_SCORES_MAX_BYTES = 8 << 30
ATTENTION_KEEP_SCORES_BUDGET_BYTES = int(os.environ.get("AMK_ATTN_KEEP_BUDGET", "0")) or None   # None: memory / 4
_kept_scores_bytes = [0]   # bytes of kept scores alive right now (released when the tensors are freed)


def _keep_budget(device):
    if ATTENTION_KEEP_SCORES_BUDGET_BYTES is not None:
        return ATTENTION_KEEP_SCORES_BUDGET_BYTES
    return torch.cuda.get_device_properties(device).total_memory // 4


def _release_kept(nbytes):
    _kept_scores_bytes[0] -= nbytes
# keys per workgroup of the fused backward: 0 = library default, 128 or 256
ATTENTION_BACKWARD_KEYS = int(os.environ.get("AMK_ATTN_BWD_KEYS", "0"))


def _attn_forward(q, k, v, key_mask, causal_mask, scale, keep_scores=False):
    """Returns (q, k, v, o, stats, scores); scores is None unless keep_scores and the f32 kernel ran."""
    B, H, I, D = q.shape
    J = k.shape[2]
    _require_device(q, k, v, key_mask, causal_mask)
    if k.shape != (B, H, J, D) or v.shape != (B, H, J, D):
        raise RuntimeError(f"attention shapes disagree: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}")
    q, k, v = _as_kernel_view(q), _as_kernel_view(k), _as_kernel_view(v)
    o = _new_bthd(B, H, I, D, q)
    stats = torch.empty((B, H, I, 2), device=q.device, dtype=torch.float32)
    L = _lib.load()
    x6 = ATTENTION_FORWARD == "bf16x6" and D == 64  # head dims 32 / 128 run the plain f32 kernels
    ws = torch.empty((L.amk_attn_fwd_x6_ws_bytes(B, H, J),), device=q.device, dtype=torch.uint8) if x6 else None
    scores = None
    # (head dim 128: the score-keeping forward exists without masks, and its one-pass backward takes atomics only.  Head dim
    # 32 has both too, but recomputing wins there -- 0.618 against 0.609 of the f32 MFMA peak: the scores are 4 bytes per
    # (query, key) against only 32 multiply-adds, 2.1 GB per launch at the benchmark shape -- so its scores are not kept)
    det = DETERMINISTIC_ATTENTION_BACKWARD or torch.are_deterministic_algorithms_enabled()
    keepable = D == 64 or (D == 128 and key_mask is None and causal_mask is None and not det)
    if (keep_scores and keepable and not x6 and ATTENTION_KEEP_SCORES
            and not ATTENTION_BACKWARD_TWO_KERNEL):
        nbytes = L.amk_attn_scores_bytes(B, H, I, J)
        if nbytes <= ATTENTION_KEEP_SCORES_MAX_BYTES and _kept_scores_bytes[0] + nbytes <= _keep_budget(q.device):
            scores = torch.empty((nbytes // 4,), device=q.device, dtype=torch.float32)
            _kept_scores_bytes[0] += nbytes
            weakref.finalize(scores.untyped_storage(), _release_kept, nbytes)
    with _timed("attn_fwd_keep_kernel" if scores is not None else "attn_fwd_kernel"):
        if scores is not None:
            rc = L.amk_attn_fwd_keep(
                _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(stats), _ptr(scores), _ptr(key_mask), _ptr(causal_mask),
                B, H, I, J, D, *_strides4(q), *_strides4(k), *_strides4(v), *_strides4(o),
                float(scale), _stream(),
            )
        elif x6:
            rc = L.amk_attn_fwd_x6(
                _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(stats), _ptr(ws), _ptr(key_mask), _ptr(causal_mask),
                B, H, I, J, D, *_strides4(q), *_strides4(k), *_strides4(v), *_strides4(o),
                float(scale), _stream(),
            )
        else:
            rc = L.amk_attn_fwd(
                _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(stats), _ptr(key_mask), _ptr(causal_mask),
                B, H, I, J, D, *_strides4(q), *_strides4(k), *_strides4(v), *_strides4(o),
                float(scale), _stream(),
            )
    _lib.check(rc, "amk_attn_fwd")
    return q, k, v, o, stats, scores


# Backward path of the attention core (include/amk.h, amk_attn_bwd `stages`): one fused pass.
# Default: dq accumulated with f32 atomics into a zeroed buffer -- the fastest form (1.135 ms at the ViT-VQGAN layer
# shape); dk, dv are reproducible, dq differs in the last bits from run to run.  Reproducible mode (4 % slower,
# 1.179 ms): dq as plain stores -- directly when one workgroup holds all keys of a (batch, head), else per-key-block
# partials summed in order by a second launch -- every gradient bitwise reproducible.  It is taken when
# DETERMINISTIC_ATTENTION_BACKWARD is set (AMK_DETERMINISTIC=1) or torch.use_deterministic_algorithms(True) is on,
# PyTorch's own convention for atomics-based backward kernels.
DETERMINISTIC_ATTENTION_BACKWARD = os.environ.get("AMK_DETERMINISTIC", "0") == "1"
ATTENTION_BACKWARD_TWO_KERNEL = False


def _attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, key_mask, causal_mask, scale, stages=None, delta=None,
                   scores=None):
    if stages is None:
        det = DETERMINISTIC_ATTENTION_BACKWARD or torch.are_deterministic_algorithms_enabled()
        stages = 7 if ATTENTION_BACKWARD_TWO_KERNEL else (73 if det else 9)
    if stages & 8 and not stages & 48:
        stages |= {128: 16, 256: 32}.get(ATTENTION_BACKWARD_KEYS, 0)
    if not stages & 8:
        scores = None
    B, H, I, D = q.shape
    if D != 64 and stages & 64:
        scores = None   # head dims 32 / 128 under the reproducible mode: the two recompute kernels (no kept scores)
    J = k.shape[2]
    d_o = _as_kernel_view(d_o)
    L = _lib.load()
    need = L.amk_attn_bwd_ws_floats(B, H, I, J, int(stages))
    if delta is None or delta.numel() < need:
        delta = torch.empty((need,), device=q.device, dtype=torch.float32)

    def call(st):
        fn, head = (L.amk_attn_bwd_kept, (_ptr(scores),)) if scores is not None and st & 8 else (L.amk_attn_bwd, ())
        rc = fn(
            *head, _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(stats), _ptr(d_o),
            _ptr(dq), _ptr(dk), _ptr(dv), _ptr(delta), _ptr(key_mask), _ptr(causal_mask),
            B, H, I, J, D,
            *_strides4(q), *_strides4(k), *_strides4(v), *_strides4(o), *_strides4(d_o),
            *_strides4(dq), *_strides4(dk), *_strides4(dv),
            float(scale), int(st), _stream(),
        )
        _lib.check(rc, "amk_attn_bwd")

    if KERNEL_EVENTS is not None and (stages & 1) and (stages & ~1):
        call(1)  # delta on its own so that the events bracket the main kernel(s) only
        name = "attn_bwd_dkdv+dq" if not stages & 8 else ("attn_bwd_fused_kernel(kept scores)" if scores is not None
                                                            else "attn_bwd_fused_kernel")
        with _timed(name):
            call(stages & ~1)
    else:
        call(stages)


# Mixed precision (the reference's shipped config trains under accelerate's bf16 autocast, cfg/vitvqgan.yaml:73): the
# kernels here are f32, so inside a torch.autocast region every autograd Function below takes its floating-point
# inputs as f32 (custom_fwd(cast_inputs=float32): bf16 activations coming out of autocast Linear layers are upcast,
# the op runs with autocast off) and hands f32 back; the library GEMMs around them run in bf16 as autocast decides.
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


class _AttnCore(torch.autograd.Function):
    """o = softmax(fill(q*scale @ k^T)) @ v on (B,H,T,D) views."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, key_mask, causal_mask, scale):
        need = any(ctx.needs_input_grad[:3])
        q, k, v, o, stats, scores = _attn_forward(q, k, v, key_mask, causal_mask, scale, keep_scores=need)
        ctx.save_for_backward(q, k, v, o, stats, key_mask, causal_mask, scores)
        ctx.scale = scale
        return o

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_o):
        q, k, v, o, stats, key_mask, causal_mask, scores = ctx.saved_tensors
        B, H, I, D = q.shape
        J = k.shape[2]
        dq = _new_bthd(B, H, I, D, q)
        dk = _new_bthd(B, H, J, D, q)
        dv = _new_bthd(B, H, J, D, q)
        _attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, key_mask, causal_mask, ctx.scale, scores=scores)
        return dq, dk, dv, None, None, None


class _AttnFusedKV(torch.autograd.Function):
    """Same core fed by the projection outputs in place: q (B,I,h*d), kv (B,J,2*h*d) with the
    reference's '(kv h d)' column order (models/softmax_attention.py:39); returns (B,I,h*d).
    The backward writes dk and dv straight into one (B,J,2*h*d) buffer."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q2, kv2, key_mask, causal_mask, H, D, scale):
        B, I, _ = q2.shape
        J = kv2.shape[1]
        q2 = q2.contiguous()
        kv2 = kv2.contiguous()
        q = q2.view(B, I, H, D).permute(0, 2, 1, 3)
        kv = kv2.view(B, J, 2, H, D)
        k = kv[:, :, 0].permute(0, 2, 1, 3)
        v = kv[:, :, 1].permute(0, 2, 1, 3)
        need = any(ctx.needs_input_grad[:2])
        q, k, v, o, stats, scores = _attn_forward(q, k, v, key_mask, causal_mask, scale, keep_scores=need)
        ctx.save_for_backward(q, k, v, o, stats, key_mask, causal_mask, scores)
        ctx.scale = scale
        return o.permute(0, 2, 1, 3).reshape(B, I, H * D)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_o2):
        q, k, v, o, stats, key_mask, causal_mask, scores = ctx.saved_tensors
        B, H, I, D = q.shape
        J = k.shape[2]
        d_o = d_o2.contiguous().view(B, I, H, D).permute(0, 2, 1, 3)
        dq2 = torch.empty((B, I, H * D), device=q.device, dtype=torch.float32)
        dkv2 = torch.empty((B, J, 2 * H * D), device=q.device, dtype=torch.float32)
        dq = dq2.view(B, I, H, D).permute(0, 2, 1, 3)
        dkv = dkv2.view(B, J, 2, H, D)
        dk = dkv[:, :, 0].permute(0, 2, 1, 3)
        dv = dkv[:, :, 1].permute(0, 2, 1, 3)
        _attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, key_mask, causal_mask, ctx.scale, scores=scores)
        return dq2, dkv2, None, None, None, None, None


# Mixed precision (the reference's shipped bf16 autocast): with bf16 projections coming in and head dim 64 the core runs
# on the bf16-MFMA kernels of csrc/attn_bf16.hip (bf16 q / k / v / o and gradients, f32 scores, softmax and statistics),
# with or without key-padding / causal masks; AMK_ATTENTION_BF16=0 keeps the round-2 behaviour (inputs upcast, exact-f32
# kernels).
ATTENTION_BF16 = os.environ.get("AMK_ATTENTION_BF16", "1") == "1"


class _AttnFusedKVBF16(torch.autograd.Function):
    """_AttnFusedKV on bf16 tensors: q2 (B,I,h*64), kv2 (B,J,2*h*64) bf16 -> (B,I,h*64) bf16."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, q2, kv2, H, scale, key_mask=None, causal_mask=None):
        """key_mask: uint8 (B, J), 1 = keep; causal_mask: uint8 (I, J), 1 = masked (see _mask_u8); or None."""
        B, I, _ = q2.shape
        J = kv2.shape[1]
        D = 64
        q2 = q2.contiguous()
        kv2 = kv2.contiguous()
        o2 = torch.empty((B, I, H * D), device=q2.device, dtype=torch.bfloat16)
        stats = torch.empty((B, H, I, 2), device=q2.device, dtype=torch.float32)
        L = _lib.load()
        qs, kvs = (I * H * D, H * D, D), (J * 2 * H * D, 2 * H * D, D)
        masked = key_mask is not None or causal_mask is not None
        with _timed("attn_bf16_fwd_kernel<masked>" if masked else "attn_bf16_fwd_kernel"):
            rc = L.amk_attn_bf16_fwd(_ptr(q2), _ptr(kv2), ctypes.c_void_p(kv2.data_ptr() + 2 * H * D), _ptr(o2), _ptr(stats),
                                     _ptr(key_mask), _ptr(causal_mask),
                                     B, H, I, J, D, *qs, *kvs, *kvs, *qs, float(scale), _stream())
        _lib.check(rc, "amk_attn_bf16_fwd")
        ctx.save_for_backward(q2, kv2, o2, stats, key_mask, causal_mask)
        ctx.cfg = (H, scale)
        return o2

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_o2):
        q2, kv2, o2, stats, key_mask, causal_mask = ctx.saved_tensors
        H, scale = ctx.cfg
        B, I, _ = q2.shape
        J = kv2.shape[1]
        D = 64
        d_o2 = d_o2.to(torch.bfloat16).contiguous()
        dq2 = torch.empty_like(q2)
        dkv2 = torch.empty_like(kv2)
        L = _lib.load()
        ws = torch.empty((L.amk_attn_bf16_bwd_ws_floats(B, H, I, J),), device=q2.device, dtype=torch.float32)
        qs, kvs = (I * H * D, H * D, D), (J * 2 * H * D, 2 * H * D, D)
        masked = key_mask is not None or causal_mask is not None
        with _timed("attn_bf16_bwd_kernel<masked>" if masked else "attn_bf16_bwd_kernel"):
            rc = L.amk_attn_bf16_bwd(_ptr(q2), _ptr(kv2), ctypes.c_void_p(kv2.data_ptr() + 2 * H * D), _ptr(o2), _ptr(stats), _ptr(d_o2),
                                     _ptr(dq2), _ptr(dkv2), ctypes.c_void_p(dkv2.data_ptr() + 2 * H * D), _ptr(ws),
                                     _ptr(key_mask), _ptr(causal_mask),
                                     B, H, I, J, D, *qs, *kvs, *kvs, *qs, *qs, *qs, *kvs, *kvs, float(scale), _stream())
        _lib.check(rc, "amk_attn_bf16_bwd")
        return dq2, dkv2, None, None, None, None


def attention(q, k, v, scale, key_mask=None, causal_mask=None):
    """Fused attention core on (B,H,T,D) tensors (any strides with a contiguous last axis).

    key_mask: bool (B,J), True = keep (reference ``context_mask``); causal_mask: bool (I,J),
    True = masked (reference ``causal_mask``); both filled with -1e9 as masked_fill does.
    """
    B, H, I, D = q.shape
    J = k.shape[2]
    km = _mask_u8(key_mask, (B, J), "context_mask")
    cm = _mask_u8(causal_mask, (I, J), "causal_mask")
    return _AttnCore.apply(q, k, v, km, cm, scale)


def attention_fused_kv(q2, kv2, num_heads, dim_head, scale, key_mask=None, causal_mask=None):
    """q2 (B,I,h*d), kv2 (B,J,2*h*d) -> (B,I,h*d); see _AttnFusedKV."""
    B, I, _ = q2.shape
    J = kv2.shape[1]
    km = _mask_u8(key_mask, (B, J), "context_mask")
    cm = _mask_u8(causal_mask, (I, J), "causal_mask")
    if ATTENTION_BF16 and q2.dtype == torch.bfloat16 and kv2.dtype == torch.bfloat16 and q2.is_cuda and dim_head == 64:
        return _AttnFusedKVBF16.apply(q2, kv2, num_heads, scale, km, cm)
    return _AttnFusedKV.apply(q2, kv2, km, cm, num_heads, dim_head, scale)


# ---------------------------------------------------------------------------- VQ lookup
# vq_gather (indices_to_embeddings) takes indices from the caller: an index outside the codebook is never
# dereferenced by the kernel, and with CHECK_INDICES the call raises IndexError like nn.Embedding does -- at
# the price of one device-to-host read.  Set False (AMK_CHECK_INDICES=0) inside HIP-graph capture.
CHECK_INDICES = os.environ.get("AMK_CHECK_INDICES", "1") == "1"


def vq_nsplit(N, K):
    """Codebook slices per row block.  The sweep kernel keeps 4 workgroups per CU resident and hides
    its LDS latency behind the other workgroups' MFMAs, so it wants >= 16 workgroups per CU
    (measured at N = 32768, K = 8192: 1 slice 0.545 of the f32 MFMA peak, 2: 0.591, 4: 0.626, 8: 0.642),
    with slices no shorter than 1024 codes (at N = 8192 sixteen slices of 512 lose to eight of 1024)."""
    row_blocks = (N + 127) // 128
    nsplit = 1
    while row_blocks * nsplit < 4096 and K // (2 * nsplit) >= 1024:
        nsplit *= 2
    return nsplit


class _VQLookup(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, z, codebook, beta):
        _require_device(z, codebook)
        C = z.shape[-1]
        K = codebook.shape[0]
        zf = z.contiguous().view(-1, C)
        cb = codebook.contiguous()
        N = zf.shape[0]
        dev = z.device
        L = _lib.load()
        nsplit = vq_nsplit(N, K)
        f32 = dict(device=dev, dtype=torch.float32)
        kp = L.amk_vq_padded_codes(K, nsplit)  # any K: the normalised copy is padded to whole tiles per slice
        en = torch.empty((kp, C), **f32)
        ee = torch.empty((kp,), **f32)
        pmin = torch.empty((N, nsplit), **f32)
        pidx = torch.empty((N, nsplit), device=dev, dtype=torch.int32)
        idx = torch.empty((N,), device=dev, dtype=torch.int64)
        out = torch.empty((N, C), **f32)
        zq = torch.empty((N, C), **f32)
        zn = torch.empty((N, C), **f32)
        partial = torch.empty((L.amk_vq_num_partials(N),), **f32)
        with _timed("vq_lookup_fwd"):
            rc = L.amk_vq_lookup_fwd(
                _ptr(zf), _ptr(cb), N, K, C, nsplit, _ptr(en), _ptr(ee), _ptr(pmin), _ptr(pidx),
                _ptr(idx), _ptr(out), _ptr(zq), _ptr(zn), _ptr(partial), _stream(),
            )
        _lib.check(rc, "amk_vq_lookup_fwd")
        mean_sq = partial.sum() / float(N * C)
        # beta*mean((zq.detach()-z)^2) + mean((zq-z.detach())^2): the two means are one value
        loss = beta * mean_sq + mean_sq
        ctx.save_for_backward(zf, cb, zn, zq, idx)
        ctx.beta = beta
        ctx.z_shape = z.shape
        idx_v = idx.view(z.shape[:-1])
        ctx.mark_non_differentiable(idx_v)
        return out.view(z.shape), idx_v, loss

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, g_out, _g_idx, g_loss):
        zf, cb, zn, zq, idx = ctx.saved_tensors
        N, C = zf.shape
        K = cb.shape[0]
        if g_out is None:
            g_out = torch.zeros_like(zf)
        if g_loss is None:
            g_loss = torch.zeros((), device=zf.device, dtype=torch.float32)
        g_out = g_out.contiguous().view(N, C)
        g_loss = g_loss.contiguous().to(torch.float32)
        dz = torch.empty_like(zf)
        L = _lib.load()
        if DETERMINISTIC_ATTENTION_BACKWARD or torch.are_deterministic_algorithms_enabled():
            # reproducible codebook gradient: per-row contributions, added in a fixed order instead of the kernel's
            # f32 atomics
            ge = torch.empty_like(zf)
            rc = L.amk_vq_lookup_bwd_rows(
                _ptr(zf), _ptr(cb), _ptr(zn), _ptr(zq), _ptr(idx), _ptr(g_out), _ptr(g_loss),
                float(ctx.beta), N, K, C, _ptr(dz), _ptr(ge), _stream(),
            )
            _lib.check(rc, "amk_vq_lookup_bwd_rows")
            # ordered sum per code without touching torch's global determinism switches: rows sorted by code
            # (stable: equal codes keep their row order), then one sequential sum per segment
            flat_idx = idx.clamp(0, K - 1)
            srt, order = torch.sort(flat_idx, stable=True)
            bounds = torch.searchsorted(srt, torch.arange(K + 1, device=srt.device, dtype=srt.dtype))
            dcb = torch.segment_reduce(ge.index_select(0, order), "sum", offsets=bounds, axis=0, unsafe=True)
            return dz.view(ctx.z_shape), dcb, None
        dcb = torch.empty_like(cb)
        rc = L.amk_vq_lookup_bwd(
            _ptr(zf), _ptr(cb), _ptr(zn), _ptr(zq), _ptr(idx), _ptr(g_out), _ptr(g_loss),
            float(ctx.beta), N, K, C, _ptr(dz), _ptr(dcb), _stream(),
        )
        _lib.check(rc, "amk_vq_lookup_bwd")
        return dz.view(ctx.z_shape), dcb, None


def vq_lookup(z, codebook, beta):
    """Codebook.forward of the reference: returns (z_q straight-through, int64 indices, loss)."""
    return _VQLookup.apply(z, codebook, beta)


def vq_gather(indices, codebook):
    """Codebook.indices_to_embeddings: l2norm(E[indices]).  Differentiable w.r.t. nothing
    (the reference only uses it for decoding under no_grad / frozen VQ)."""
    _require_device(indices, codebook)
    K, C = codebook.shape
    flat = indices.contiguous().view(-1).to(torch.int64)
    N = flat.shape[0]
    out = torch.empty((N, C), device=codebook.device, dtype=torch.float32)
    bad = torch.zeros((1,), device=codebook.device, dtype=torch.int32)
    L = _lib.load()
    rc = L.amk_vq_gather(_ptr(flat), _ptr(codebook.detach().contiguous()), N, K, C, _ptr(out), _ptr(bad), _stream())
    _lib.check(rc, "amk_vq_gather")
    if CHECK_INDICES and int(bad.item()):  # one host read per call; nn.Embedding raises here too
        raise IndexError(f"vq_gather: {int(bad.item())} of {N} indices are outside the codebook [0, {K})")
    return out.view(*indices.shape, C)


# ---------------------------------------------------------------------------- masked-token sampling step
_SAMPLE_CALLS = 0


def sample_step(logits, ids, mask=None, null_logits=None, cfg_scale=3.0, tau=1.0, p=0.9, gumbel=None, seed=None,
                unmasked_score=None, generator=None):
    """One step of the parallel decode on (B, T, V) logits, in one pass (csrc/sample.hip): classifier-free
    guidance, softmax, top-(1-p) filter, Gumbel-argmax at temperature tau, chosen probability.
    ids (B, T) int64 is updated IN PLACE where mask (B, T) bool is True (everywhere without a mask);
    returns scores (B, T).  gumbel: explicit noise (B, T, V) -- else drawn in-kernel: Philox4x32-10 keyed by the seed
    and stream offset of torch's CUDA generator (`generator`, default the device's), which is advanced by the call, so
    torch.manual_seed reproduces a run; an explicit `seed` uses this module's call counter as the offset instead.
    Ties at the keep-th largest logit: every logit >= that value is kept (the reference's topk keeps exactly k)."""
    import math

    global _SAMPLE_CALLS
    _require_device(logits, null_logits, gumbel)
    B, T, V = logits.shape
    logits = logits.contiguous()
    null_logits = null_logits.contiguous() if null_logits is not None else None
    gumbel = gumbel.contiguous() if gumbel is not None else None
    if ids.dtype != torch.int64 or not ids.is_contiguous():
        raise RuntimeError("sample_step: ids must be a contiguous int64 tensor (updated in place)")
    m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
    scores = torch.empty((B, T), device=logits.device, dtype=torch.float32)
    if seed is not None:
        _SAMPLE_CALLS += 1          # explicit seed: the stream position is this module's call counter
        offset = _SAMPLE_CALLS
    else:
        # torch's generator for this device supplies the Philox key and stream position, and is advanced past what
        # this call consumes: torch.manual_seed(s); generate(); torch.manual_seed(s); generate() repeats the samples
        gen = generator if generator is not None else torch.cuda.default_generators[logits.device.index or 0]
        seed = int(gen.initial_seed()) & ((1 << 63) - 1)
        offset = int(gen.get_offset())
        gen.set_offset(offset + 4 * ((B * T * V + 3) // 4))
    L = _lib.load()
    rc = L.amk_sample_step(_ptr(logits), _ptr(null_logits), float(cfg_scale), _ptr(gumbel), seed, offset, float(tau),
                           B * T, V, math.ceil((1 - p) * V), _ptr(m8), -1.0 if unmasked_score is None else float(unmasked_score),
                           _ptr(ids), _ptr(scores), _stream())
    _lib.check(rc, "amk_sample_step")
    return scores


# ---------------------------------------------------------------------------- routed experts
def _i32(n, dev):
    return torch.empty((n,), device=dev, dtype=torch.int32)


def moe_route(logits2, k):
    """logits2 (U,E) -> dict(ids int64 (U,k), gate (U,k), offsets int32 (E+1), perm int32 (U*k))."""
    _require_device(logits2)
    U, E = logits2.shape
    dev = logits2.device
    ids = torch.empty((U, k), device=dev, dtype=torch.int64)
    gate = torch.empty((U, k), device=dev, dtype=torch.float32)
    L = _lib.load()
    counts, rank = _i32(E, dev), _i32(U * k, dev)
    blockhist = _i32(L.amk_moe_route_ws_ints(U, E, k), dev)
    offsets, perm = _i32(E + 1, dev), _i32(U * k, dev)
    rc = L.amk_moe_route(_ptr(logits2.contiguous()), U, E, k, _ptr(ids), _ptr(gate), _ptr(counts), _ptr(rank),
                         _ptr(blockhist), _ptr(offsets), _ptr(perm), _stream())
    _lib.check(rc, "amk_moe_route")
    return dict(ids=ids, gate=gate, offsets=offsets, perm=perm)


# Sums over the pairs of a row inside the expert GEMM (f32 atomics, amk_grouped_gemm_*_acc) instead of a (pairs, width)
# intermediate and an ordered pass over it.  OFF by default: measured at the ViTMoE SwitchHead layer (16 pairs per row,
# 68 M element adds) the atomics cost more than the 272 MB intermediate they remove -- output experts 0.237 ms against
# 0.131 + 0.065 ms, V experts' input gradient 0.246 against 0.140 + 0.065 -- and the ordered combine keeps the
# reference's accumulation order.  AMK_MOE_SUM_IN_GEMM=1 turns it on (never under the deterministic switches).
MOE_SUM_IN_GEMM = os.environ.get("AMK_MOE_SUM_IN_GEMM", "0") == "1"


# Where a row sums many pairs -- SwitchHead: 8 heads x top-2 = 16 pairs per token over 32 experts -- the sum over the pairs
# of (pair row) x (its expert's matrix) is one dense product of the per-expert sums of the pair rows, Z (G, E*d), with the
# stacked expert matrices (E*d, N): E/fan times the routed FLOPs, but no (pairs, N) intermediate to write and re-read
# (272 MB per launch at the ViTMoE layer).  Measured there: 0.147 + 0.074 ms (routed GEMM + combine) against
# ~0.17 ms (sums + dense GEMM), and 0.163 + 0.074 against ~0.155 for the V experts' input gradient.  The sums differ
# from the ordered combine in accumulation order only; both are reproducible.  AMK_MOE_DENSE_Z=0: the routed form.
MOE_DENSE_Z = os.environ.get("AMK_MOE_DENSE_Z", "1") != "0"


def _moe_dense_z(width, depth, fan, E):
    """width: output row length; depth: contraction length per pair; fan: pairs per output row; E experts."""
    return MOE_DENSE_Z and E <= 2 * fan and width >= 256 and width >= 4 * depth and depth % 4 == 0


def _expert_sums(a2, a_div, ids, scale, G, fan, E, d):
    """Z (G, E*d): per output row and expert, the (scaled) sum of the rows of a2 its pairs read."""
    Z = torch.empty((G, E * d), device=a2.device, dtype=torch.float32)
    with _timed(f"moe_expert_sums G{G} fan{fan} E{E} d{d}"):
        rc = _lib.load().amk_moe_expert_sums(_ptr(a2), a2.stride(0), a_div, _ptr(ids), scale, G, fan, E, d, _ptr(Z), _stream())
    _lib.check(rc, "amk_moe_expert_sums")
    return Z


def _moe_sum_in_gemm(width, depth, fan):
    """width: output row length; depth: contraction length; fan: pairs per output row."""
    if not MOE_SUM_IN_GEMM or DETERMINISTIC_ATTENTION_BACKWARD or torch.are_deterministic_algorithms_enabled():
        return False
    return fan >= 4 and width >= 128 and depth % 32 == 0


class _RoutedLinear(torch.autograd.Function):
    """Top-k routed expert Linear + ordered combine, one launch per stage for ALL experts.

    x2 (Rx,Kd); logits2 (U,E) with U*k pairs, pair p reads x row p // x_div; W (E,N,Kd);
    bias (E,N) or None.  out[g] = sum over the `outer` units of group g and their k slots (in
    ascending expert id) of gate*expert(x) (weighted) or expert(x) (un-weighted, SwitchHead
    moe_out).  Returns (out (U/outer, N), ids (U,k) int64)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x2, logits2, W, bias, k, x_div, weighted, outer):
        _require_device(x2, logits2, W, bias)
        x2 = x2.contiguous()
        W = W.contiguous()
        U, E = logits2.shape
        N, Kd = W.shape[1], W.shape[2]
        P = U * k
        dev = x2.device
        if x2.shape != (P // x_div, Kd) or U % outer:
            raise RuntimeError(f"routed linear shapes disagree: x {tuple(x2.shape)} U {U} k {k} x_div {x_div} W {tuple(W.shape)}")
        with _timed(f"moe_route U{U} E{E} k{k}"):
            r = moe_route(logits2.detach(), k)
        L = _lib.load()
        G = U // outer
        if not weighted and bias is None and _moe_dense_z(N, Kd, outer * k, E):
            Y = None
            Z = _expert_sums(x2, x_div, r["ids"], _NULL, G, outer * k, E, Kd)
            with _timed(f"dense_z_gemm M{G} N{N} K{E * Kd}"):
                out = Z @ W.permute(0, 2, 1).reshape(E * Kd, N)   # (E, N, Kd) -> (E*Kd, N): an 8 MB copy at the ViTMoE layer
        elif not weighted and _moe_sum_in_gemm(N, Kd, outer * k):
            # un-weighted sum over the pairs of an output row (SwitchHead's output experts): folded into the GEMM's
            # epilogue -- the (P, N) per-pair intermediate (272 MB at the ViTMoE layer) is never written
            Y = None
            out = torch.zeros((G, N), device=dev, dtype=torch.float32)
            with _timed(f"grouped_nt_acc P{P} N{N} K{Kd}"):
                rc = L.amk_grouped_gemm_nt_acc(_ptr(x2), Kd, x_div, _ptr(W), _ptr(bias), _ptr(r["offsets"]), _ptr(r["perm"]),
                                               P, E, N, Kd, _ptr(out), outer * k, _stream())
            _lib.check(rc, "amk_grouped_gemm_nt_acc")
        else:
            Y = torch.empty((P, N), device=dev, dtype=torch.float32)
            with _timed(f"grouped_nt P{P} N{N} K{Kd}"):
                rc = L.amk_grouped_gemm_nt(_ptr(x2), Kd, x_div, _ptr(W), _ptr(bias), _ptr(r["offsets"]), _ptr(r["perm"]),
                                           P, E, N, Kd, _ptr(Y), _stream())
            _lib.check(rc, "amk_grouped_gemm_nt")
            out = torch.empty((G, N), device=dev, dtype=torch.float32)
            with _timed(f"moe_combine G{G} N{N} x{outer * k}"):
                rc = L.amk_moe_combine(_ptr(Y), _ptr(r["ids"]), _ptr(r["gate"]) if weighted else _NULL, G, outer, k, N,
                                       _ptr(out), _stream())
            _lib.check(rc, "amk_moe_combine")
        ctx.save_for_backward(x2, W, Y, r["ids"], r["gate"], r["offsets"], r["perm"])
        ctx.cfg = (k, x_div, weighted, outer, E, bias is not None)
        ctx.mark_non_differentiable(r["ids"])
        ctx.set_materialize_grads(False)   # no zero-filled "gradient" of the ids handed to backward
        return out, r["ids"]

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_out, _d_ids):
        if d_out is None:
            return (None,) * 8
        x2, W, Y, ids, gate, offsets, perm = ctx.saved_tensors
        k, x_div, weighted, outer, E, has_bias = ctx.cfg
        N, Kd = W.shape[1], W.shape[2]
        P = ids.numel()
        U = P // k
        dev = x2.device
        d_out = d_out.contiguous()
        g_div = outer * k
        scale = _ptr(gate) if weighted else _NULL
        L = _lib.load()
        dlogits = None
        if weighted:
            dlogits = torch.empty((U, E), device=dev, dtype=torch.float32)
            rc = L.amk_moe_gate_grad(_ptr(d_out), _ptr(Y), _ptr(ids), _ptr(gate), P, k, E, N, g_div, _ptr(dlogits), _stream())
            _lib.check(rc, "amk_moe_gate_grad")
        if not ctx.needs_input_grad[0]:
            dx = None
        elif _moe_dense_z(Kd, N, x_div, E):
            # dx[r] = sum over the x_div pairs of input row r of scale * dOut-row x W[e] (N, Kd): per-expert sums of the
            # dOut rows, then one product with W viewed as (E*N, Kd) -- no copy
            Z = _expert_sums(d_out, g_div, ids, scale, x2.shape[0], x_div, E, N)
            with _timed(f"dense_z_gemm M{x2.shape[0]} N{Kd} K{E * N}"):
                dx = Z @ W.view(E * N, Kd)
        elif _moe_sum_in_gemm(Kd, N, x_div):
            # the input rows' gradients are sums over the x_div pairs that read them: folded into the GEMM's epilogue
            dx = torch.zeros_like(x2)
            with _timed(f"grouped_nn_acc P{P} N{N} K{Kd}"):
                rc = L.amk_grouped_gemm_nn_acc(_ptr(d_out), N, g_div, _ptr(W), scale, _ptr(offsets), _ptr(perm), P, E, N, Kd,
                                               _ptr(dx), x_div, _stream())
            _lib.check(rc, "amk_grouped_gemm_nn_acc")
        else:
            dxp = torch.empty((P, Kd), device=dev, dtype=torch.float32)
            with _timed(f"grouped_nn P{P} N{N} K{Kd}"):
                rc = L.amk_grouped_gemm_nn(_ptr(d_out), N, g_div, _ptr(W), scale, _ptr(offsets), _ptr(perm), P, E, N, Kd,
                                           _ptr(dxp), _stream())
            _lib.check(rc, "amk_grouped_gemm_nn")
            dx = torch.empty_like(x2)
            rc = L.amk_moe_combine(_ptr(dxp), _ptr(ids), _NULL, x2.shape[0], x_div // k, k, Kd, _ptr(dx), _stream())
            _lib.check(rc, "amk_moe_combine")
        dW = torch.empty_like(W)
        db = torch.empty((E, N), device=dev, dtype=torch.float32) if has_bias else None
        with _timed(f"grouped_wgrad P{P} N{N} K{Kd}"):
            rc = L.amk_grouped_gemm_wgrad(_ptr(d_out), N, g_div, _ptr(x2), Kd, x_div, scale, _ptr(offsets), _ptr(perm),
                                          P, E, N, Kd, _ptr(dW), _ptr(db), _stream())
        _lib.check(rc, "amk_grouped_gemm_wgrad")
        return dx, dlogits, dW, db, None, None, None, None


def routed_linear(x2, logits2, W, bias, k, x_div, weighted=True, outer=1):
    return _RoutedLinear.apply(x2, logits2, W, bias, k, x_div, weighted, outer)


def _topk(logits2, k):
    """(ids int64 (U,k), gate (U,k)): amk_moe_route's selection stage alone."""
    U, E = logits2.shape
    ids = torch.empty((U, k), device=logits2.device, dtype=torch.int64)
    gate = torch.empty((U, k), device=logits2.device, dtype=torch.float32)
    _lib.check(_lib.load().amk_moe_topk(_ptr(logits2.contiguous()), U, E, k, _ptr(ids), _ptr(gate), _stream()), "amk_moe_topk")
    return ids, gate


def _route_distinct(ids, G, fan, E):
    """(offsets (E+1), perm): the distinct (row group, expert) combinations as virtual pairs g*E + e, by expert."""
    dev = ids.device
    mask = torch.empty((G,), device=dev, dtype=torch.int64)
    offsets, perm = _i32(E + 1, dev), _i32(G * min(fan, E), dev)
    _lib.check(_lib.load().amk_moe_route_distinct(_ptr(ids), G, fan, E, _ptr(mask), _ptr(offsets), _ptr(perm), _stream()),
               "amk_moe_route_distinct")
    return offsets, perm


def distinct_experts_ok(dim, d, fan, E, *weights):
    """The SwitchHead forms below apply: a token's fan = h*k pairs are at least E/2, at most 64 experts, model rows at least
    256 wide and 4x the head dim (ops._moe_dense_z), everything on the GPU."""
    return E <= 64 and _moe_dense_z(dim, d, fan, E) and dim % 4 == 0 and all(w.is_cuda for w in weights)


class _SharedRowExperts(torch.autograd.Function):
    """SwitchHead's V experts (reference switchhead_attention.py:58-73): U = G*H units, the H units (heads) of group g
    (token) all read x row g; out[u] = sum over the unit's k slots of gate * x[g] W[e]^T.  The product x[g] W[e]^T does not
    depend on the head, so it is formed once per DISTINCT (token, expert) -- amk_moe_route_distinct's lists feed the same
    grouped GEMMs -- and fanned out to the units that chose it; backward: per-expert sums of the (gated) output gradients,
    then dx as one dense product and dW from the distinct rows.  Returns (out (U, N), ids (U, k))."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x2, logits2, W, k, H):
        _require_device(x2, logits2, W)
        x2, W = x2.contiguous(), W.contiguous()
        U, E = logits2.shape
        N, Kd = W.shape[1], W.shape[2]
        G, fan = U // H, H * k
        if x2.shape != (G, Kd) or U % H:
            raise RuntimeError(f"shared-row experts: shapes disagree: x {tuple(x2.shape)} U {U} H {H} W {tuple(W.shape)}")
        L = _lib.load()
        with _timed(f"moe_topk U{U} E{E} k{k} + distinct lists"):
            ids, gate = _topk(logits2.detach(), k)
            offsets, perm = _route_distinct(ids, G, fan, E)
        V = torch.empty((G * E, N), device=x2.device, dtype=torch.float32)   # rows of chosen (token, expert) only
        with _timed(f"grouped_nt P{U * k} N{N} K{Kd} (distinct rows)"):
            _lib.check(L.amk_grouped_gemm_nt(_ptr(x2), Kd, E, _ptr(W), _NULL, _ptr(offsets), _ptr(perm), G * E, E, N, Kd, _ptr(V),
                                             _stream()), "amk_grouped_gemm_nt")
        out = torch.empty((U, N), device=x2.device, dtype=torch.float32)
        _lib.check(L.amk_moe_combine_rows(_ptr(V), _ptr(ids), _ptr(gate), U, 1, k, N, fan, E, _ptr(out), _stream()),
                   "amk_moe_combine_rows")
        ctx.save_for_backward(x2, W, V, ids, gate, offsets, perm)
        ctx.cfg = (k, H)
        ctx.mark_non_differentiable(ids)
        ctx.set_materialize_grads(False)
        return out, ids

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_out, _d_ids):
        if d_out is None:
            return (None,) * 5
        x2, W, V, ids, gate, offsets, perm = ctx.saved_tensors
        k, H = ctx.cfg
        E, N, Kd = W.shape
        G, fan = x2.shape[0], H * k
        U = G * H
        d_out = d_out.contiguous()
        L = _lib.load()
        dlogits = torch.empty((U, E), device=x2.device, dtype=torch.float32)
        _lib.check(L.amk_moe_gate_grad_rows(_ptr(d_out), _ptr(V), _ptr(ids), _ptr(gate), U * k, k, E, N, k, fan, _ptr(dlogits),
                                            _stream()), "amk_moe_gate_grad_rows")
        Z = _expert_sums(d_out, k, ids, _ptr(gate), G, fan, E, N)      # (G, E*N): gated output gradients per (token, expert)
        dx = None
        if ctx.needs_input_grad[0]:
            with _timed(f"dense_z_gemm M{G} N{Kd} K{E * N}"):
                dx = Z @ W.view(E * N, Kd)
        dW = torch.empty_like(W)
        with _timed(f"grouped_wgrad P{U * k} N{N} K{Kd} (distinct rows)"):
            _lib.check(L.amk_grouped_gemm_wgrad(_ptr(Z), N, 1, _ptr(x2), Kd, E, _NULL, _ptr(offsets), _ptr(perm), G * E, E, N, Kd,
                                                _ptr(dW), _NULL, _stream()), "amk_grouped_gemm_wgrad")
        return dx, dlogits, dW, None, None


class _SummedExperts(torch.autograd.Function):
    """SwitchHead's output experts + head sum (reference switchhead_attention.py:75-88,115): U = G*H units with their own
    rows a2[u]; out[g] = sum over the group's H*k pairs of a2[u] W[e]^T, un-weighted.  Forward: per-expert sums of the rows,
    one dense product.  Backward: d_out[g] W[e] once per distinct (token, expert), fanned out to the units; dW from the
    distinct rows of the forward's sums.  Returns (out (G, N), ids (U, k)); the logits get no gradient (un-weighted)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, a2, logits2, W, k, H):
        _require_device(a2, logits2, W)
        a2, W = a2.contiguous(), W.contiguous()
        U, E = logits2.shape
        N, Kd = W.shape[1], W.shape[2]
        G, fan = U // H, H * k
        if a2.shape != (U, Kd) or U % H:
            raise RuntimeError(f"summed experts: shapes disagree: a {tuple(a2.shape)} U {U} H {H} W {tuple(W.shape)}")
        with _timed(f"moe_topk U{U} E{E} k{k}"):
            ids, _gate = _topk(logits2.detach(), k)
        Z = _expert_sums(a2, k, ids, _NULL, G, fan, E, Kd)
        with _timed(f"dense_z_gemm M{G} N{N} K{E * Kd}"):
            out = Z @ W.permute(0, 2, 1).reshape(E * Kd, N)
        ctx.save_for_backward(W, Z, ids)
        ctx.cfg = (k, H)
        ctx.mark_non_differentiable(ids)
        ctx.set_materialize_grads(False)
        return out, ids

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_out, _d_ids):
        if d_out is None:
            return (None,) * 5
        W, Z, ids = ctx.saved_tensors
        k, H = ctx.cfg
        E, N, Kd = W.shape
        G, fan = Z.shape[0], H * k
        U = G * H
        d_out = d_out.contiguous()
        L = _lib.load()
        offsets, perm = _route_distinct(ids, G, fan, E)
        da = None
        if ctx.needs_input_grad[0]:
            D = torch.empty((G * E, Kd), device=W.device, dtype=torch.float32)
            with _timed(f"grouped_nn P{U * k} N{N} K{Kd} (distinct rows)"):
                _lib.check(L.amk_grouped_gemm_nn(_ptr(d_out), N, E, _ptr(W), _NULL, _ptr(offsets), _ptr(perm), G * E, E, N, Kd, _ptr(D),
                                                 _stream()), "amk_grouped_gemm_nn")
            da = torch.empty((U, Kd), device=W.device, dtype=torch.float32)
            _lib.check(L.amk_moe_combine_rows(_ptr(D), _ptr(ids), _NULL, U, 1, k, Kd, fan, E, _ptr(da), _stream()),
                       "amk_moe_combine_rows")
        dW = torch.empty_like(W)
        with _timed(f"grouped_wgrad P{U * k} N{N} K{Kd} (distinct rows)"):
            _lib.check(L.amk_grouped_gemm_wgrad(_ptr(d_out), N, E, _ptr(Z), Kd, 1, _NULL, _ptr(offsets), _ptr(perm), G * E, E, N, Kd,
                                                _ptr(dW), _NULL, _stream()), "amk_grouped_gemm_wgrad")
        return da, None, dW, None, None


def shared_row_experts(x2, logits2, W, k, H):
    return _SharedRowExperts.apply(x2, logits2, W, k, H)


def summed_experts(a2, logits2, W, k, H):
    return _SummedExperts.apply(a2, logits2, W, k, H)


# ---------------------------------------------------------------------------- agent attention
class _AgentAttn(torch.autograd.Function):
    """qkv (B,T,3*h*d) with the reference's '(qkv h d)' column order -> o (B,T,h*d)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, qkv2, conv_w, conv_b, H, D, P, scale):
        _require_device(qkv2, conv_w, conv_b)
        B, T, _ = qkv2.shape
        qkv2 = qkv2.contiguous()
        qkv = qkv2.view(B, T, 3, H, D)
        q, k, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        dev = qkv2.device
        o = _new_bthd(B, H, T, D, qkv2)
        agents = torch.empty((B, H, P, D), device=dev, dtype=torch.float32)
        vagent = torch.empty_like(agents)
        stats1 = torch.empty((B, H, P, 2), device=dev, dtype=torch.float32)
        cw, cb = conv_w.contiguous(), conv_b.contiguous()
        L = _lib.load()
        ws = torch.empty((L.amk_agent_ws_floats(B, H, T, P, 0),), device=dev, dtype=torch.float32)
        rc = L.amk_agent_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(cw), _ptr(cb), _ptr(o), _ptr(agents), _ptr(vagent),
                                  _ptr(stats1), _ptr(ws), B, H, T, D, P, *_strides4(q), *_strides4(k), *_strides4(v), *_strides4(o),
                                  float(scale), _stream())
        _lib.check(rc, "amk_agent_attn_fwd")
        ctx.save_for_backward(qkv2, cw, agents, vagent, stats1)
        ctx.cfg = (H, D, P, scale)
        return o.permute(0, 2, 1, 3).reshape(B, T, H * D)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, d_o2):
        qkv2, cw, agents, vagent, stats1 = ctx.saved_tensors
        H, D, P, scale = ctx.cfg
        B, T, _ = qkv2.shape
        dev = qkv2.device
        qkv = qkv2.view(B, T, 3, H, D)
        q, k, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        d_o = d_o2.contiguous().view(B, T, H, D).permute(0, 2, 1, 3)
        dqkv2 = torch.empty_like(qkv2)
        dqkv = dqkv2.view(B, T, 3, H, D)
        dq, dk, dv = (dqkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        L = _lib.load()
        cells = B * H * L.amk_agent_num_chunks(T)
        ws = torch.empty((L.amk_agent_ws_floats(B, H, T, P, 1),), device=dev, dtype=torch.float32)
        dw_part = torch.empty((cells, 9, D), device=dev, dtype=torch.float32)
        db_part = torch.empty((cells, D), device=dev, dtype=torch.float32)
        rc = L.amk_agent_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(cw), _ptr(d_o), _ptr(agents), _ptr(vagent), _ptr(stats1),
                                  _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), _ptr(dw_part), _ptr(db_part),
                                  B, H, T, D, P, *_strides4(q), *_strides4(k), *_strides4(v), *_strides4(d_o),
                                  *_strides4(dq), *_strides4(dk), *_strides4(dv), float(scale), _stream())
        _lib.check(rc, "amk_agent_attn_bwd")
        dconv_w = torch.empty((D, 1, 3, 3), device=dev, dtype=torch.float32)
        dconv_b = torch.empty((D,), device=dev, dtype=torch.float32)
        _lib.check(L.amk_agent_conv_grad_reduce(_ptr(dw_part), _ptr(db_part), cells, D, _ptr(dconv_w), _ptr(dconv_b), _stream()),
                   "amk_agent_conv_grad_reduce")
        return dqkv2, dconv_w, dconv_b, None, None, None, None


def agent_attention(qkv2, conv_w, conv_b, num_heads, dim_head, pool, scale):
    return _AgentAttn.apply(qkv2, conv_w, conv_b, num_heads, dim_head, pool, scale)


# ---------------------------------------------------------------------------- SwiGLU gate
class _SwiGLU(torch.autograd.Function):
    FWD, BWD = "amk_swiglu_fwd", "amk_swiglu_bwd"

    @classmethod
    def forward(cls, ctx, ab):
        _require_device(ab)
        H = ab.shape[-1] // 2
        ab2 = ab.contiguous().view(-1, 2 * H)
        M = ab2.shape[0]
        out = torch.empty((M, H), device=ab.device, dtype=torch.float32)
        L = _lib.load()
        _lib.check(getattr(L, cls.FWD)(_ptr(ab2), M, H, _ptr(out), _stream()), cls.FWD)
        ctx.save_for_backward(ab2)
        ctx.shape = ab.shape
        return out.view(*ab.shape[:-1], H)

    @classmethod
    def backward(cls, ctx, d_out):
        (ab2,) = ctx.saved_tensors
        M, H2 = ab2.shape
        d_out = d_out.contiguous().view(M, H2 // 2)
        d_ab = torch.empty_like(ab2)
        L = _lib.load()
        _lib.check(getattr(L, cls.BWD)(_ptr(ab2), _ptr(d_out), M, H2 // 2, _ptr(d_ab), _stream()), cls.BWD)
        return d_ab.view(ctx.shape)


class _GEGLU(_SwiGLU):
    FWD, BWD = "amk_geglu_fwd", "amk_geglu_bwd"


def _f32_outside_autocast(fn, x):
    """(the gate Functions are classmethod-based: no custom_fwd; under autocast the input is upcast here)"""
    if torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return fn(x.float())
    return fn(x)


class _SwiGLUBF16(torch.autograd.Function):
    """The gate on bf16 tensors (mixed precision): bf16 in, bf16 out, f32 arithmetic (csrc/mixed_bf16.hip)."""

    @staticmethod
    def forward(ctx, ab):
        H = ab.shape[-1] // 2
        ab2 = ab.contiguous().view(-1, 2 * H)
        M = ab2.shape[0]
        out = torch.empty((M, H), device=ab.device, dtype=torch.bfloat16)
        _lib.check(_lib.load().amk_swiglu_bf16_fwd(_ptr(ab2), M, H, _ptr(out), _stream()), "amk_swiglu_bf16_fwd")
        ctx.save_for_backward(ab2)
        ctx.shape = ab.shape
        return out.view(*ab.shape[:-1], H)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        (ab2,) = ctx.saved_tensors
        M, H2 = ab2.shape
        d_out = d_out.to(torch.bfloat16).contiguous().view(M, H2 // 2)
        d_ab = torch.empty_like(ab2)
        _lib.check(_lib.load().amk_swiglu_bf16_bwd(_ptr(ab2), _ptr(d_out), M, H2 // 2, _ptr(d_ab), _stream()), "amk_swiglu_bf16_bwd")
        return d_ab.view(ctx.shape)


# Mixed precision: bf16-in / bf16-out element-wise kernels between the bf16 GEMMs (no separate cast passes);
# AMK_MIXED_ELEMENTWISE=0 returns to the round-2 behaviour (inputs upcast, f32 kernels, outputs cast by autocast).
MIXED_ELEMENTWISE = os.environ.get("AMK_MIXED_ELEMENTWISE", "1") == "1"


def swiglu(ab):
    """silu(a) * b for ab = (..., 2H) = (a | b): one HBM pass forward, one backward."""
    if MIXED_ELEMENTWISE and ab.dtype == torch.bfloat16 and ab.is_cuda and ab.shape[-1] % 8 == 0:
        with torch.autocast("cuda", enabled=False):
            return _SwiGLUBF16.apply(ab)
    return _f32_outside_autocast(_SwiGLU.apply, ab)


def geglu(ab):
    """gelu(a) * b for ab = (..., 2H) = (a | b) (exact erf GELU): the transformer FFN gate."""
    return _f32_outside_autocast(_GEGLU.apply, ab)


# ---------------------------------------------------------------------------- residual + LayerNorm
def _ln_supported(D):
    return 0 < D <= 4096 and D % 4 == 0


def _ln_param_grads(part, params):
    """(d gamma, d beta) from the kernel's (rows, 2, D) partial sums; summed straight into the reducer's gradient views
    where they can be claimed (amk.dp.GradReducer(direct_grads=True): those come back as None)."""
    weight, bias = params if params is not None else (None, None)
    (wr, wv), (br, bv) = _claim_late(weight), _claim_late(bias)
    if wv is None and bv is None:
        dgb = part.sum(0)
        return dgb[0], dgb[1]
    dw = db = None
    if wv is not None:
        torch.sum(part[:, 0], dim=0, out=wv)
        wr.wrote(weight)
    else:
        dw = part[:, 0].sum(0)
    if bv is not None:
        torch.sum(part[:, 1], dim=0, out=bv)
        br.wrote(bias)
    else:
        db = part[:, 1].sum(0)
    return dw, db


def _claim_late(p):
    return _claim(p)   # (_claim is defined with the Linear layers below)


def _ln_backward(dy, h, dh_in, weight, mean, rstd, params=None):
    M, D = h.shape
    L = _lib.load()
    dh = torch.empty_like(h)
    part = torch.empty((L.amk_rowsum_num_partials(M), 2, D), device=h.device, dtype=torch.float32)
    _lib.check(L.amk_add_layernorm_bwd(_ptr(dy), _ptr(h), _ptr(dh_in), _ptr(weight), _ptr(mean), _ptr(rstd),
                                       M, D, _ptr(dh), _ptr(part), _stream()), "amk_add_layernorm_bwd")
    dw, db = _ln_param_grads(part, params)
    return dh, dw, db


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias, eps):
        _require_device(x, weight, bias)
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        M = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty((M,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        w, b = weight.contiguous(), bias.contiguous()
        L = _lib.load()
        _lib.check(L.amk_add_layernorm_fwd(_ptr(x2), _NULL, _ptr(w), _ptr(b), M, D, float(eps), _NULL, _ptr(y),
                                           _ptr(mean), _ptr(rstd), _stream()), "amk_add_layernorm_fwd")
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape = x.shape
        ctx.params = (weight, bias)
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        x2, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = _ln_backward(dy.contiguous().view(x2.shape), x2, None, w, mean, rstd, ctx.params)
        return dx.view(ctx.shape), dw, db, None


class _AddLayerNorm(torch.autograd.Function):
    """(x, res) -> (h, y) with h = x + res, y = LN(h); the gradient of h is added inside the LN backward."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, res, weight, bias, eps):
        _require_device(x, res, weight, bias)
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        r2 = res.contiguous().view(-1, D)
        M = x2.shape[0]
        h = torch.empty_like(x2)
        y = torch.empty_like(x2)
        mean = torch.empty((M,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        w, b = weight.contiguous(), bias.contiguous()
        L = _lib.load()
        _lib.check(L.amk_add_layernorm_fwd(_ptr(x2), _ptr(r2), _ptr(w), _ptr(b), M, D, float(eps), _ptr(h), _ptr(y),
                                           _ptr(mean), _ptr(rstd), _stream()), "amk_add_layernorm_fwd")
        ctx.save_for_backward(h, w, mean, rstd)
        ctx.shape = x.shape
        ctx.params = (weight, bias)
        ctx.set_materialize_grads(False)
        return h.view(x.shape), y.view(x.shape)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dh_in, dy):
        h, w, mean, rstd = ctx.saved_tensors
        if dy is None:  # y unused downstream: only the residual stream carries gradient
            if dh_in is None:
                return None, None, None, None, None
            return dh_in, dh_in, None, None, None
        dh2 = dh_in.contiguous().view(h.shape) if dh_in is not None else None
        dh, dw, db = _ln_backward(dy.contiguous().view(h.shape), h, dh2, w, mean, rstd, ctx.params)
        dh = dh.view(ctx.shape)
        return dh, dh, dw, db, None


class _AddLayerNormMixed(torch.autograd.Function):
    """(x, res) -> (h, y) for the mixed-precision mode: x f32 or bf16 (a branch's output), res f32 (the residual stream)
    or None; h = x + res in f32 (x itself without a residual), y = LN(h) in bf16 for the bf16 GEMMs that consume it.
    Backward: dy bf16 or f32; the gradient of the residual stream stays f32, the branch gets it rounded to bf16."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        _require_device(res, weight, bias)
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        M = x2.shape[0]
        xb = x2.dtype == torch.bfloat16
        if not xb and x2.dtype != torch.float32:
            raise RuntimeError(f"mixed LayerNorm takes f32 or bf16 inputs, got {x2.dtype}")
        r2 = res.contiguous().view(-1, D) if res is not None else None
        need_h = r2 is not None or xb
        h = torch.empty((M, D), device=x.device, dtype=torch.float32) if need_h else None
        y = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
        mean = torch.empty((M,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        w, b = weight.contiguous(), bias.contiguous()
        _lib.check(_lib.load().amk_add_layernorm_mixed_fwd(_ptr(x2), 1 if xb else 0, _ptr(r2), _ptr(w), _ptr(b), M, D, float(eps),
                                                          _ptr(h), _ptr(y), _ptr(mean), _ptr(rstd), _stream()),
                   "amk_add_layernorm_mixed_fwd")
        hs = h if need_h else x2
        ctx.save_for_backward(hs, w, mean, rstd)
        ctx.cfg = (x.shape, xb, res is not None)
        ctx.params = (weight, bias)
        ctx.set_materialize_grads(False)
        return hs.view(x.shape), y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dh_in, dy):
        h, w, mean, rstd = ctx.saved_tensors
        shape, xb, has_res = ctx.cfg
        M, D = h.shape
        if dy is None:
            if dh_in is None:
                return None, None, None, None, None
            dx = dh_in.to(torch.bfloat16) if xb else dh_in
            return dx, (dh_in if has_res else None), None, None, None
        dyb = dy.dtype == torch.bfloat16
        dy2 = (dy if dyb else dy.float()).contiguous().view(M, D)
        dh2 = dh_in.float().contiguous().view(M, D) if dh_in is not None else None
        L = _lib.load()
        dh = torch.empty((M, D), device=h.device, dtype=torch.float32)
        dh16 = torch.empty((M, D), device=h.device, dtype=torch.bfloat16) if xb else None
        part = torch.empty((L.amk_rowsum_num_partials(M), 2, D), device=h.device, dtype=torch.float32)
        _lib.check(L.amk_add_layernorm_mixed_bwd(_ptr(dy2), 1 if dyb else 0, _ptr(h), _ptr(dh2), _ptr(w), _ptr(mean), _ptr(rstd),
                                                 M, D, _ptr(dh), _ptr(dh16), _ptr(part), _stream()), "amk_add_layernorm_mixed_bwd")
        dw, db = _ln_param_grads(part, ctx.params)
        dhv = dh.view(shape)
        dx = dh16.view(shape) if xb else dhv
        return dx, (dhv if has_res else None), dw, db, None


def _mixed_ln():
    return (MIXED_ELEMENTWISE and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def layer_norm(x, weight, bias, eps=1e-5, branch=False):
    """LayerNorm over the last axis (one HBM pass forward, one backward incl. the gamma / beta partials).
    branch=True: the result only feeds Linear layers -- under bf16 autocast it is then produced in bf16 directly."""
    if not _ln_supported(x.shape[-1]) or x.numel() == 0 or not x.is_cuda:
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), weight, bias, eps)
    if branch and _mixed_ln():
        with torch.autocast("cuda", enabled=False):
            return _AddLayerNormMixed.apply(x, None, weight, bias, eps)[1]
    return _LayerNorm.apply(x, weight, bias, eps)


def add_layer_norm(x, res, weight, bias, eps=1e-5, branch=False):
    """h = x + res, y = LayerNorm(h): returns (h, y) from one kernel (amk_add_layernorm_fwd).  branch: see layer_norm."""
    if not _ln_supported(x.shape[-1]) or x.numel() == 0 or not x.is_cuda:
        h = x + res
        return h, torch.nn.functional.layer_norm(h, (h.shape[-1],), weight, bias, eps)
    if branch and _mixed_ln():
        with torch.autocast("cuda", enabled=False):
            return _AddLayerNormMixed.apply(x, res.float(), weight, bias, eps)
    return _AddLayerNorm.apply(x, res, weight, bias, eps)


# ---------------------------------------------------------------------------- Linear with bias
def colsum(x2):
    """Column sums of a contiguous (M, N) matrix (the bias gradient of a Linear)."""
    M, N = x2.shape
    if N % 4 != 0 or N > 1024:  # wide matrices: torch's reduction is as fast (measured at N = 2736: 71 vs 88 us)
        return x2.sum(0)
    L = _lib.load()
    part = torch.empty((L.amk_rowsum_num_partials(M), N), device=x2.device, dtype=torch.float32)
    _lib.check(L.amk_colsum(_ptr(x2), M, N, _ptr(part), _stream()), "amk_colsum")
    return part.sum(0)


class _BiasLinear(torch.autograd.Function):
    """F.linear with bias; the GEMMs stay with the vendor library (same calls autograd would make, so
    the shipped TunableOp selections apply), the bias gradient is amk_colsum instead of a strided sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_device(x, weight, bias)
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, dy.shape[-1])
        dx = dy2.mm(weight).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = dy2.t().mm(x.reshape(-1, x.shape[-1])) if ctx.needs_input_grad[1] else None
        db = colsum(dy2) if ctx.needs_input_grad[2] else None
        return dx, dw, db


# Dense GEMMs of the nn.Linear layers: "f32" = the vendor library's exact-f32 kernels (rocBLAS / hipBLASLt, chosen
# by TunableOp; 0.79 of the f32 MFMA peak) -- the parity mode and the headline; "bf16x6" = csrc/gemm_x6.hip for the
# forward and the input gradient (split-bf16 products: f32-level error, bound = bf16 MFMA peak / 6); the weight
# gradient stays with the library.
GEMM_MODE = os.environ.get("AMK_GEMM", "f32")


def gemm_x6_nt(a2, b2, bias=None):
    """a2 (M, K) @ b2 (N, K)^T (+ bias): F.linear on 2-D operands through amk_gemm_x6_split + amk_gemm_x6_nt
    (b2, the weight, is split into bf16 planes once per call; a2 inside the GEMM)."""
    _require_device(a2, b2, bias)
    if a2.stride(1) != 1:
        a2 = a2.contiguous()
    if b2.stride(1) != 1:
        b2 = b2.contiguous()
    M, K = a2.shape
    N = b2.shape[0]
    L = _lib.load()
    planes = torch.empty((L.amk_gemm_x6_planes_bytes(N, K),), device=a2.device, dtype=torch.uint8)
    c = torch.empty((M, N), device=a2.device, dtype=torch.float32)
    with _timed(f"gemm_x6_nt M{M} N{N} K{K}"):
        _lib.check(L.amk_gemm_x6_split(_ptr(b2), b2.stride(0), N, K, _ptr(planes), _stream()), "amk_gemm_x6_split")
        rc = L.amk_gemm_x6_nt(_ptr(a2), a2.stride(0), _ptr(planes), _ptr(bias), _ptr(c), N, M, N, K, _stream())
    _lib.check(rc, "amk_gemm_x6_nt")
    return c


def _x6_ok(x, weight):
    """The preconditions of amk_gemm_x6_nt (csrc/gemm_x6.hip), so that other shapes take the library GEMM instead of
    an error from the C side: rows 16-byte aligned (K and the leading dimensions multiples of 4), operands < 2 GiB."""
    K = weight.shape[1]
    if K % 4 or x.numel() == 0 or x.data_ptr() % 16 or weight.data_ptr() % 16:
        return False
    if x.stride(-1) != 1 or (x.dim() >= 2 and x.stride(-2) % 4):
        return False
    return x.numel() * 4 < 0x7ffffff0 and weight.numel() * 8 < 0x7ffffff0   # (A's span; the three bf16 planes of W, padded)


class _LinearX6(torch.autograd.Function):
    """F.linear with the forward and the input gradient on the split-bf16 GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_device(x, weight, bias)
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        ctx.has_bias = bias is not None
        return gemm_x6_nt(x2, weight, bias).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            N = weight.shape[0]
            if N % 4 == 0:
                dx = gemm_x6_nt(dy2, weight.t().contiguous()).view(ctx.x_shape)   # dY W = dY (W^T)^T
            else:
                dx = dy2.mm(weight).view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dw = dy2.t().mm(x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dw, db


# Dense GEMMs of the nn.Linear layers, default path: csrc/gemm_f32.hip (amk_gemm_f32) -- the forward with the bias in
# its epilogue, the weight gradient with the bias gradient (column sums of dY) inside the same product, q and kv
# projections as one launch, the SwiGLU gate of the FFN in the w12 GEMM's epilogue and its derivative in the epilogue
# of the w3 input gradient.  DENSE_MODE (AMK_DENSE): "amk" = every product on amk_gemm_f32; "auto" (default) = the
# plain input gradient dY W and forwards with fewer than 512 output columns stay with the vendor library where that is
# the faster kernel (measured, tools/kbench_dense.py: 0.80-0.96x), everything else on amk_gemm_f32; "lib" = the
# round-2 path (library GEMMs + amk_colsum / amk_swiglu launches).
DENSE_MODE = os.environ.get("AMK_DENSE", "auto")


# `auto` keeps the own kernels to contractions of at most this many input features.  They were built for (and win or tie
# on) the ViT-VQGAN layers, K = 256 / 512: short contractions, where the vendor kernels are epilogue-dominated and the
# persistent walk hides the epilogue.  At K >= 1024 -- the ViT classifier of configs[1], the ViTMoE projections -- the tuned
# vendor kernels are 10-45 % faster (tools/kbench_vit.py: 7.5 against 9.8 ms per step at batch 64; tools/scratch/dense_z_probe.py).
DENSE_AUTO_MAX_K = int(os.environ.get("AMK_DENSE_MAX_K", "512"))


def _dense_ok(x, *weights):
    if DENSE_MODE == "lib" or torch.is_autocast_enabled() or not x.is_cuda or x.dtype != torch.float32 or x.numel() == 0:
        return False
    if x.shape[-1] % 4 or x.shape[-1] < 128:
        return False
    if DENSE_MODE == "auto" and x.shape[-1] > DENSE_AUTO_MAX_K:
        return False
    return all(w.dtype == torch.float32 and w.shape[0] % 4 == 0 and w.shape[0] >= 128 and w.is_contiguous() for w in weights)


def _nn_plain(dy2, weight):
    """dY W: the one product the library still wins on the step's shapes."""
    from . import dense

    if DENSE_MODE == "amk":
        return dense.gemm_nn(dy2, weight)
    return dy2.mm(weight)


def _claim(p):
    """(reducer, view): the gradient view of amk.dp.GradReducer(direct_grads=True) to write parameter p's gradient into,
    when this is its first gradient of the step -- else (None, None) and the caller returns a tensor for autograd."""
    ref = getattr(p, "_amk_reducer", None) if p is not None else None
    red = ref() if ref is not None else None
    if red is None:
        return None, None
    view = red.claim(p)
    return (red, view) if view is not None else (None, None)


class _DenseLinear(torch.autograd.Function):
    """F.linear on amk_gemm_f32: forward NT (+ bias), dX = dY W, dW = dY^T X with db = colsum(dY) in the same launch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import dense

        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        if DENSE_MODE == "auto" and weight.shape[0] < 512:
            # outputs of one or two column tiles (W_o, w3, patch / quant layers: one tile per workgroup, nothing for the
            # persistent walk to overlap): the library's kernel is 5-10 % faster there (tools/kbench_dense.py)
            return torch.nn.functional.linear(x, weight, bias)
        return dense.gemm_nt(x2, weight, bias).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from . import dense

        x2, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _nn_plain(dy2, weight).view(ctx.x_shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            (wr, wv), (br, bv) = _claim(ctx.params[0]), (_claim(ctx.params[1]) if want_b else (None, None))
            dw, _, db = dense.gemm_tn(dy2, x2, want_bias=want_b, out=wv, bias_out=bv)
            if wv is not None:
                wr.wrote(ctx.params[0])
                dw = None
            if bv is not None:
                br.wrote(ctx.params[1])
                db = None
        return dx, dw, db


def _tn_into(tn, y, x, weight, bias):
    """(dW, db) of a Linear layer by `tn` (dense.gemm_tn / gemm_tn_bf16), written straight into the reducer's gradient views
    where they can be claimed (those come back as None: nothing for autograd to add)."""
    (wr, wv), (br, bv) = _claim(weight), _claim(bias)
    res = tn(y, x, want_bias=bias is not None, out=wv, bias_out=bv)
    dw, db = res[0], res[-1]
    if wv is not None:
        wr.wrote(weight)
        dw = None
    if bv is not None:
        br.wrote(bias)
        db = None
    return dw, db


class _DenseLinear2(torch.autograd.Function):
    """Two bias-free projections of the same input (q and kv of SoftmaxAttention) as one launch each way:
    (x Wa^T, x Wb^T); dX = dA Wa + dB Wb as one contraction in two segments; (dWa, dWb) from one launch."""

    @staticmethod
    def forward(ctx, x, wa, wb):
        from . import dense

        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, wa, wb)
        ctx.x_shape = x.shape
        ctx.params = (wa, wb)
        a, b = dense.gemm_nt(x2, wa, None, w2=wb)
        return a.view(*x.shape[:-1], wa.shape[0]), b.view(*x.shape[:-1], wb.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, da, db_):
        from . import dense

        x2, wa, wb = ctx.saved_tensors
        da2 = da.reshape(-1, da.shape[-1])
        db2 = db_.reshape(-1, db_.shape[-1])
        dx = dense.gemm_nn(da2, wa, a2=db2, w2=wb).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        dwa = dwb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            (ar, av), (br, bv) = _claim(ctx.params[0]), _claim(ctx.params[1])
            dwa, dwb, _ = dense.gemm_tn(da2, x2, y2=db2, out=av, out2=bv)
            if av is not None:
                ar.wrote(ctx.params[0])
                dwa = None
            if bv is not None:
                br.wrote(ctx.params[1])
                dwb = None
        return dx, dwa, dwb


class _SwiGLUFFN(torch.autograd.Function):
    """w3(silu(a) * b) with (a | b) = w12(x) (models/vitvqgan.py:20-34 as built, see amk/models/vitvqgan.py): the gate in
    the epilogue of the w12 product ((a | b) is written only when a backward will read it), its derivative in the epilogue
    of dGate = dOut W3, both bias gradients inside the weight-gradient products: six launches forward + backward where
    the separate form took eighteen."""

    @staticmethod
    def forward(ctx, x, w12, b12, w3, b3):
        from . import dense

        x2 = x.reshape(-1, x.shape[-1])
        need = any(ctx.needs_input_grad)
        gate, ab = dense.gemm_nt_swiglu(x2, w12, b12, keep_ab=need)
        out = dense.gemm_nt(gate, w3, b3)
        if need:
            ctx.save_for_backward(x2, w12, w3, ab, gate)
        ctx.x_shape = x.shape
        ctx.has_bias = (b12 is not None, b3 is not None)
        ctx.params = (w12, b12, w3, b3)
        return out.view(*x.shape[:-1], w3.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        from . import dense

        x2, w12, w3, ab, gate = ctx.saved_tensors
        d2 = dout.reshape(-1, dout.shape[-1])
        d_ab = dense.gemm_nn(d2, w3, swiglu_ab=ab)
        dw3, db3 = _tn_into(dense.gemm_tn, d2, gate, ctx.params[2], ctx.params[3] if ctx.has_bias[1] else None)
        dx = _nn_plain(d_ab, w12).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        dw12, db12 = _tn_into(dense.gemm_tn, d_ab, x2, ctx.params[0], ctx.params[1] if ctx.has_bias[0] else None)
        return dx, dw12, db12, dw3, db3


# AMK_MIXED_WGRAD=0: nn.Linear under bf16 autocast entirely by the library and autograd (the round-2 behaviour).
MIXED_WGRAD = os.environ.get("AMK_MIXED_WGRAD", "1") == "1"


def _w16(t):
    """The bf16 copy of a parameter: the one FlatAdam(bf16_shadow=True) keeps current, else a cast."""
    if t is None:
        return None
    shadow = getattr(t, "_amk_bf16", None)
    if shadow is not None and t._version == t._amk_bf16_version:
        return shadow
    return t.to(torch.bfloat16)   # no copy kept, or the parameter was written in place since (load_state_dict, init): cast


class _LinearMixed(torch.autograd.Function):
    """nn.Linear under bf16 autocast: forward and input gradient by the library's bf16 GEMMs as autocast runs them,
    weight and bias gradient by amk_gemm_tn_bf16 -- one pass over dY and X, f32 results (csrc/gemm_bf16.hip; the library
    runs these products at 50-180 TFLOP/s, rounds dW to bf16 and leaves db to a separate reduction)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x16, w16 = x.to(torch.bfloat16), _w16(weight)
        ctx.save_for_backward(x16, w16)
        ctx.has_bias, ctx.x_dtype = bias is not None, x.dtype
        ctx.params = (weight, bias)
        return torch.nn.functional.linear(x16, w16, _w16(bias))

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from . import dense

        x16, w16 = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dy2.mm(w16).view(x16.shape).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        dw, db = _tn_into(dense.gemm_tn_bf16, dy2, x16.reshape(-1, x16.shape[-1]), ctx.params[0], ctx.params[1])
        return dx, dw, db


def _mixed_linear_ok(x, weight):
    return (MIXED_WGRAD and x.is_cuda and torch.get_autocast_dtype("cuda") == torch.bfloat16 and torch.is_grad_enabled()
            and weight.requires_grad and weight.dtype == torch.float32 and x.dtype in (torch.float32, torch.bfloat16)
            and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0 and x.numel() > 0)


def linear(x, weight, bias=None):
    if torch.is_autocast_enabled():  # mixed precision: the library GEMMs in the autocast dtype, own weight gradient
        if _mixed_linear_ok(x, weight):
            return _LinearMixed.apply(x, weight, bias)
        return torch.nn.functional.linear(x, weight, bias)
    if GEMM_MODE == "bf16x6" and _x6_ok(x, weight):
        return _LinearX6.apply(x, weight, bias)
    if _dense_ok(x, weight):
        return _DenseLinear.apply(x, weight, bias)
    if bias is None:
        return torch.nn.functional.linear(x, weight)
    return _BiasLinear.apply(x, weight, bias)


def linear2(x, wa, wb):
    """(F.linear(x, wa), F.linear(x, wb)) -- one launch when the shapes allow (wa.shape[0] a multiple of 128)."""
    if GEMM_MODE != "bf16x6" and _dense_ok(x, wa, wb) and wa.shape[0] % 128 == 0:
        return _DenseLinear2.apply(x, wa, wb)
    return linear(x, wa), linear(x, wb)


class _SwiGLUFFNMixed(torch.autograd.Function):
    """The SwiGLU FFN under bf16 autocast (models/vitvqgan.py:20-34 under cfg/vitvqgan.yaml:73): w12 with the gate in its
    epilogue (amk_gemm_bf16, epi 1: (a | b) and silu(a) b from one launch), the gate's input gradient dY W3 by
    amk_gemm_bf16 (op 1), both weight / bias gradients by amk_gemm_tn_bf16; w3's forward and w12's input gradient stay
    on the library, which is faster on those two shapes (tools/kbench_tn_bf16.py)."""

    @staticmethod
    def forward(ctx, x, w12, b12, w3, b3):
        from . import dense

        x16 = x.to(torch.bfloat16).reshape(-1, x.shape[-1])
        w12h, w3h = _w16(w12), _w16(w3)
        g, ab = dense.gemm_nt_swiglu_bf16(x16, w12h, b12)
        y = torch.nn.functional.linear(g, w3h, _w16(b3))
        ctx.save_for_backward(x16, ab, g, w12h, w3h)
        ctx.x_shape, ctx.x_dtype, ctx.bias = x.shape, x.dtype, (b12 is not None, b3 is not None)
        ctx.params = (w12, b12, w3, b3)
        return y.view(*x.shape[:-1], w3.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from . import dense

        x16, ab, g, w12h, w3h = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dw3, db3 = _tn_into(dense.gemm_tn_bf16, dy2, g, ctx.params[2], ctx.params[3])
        dab = dense.gemm_nn_swiglu_bwd_bf16(dy2, w3h, ab)   # dY W3 with the gate's backward in its epilogue
        dw12, db12 = _tn_into(dense.gemm_tn_bf16, dab, x16, ctx.params[0], ctx.params[1])
        dx = dab.mm(w12h).view(ctx.x_shape).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        return dx, dw12, db12, dw3, db3


def swiglu_ffn(x, w12, b12, w3, b3):
    """w3(silu(a) * b), (a | b) = w12(x): fused (see _SwiGLUFFN) when the shapes allow, else the separate launches."""
    if torch.is_autocast_enabled() and w12.shape[0] % 16 == 0 and w12.shape[0] >= 1024:
        if _mixed_linear_ok(x, w12) and _mixed_linear_ok(x, w3) and w3.requires_grad:
            with torch.autocast("cuda", enabled=False):
                return _SwiGLUFFNMixed.apply(x, w12, b12, w3, b3)
        no_grad = not torch.is_grad_enabled() or not any(
            t is not None and t.requires_grad for t in (x, w12, b12, w3, b3))   # (e.g. the generator's forward in the D phase)
        if (MIXED_WGRAD and no_grad and x.is_cuda and torch.get_autocast_dtype("cuda") == torch.bfloat16
                and w12.dtype == torch.float32 and x.dtype in (torch.float32, torch.bfloat16) and w12.shape[1] % 8 == 0 and x.numel()):
            from . import dense   # inference: the gate only, the pre-activations never reach HBM

            g, _ = dense.gemm_nt_swiglu_bf16(x.to(torch.bfloat16).reshape(-1, x.shape[-1]), _w16(w12), b12, keep_ab=False)
            return torch.nn.functional.linear(g, _w16(w3), _w16(b3)).view(*x.shape[:-1], w3.shape[0])
    if GEMM_MODE != "bf16x6" and _dense_ok(x, w12) and w3.shape[1] % 4 == 0 and w3.shape[0] % 4 == 0 and w3.is_contiguous():
        return _SwiGLUFFN.apply(x, w12, b12, w3, b3)
    ab = linear(x, w12, b12)
    if ab.shape[-1] % 8 == 0:
        return linear(swiglu(ab), w3, b3)
    a, b = ab.chunk(2, dim=-1)
    return linear(torch.nn.functional.silu(a) * b, w3, b3)
