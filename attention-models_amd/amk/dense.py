"""Dense exact-f32 GEMMs of libamk.so (csrc/gemm_f32.hip, ``amk_gemm_f32``) on 2-D row-major tensors, with the
element-wise passes around an nn.Linear folded in: LayerNorm on the input while its tile is staged, bias,
residual add, the SwiGLU gate between w12 and w3 (forward and backward), bias gradients as column sums inside the
weight-gradient product.  Reference: models/softmax_attention.py:30-42,80, models/vitvqgan.py:20-61.

These are the building blocks of ``amk.blocks`` (the fused pre-LN attention / FFN blocks); every function here is a
plain call into the C ABI -- no autograd, no CPU path.
"""
import ctypes

import torch

from . import lib as _lib
from .lib import EPI_BIAS, EPI_RESID, EPI_SWIGLU, EPI_SWIGLU_BWD, GEMM_NN, GEMM_NT, GEMM_TN, GemmDesc


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _mat(t, what):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("amk ops run only on MI355X (HIP) tensors; got a CPU tensor. There is no CPU fallback.")
    if t.dtype != torch.float32:
        raise RuntimeError(f"amk kernels compute in fp32; got {t.dtype} for {what}")
    if t.dim() != 2:
        raise RuntimeError(f"{what} must be a matrix, got shape {tuple(t.shape)}")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _vec(t, n, what):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{what}: an fp32 HIP tensor is required")
    t = t.contiguous()
    if t.numel() != n:
        raise RuntimeError(f"{what}: {t.numel()} elements, expected {n}")
    return t


def supported(*dims):
    """Shapes amk_gemm_f32 takes: every contraction length / leading dimension a multiple of 4."""
    return all(d > 0 and d % 4 == 0 for d in dims)


def _run(d, ws=None):
    L = _lib.load()
    rc = L.amk_gemm_f32(ctypes.byref(d), _p(ws), ws.numel() * 4 if ws is not None else 0, _stream())
    _lib.check(rc, "amk_gemm_f32")


def _set_ln(d, ln, M, K):
    if ln is None:
        return ()
    mean, rstd, gamma, beta = ln
    keep = (_vec(mean, M, "ln mean"), _vec(rstd, M, "ln rstd"), _vec(gamma, K, "ln gamma"), _vec(beta, K, "ln beta"))
    d.ln_mean, d.ln_rstd, d.ln_gamma, d.ln_beta = (_p(t) for t in keep)
    return keep


def row_stats(x2, eps=1e-5):
    """(mean, rstd) of every row of x2 (M, D): nn.LayerNorm's statistics."""
    x2 = _mat(x2, "x")
    M, D = x2.shape
    if x2.stride(0) != D:
        x2 = x2.contiguous()
    mean = torch.empty((M,), device=x2.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    _lib.check(_lib.load().amk_row_stats(_p(x2), M, D, float(eps), _p(mean), _p(rstd), _stream()), "amk_row_stats")
    return mean, rstd


def gemm_nt(a, w, bias=None, *, w2=None, bias2=None, resid=None, ln=None, out=None):
    """F.linear(a', w, bias) (+ resid) with a' = a or LayerNorm(a) from ln = (mean, rstd, gamma, beta).
    With w2: returns (a' w^T + bias, a' w2^T + bias2) from one launch (w.shape[0] a multiple of 128)."""
    a, w, w2, resid = _mat(a, "a"), _mat(w, "w"), _mat(w2, "w2"), _mat(resid, "resid")
    M, K = a.shape
    N1 = w.shape[0]
    N2 = w2.shape[0] if w2 is not None else 0
    if w.shape[1] != K or (w2 is not None and w2.shape[1] != K):
        raise RuntimeError(f"gemm_nt: a {tuple(a.shape)} against w {tuple(w.shape)}")
    c = out if out is not None else torch.empty((M, N1), device=a.device, dtype=torch.float32)
    c2 = torch.empty((M, N2), device=a.device, dtype=torch.float32) if w2 is not None else None
    d = GemmDesc(op=GEMM_NT, epilogue=EPI_RESID if resid is not None else EPI_BIAS, m=M, n=N1 + N2, k=K,
                 split=N1 if w2 is not None else 0)
    bias, bias2 = _vec(bias, N1, "bias"), _vec(bias2, N2, "bias2")
    d.a, d.lda, d.w, d.ldw, d.c, d.ldc, d.bias = _p(a), a.stride(0), _p(w), w.stride(0), _p(c), c.stride(0), _p(bias)
    if w2 is not None:
        d.w2, d.ldw2, d.c2, d.ldc2, d.bias2 = _p(w2), w2.stride(0), _p(c2), c2.stride(0), _p(bias2)
    if resid is not None:
        if resid.shape != (M, N1):
            raise RuntimeError(f"gemm_nt: resid {tuple(resid.shape)} against output {(M, N1)}")
        d.resid, d.ldr = _p(resid), resid.stride(0)
    keep = _set_ln(d, ln, M, K)  # noqa: F841  (keeps the contiguous copies alive until the launch is enqueued)
    _run(d)
    return (c, c2) if w2 is not None else c


def gemm_nt_swiglu(a, w12, b12=None, *, ln=None, keep_ab=True):
    """(a | b) = a' w12^T + b12; returns (silu(a) * b, (a | b) or None)."""
    a, w12 = _mat(a, "a"), _mat(w12, "w12")
    M, K = a.shape
    H = w12.shape[0] // 2
    if w12.shape[1] != K or w12.shape[0] != 2 * H:
        raise RuntimeError(f"gemm_nt_swiglu: a {tuple(a.shape)} against w12 {tuple(w12.shape)}")
    gate = torch.empty((M, H), device=a.device, dtype=torch.float32)
    ab = torch.empty((M, 2 * H), device=a.device, dtype=torch.float32) if keep_ab else None
    d = GemmDesc(op=GEMM_NT, epilogue=EPI_SWIGLU, m=M, n=H, k=K)
    b12 = _vec(b12, 2 * H, "b12")
    d.a, d.lda, d.w, d.ldw, d.bias = _p(a), a.stride(0), _p(w12), w12.stride(0), _p(b12)
    d.gate, d.ldg = _p(gate), H
    d.c, d.ldc = _p(ab), 2 * H  # (NULL without keep_ab: this epilogue then writes the gate only)
    keep = _set_ln(d, ln, M, K)  # noqa: F841
    _run(d)
    return gate, ab


def gemm_nn(a, w, *, a2=None, w2=None, swiglu_ab=None):
    """a @ w (+ a2 @ w2): the input gradient dY W of F.linear.  With swiglu_ab = the forward's (a | b): the product is
    dGate and the result is (dA | dB) (M, 2H)."""
    a, w, a2, w2, swiglu_ab = _mat(a, "a"), _mat(w, "w"), _mat(a2, "a2"), _mat(w2, "w2"), _mat(swiglu_ab, "ab")
    M, K1 = a.shape
    N = w.shape[1]
    K2 = a2.shape[1] if a2 is not None else 0
    if w.shape[0] != K1 or (a2 is not None and (w2 is None or w2.shape != (K2, N) or a2.shape[0] != M)):
        raise RuntimeError(f"gemm_nn: a {tuple(a.shape)} against w {tuple(w.shape)}")
    width = 2 * N if swiglu_ab is not None else N
    c = torch.empty((M, width), device=a.device, dtype=torch.float32)
    d = GemmDesc(op=GEMM_NN, epilogue=EPI_SWIGLU_BWD if swiglu_ab is not None else EPI_BIAS, m=M, n=N, k=K1 + K2,
                 split=K1 if a2 is not None else 0)
    d.a, d.lda, d.w, d.ldw, d.c, d.ldc = _p(a), a.stride(0), _p(w), w.stride(0), _p(c), width
    if a2 is not None:
        d.a2, d.lda2, d.w2, d.ldw2 = _p(a2), a2.stride(0), _p(w2), w2.stride(0)
    if swiglu_ab is not None:
        if swiglu_ab.shape != (M, 2 * N):
            raise RuntimeError(f"gemm_nn: (a | b) {tuple(swiglu_ab.shape)} against dGate {(M, N)}")
        d.ab, d.ldab = _p(swiglu_ab), swiglu_ab.stride(0)
    _run(d)
    return c


def _out(t, shape, what):
    """A caller-provided result buffer (a gradient view of amk.dp.GradReducer): fp32, contiguous, 16-byte aligned."""
    if t is None:
        return None
    if (not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != tuple(shape) or not t.is_contiguous()
            or t.data_ptr() % 16):
        raise RuntimeError(f"{what}: an fp32 contiguous 16-byte aligned HIP tensor of shape {tuple(shape)} is required")
    return t


def gemm_tn(y, x, *, y2=None, ln=None, want_bias=False, out=None, out2=None, bias_out=None):
    """y^T @ x' (x' = x or LayerNorm(x)): the weight gradient dY^T X of F.linear; with y2 a second gradient against the
    same x (y.shape[1] a multiple of 128).  Returns (dw, dw2 or None, colsum(y | y2) or None).  out / out2 / bias_out:
    write the results there (OVERWRITING) instead of into fresh tensors."""
    y, x, y2 = _mat(y, "y"), _mat(x, "x"), _mat(y2, "y2")
    M, N1 = y.shape
    K = x.shape[1]
    N2 = y2.shape[1] if y2 is not None else 0
    if x.shape[0] != M or (y2 is not None and y2.shape[0] != M):
        raise RuntimeError(f"gemm_tn: y {tuple(y.shape)} against x {tuple(x.shape)}")
    dw = _out(out, (N1, K), "gemm_tn out")
    if dw is None:
        dw = torch.empty((N1, K), device=y.device, dtype=torch.float32)
    dw2 = None
    if y2 is not None:
        dw2 = _out(out2, (N2, K), "gemm_tn out2")
        if dw2 is None:
            dw2 = torch.empty((N2, K), device=y.device, dtype=torch.float32)
    db = None
    if want_bias:
        db = _out(bias_out, (N1 + N2,), "gemm_tn bias_out")
        if db is None:
            db = torch.empty((N1 + N2,), device=y.device, dtype=torch.float32)
    d = GemmDesc(op=GEMM_TN, epilogue=EPI_BIAS, m=M, n=N1 + N2, k=K, split=N1 if y2 is not None else 0)
    d.a, d.lda, d.w, d.ldw, d.c, d.ldc, d.dbias = _p(y), y.stride(0), _p(x), x.stride(0), _p(dw), K, _p(db)
    if y2 is not None:
        d.a2, d.lda2, d.c2, d.ldc2 = _p(y2), y2.stride(0), _p(dw2), K
    keep = _set_ln(d, ln, M, K)  # noqa: F841
    L = _lib.load()
    nbytes = L.amk_gemm_f32_ws_bytes(ctypes.byref(d))
    ws = torch.empty((nbytes // 4,), device=y.device, dtype=torch.float32) if nbytes else None
    _run(d, ws)
    return dw, dw2, db


def _mat16(t, what):
    if not t.is_cuda:
        raise RuntimeError("amk ops run only on MI355X (HIP) tensors; got a CPU tensor. There is no CPU fallback.")
    if t.dtype != torch.bfloat16 or t.dim() != 2:
        raise RuntimeError(f"{what} must be a bf16 matrix, got {t.dtype} {tuple(t.shape)}")
    if t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def supported_bf16(N, K):
    """Shapes amk_gemm_tn_bf16 takes: gradient rows and columns multiples of 8."""
    return N > 0 and K > 0 and N % 8 == 0 and K % 8 == 0


def gemm_tn_bf16(y, x, want_bias=False, out=None, bias_out=None):
    """(dW (N, K) f32, db (N,) f32 or None) = (y^T x, column sums of y) for bf16 y (M, N), x (M, K): the weight and bias
    gradient of nn.Linear under autocast (csrc/gemm_bf16.hip).  out / bias_out: write there (OVERWRITING)."""
    y, x = _mat16(y, "y"), _mat16(x, "x")
    M, N = y.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise RuntimeError(f"gemm_tn_bf16: y has {M} rows, x {x.shape[0]}")
    if not supported_bf16(N, K):
        raise RuntimeError(f"gemm_tn_bf16: N and K must be multiples of 8, got {N}, {K}")
    L = _lib.load()
    dw = _out(out, (N, K), "gemm_tn_bf16 out")
    if dw is None:
        dw = torch.empty(N, K, device=y.device, dtype=torch.float32)
    db = None
    if want_bias:
        db = _out(bias_out, (N,), "gemm_tn_bf16 bias_out")
        if db is None:
            db = torch.empty(N, device=y.device, dtype=torch.float32)
    nbytes = L.amk_gemm_tn_bf16_ws_bytes(M, N, K)
    ws = torch.empty(nbytes // 4, device=y.device, dtype=torch.float32) if nbytes else None
    rc = L.amk_gemm_tn_bf16(_p(y), y.stride(0), _p(x), x.stride(0), _p(dw), K, _p(db), M, N, K, _p(ws), nbytes, _stream())
    _lib.check(rc, "amk_gemm_tn_bf16")
    return dw, db


def _bias32(bias, n):
    if bias is None:
        return None
    if bias.dtype != torch.float32 or not bias.is_cuda:
        raise RuntimeError("gemm bf16: the bias stays in fp32 (it is added to the fp32 accumulators)")
    bias = bias.contiguous()
    if bias.numel() != n or bias.data_ptr() % 16:
        raise RuntimeError(f"gemm bf16: bias of {bias.numel()} elements (16-byte aligned) expected {n}")
    return bias


def gemm_nt_bf16(a, w, bias=None):
    """a (M, K) w (N, K)^T + bias -> (M, N), bf16 in and out, fp32 accumulation and bias (csrc/gemm_bf16.hip)."""
    a, w = _mat16(a, "a"), _mat16(w, "w")
    (M, K), N = a.shape, w.shape[0]
    if w.shape[1] != K or not supported_bf16(N, K):
        raise RuntimeError(f"gemm_nt_bf16: a {tuple(a.shape)} x w {tuple(w.shape)}^T (multiples of 8 required)")
    bias = _bias32(bias, N)
    c = torch.empty(M, N, device=a.device, dtype=torch.bfloat16)
    if M:
        rc = _lib.load().amk_gemm_bf16(0, 0, _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(c), N, _p(None), 0, M, N, K, _stream())
        _lib.check(rc, "amk_gemm_bf16")
    return c


def gemm_nn_bf16(a, w):
    """a (M, K) w (K, N) -> (M, N): the input gradient dY W with W as nn.Linear stores it."""
    a, w = _mat16(a, "a"), _mat16(w, "w")
    (M, K), N = a.shape, w.shape[1]
    if w.shape[0] != K or not supported_bf16(N, K):
        raise RuntimeError(f"gemm_nn_bf16: a {tuple(a.shape)} x w {tuple(w.shape)} (multiples of 8 required)")
    c = torch.empty(M, N, device=a.device, dtype=torch.bfloat16)
    if M:
        rc = _lib.load().amk_gemm_bf16(1, 0, _p(a), a.stride(0), _p(w), w.stride(0), _p(None), _p(c), N, _p(None), 0, M, N, K, _stream())
        _lib.check(rc, "amk_gemm_bf16")
    return c


def gemm_nt_swiglu_bf16(a, w12, b12=None, keep_ab=True):
    """(g, ab): ab = a w12^T + b12 (M, 2 H), g = silu(ab[:, :H]) * ab[:, H:] (M, H) from one launch; ab is None with
    keep_ab=False."""
    a, w12 = _mat16(a, "a"), _mat16(w12, "w12")
    (M, K), N = a.shape, w12.shape[0]
    if w12.shape[1] != K or N % 16 or K % 8:
        raise RuntimeError(f"gemm_nt_swiglu_bf16: a {tuple(a.shape)} x w12 {tuple(w12.shape)}^T (2 H a multiple of 16, K of 8)")
    b12 = _bias32(b12, N)
    g = torch.empty(M, N // 2, device=a.device, dtype=torch.bfloat16)
    ab = torch.empty(M, N, device=a.device, dtype=torch.bfloat16) if keep_ab else None
    if M:
        rc = _lib.load().amk_gemm_bf16(0, 1, _p(a), a.stride(0), _p(w12), w12.stride(0), _p(b12), _p(ab), N, _p(g), N // 2, M, N, K, _stream())
        _lib.check(rc, "amk_gemm_bf16")
    return g, ab


def gemm_nn_swiglu_bwd_bf16(dy, w3, ab):
    """(dA | dB) (M, 2 H) for dG = dy (M, K) w3 (K, H) and the forward's ab = (a | b) (M, 2 H), one launch."""
    dy, w3, ab = _mat16(dy, "dy"), _mat16(w3, "w3"), _mat16(ab, "ab")
    (M, K), H = dy.shape, w3.shape[1]
    if w3.shape[0] != K or ab.shape != (M, 2 * H) or not supported_bf16(H, K):
        raise RuntimeError(f"gemm_nn_swiglu_bwd_bf16: dy {tuple(dy.shape)}, w3 {tuple(w3.shape)}, ab {tuple(ab.shape)}")
    dab = torch.empty(M, 2 * H, device=dy.device, dtype=torch.bfloat16)
    if M:
        rc = _lib.load().amk_gemm_bf16_swiglu_bwd(_p(dy), dy.stride(0), _p(w3), w3.stride(0), _p(ab), ab.stride(0), _p(dab), 2 * H,
                                                  M, H, K, _stream())
        _lib.check(rc, "amk_gemm_bf16_swiglu_bwd")
    return dab
