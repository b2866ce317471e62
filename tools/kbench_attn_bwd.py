"""Attention core at the ViT-VQGAN layer shape: forward with / without kept scores, fused backward with
128 / 256 keys per workgroup, recomputed / kept scores.
    python tools/kbench_attn_bwd.py [--batch 32] [--iters 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))

import torch  # noqa: E402

from bench import time_launches  # noqa: E402

PEAK = 157.3

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--H", type=int, default=8)
    a = ap.parse_args()
    from amk import lib, ops

    L_ws = lambda B_, H_, I_, J_: lib.load().amk_attn_bwd_ws_floats(B_, H_, I_, J_, 8 | 64 | 32)

    dev = torch.device("cuda:0")
    B, H, T, D = a.batch, a.H, a.T, 64
    g = torch.Generator().manual_seed(99)
    mk = lambda: torch.randn(B, T, H, D, generator=g).to(dev).permute(0, 2, 1, 3)
    q, k, v, d_o = mk(), mk(), mk(), mk()
    scale = D ** -0.5
    core = 4.0 * B * H * T * T * D
    rows = []
    t = time_launches(lambda: ops._attn_forward(q, k, v, None, None, scale), a.iters)
    rows.append(("attn_fwd", t, core))
    t = time_launches(lambda: ops._attn_forward(q, k, v, None, None, scale, keep_scores=True), a.iters)
    rows.append(("attn_fwd_keep", t, core))
    q, k, v, o, stats, scores = ops._attn_forward(q, k, v, None, None, scale, keep_scores=True)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    delta = torch.empty(B, H, T, device=dev)
    ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=1, delta=delta)
    for keys, bit in ((128, 16), (256, 32)):
        for kept in (False, True):
            fn = lambda: ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=8 | bit,
                                            delta=delta, scores=scores if kept else None)
            t = time_launches(fn, a.iters)
            rows.append((f"attn_bwd_fused keys={keys} {'kept' if kept else 'recompute'}", t, 2 * core))
    big = torch.empty(L_ws(B, H, T, T), device=dev)
    for kept in (False, True):
        fn = lambda: ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=8 | 64 | 32,
                                        delta=big, scores=scores if kept else None)
        ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=1, delta=big)
        t = time_launches(fn, a.iters)
        rows.append((f"attn_bwd_fused keys=256 {'kept' if kept else 'recompute'} reproducible dq", t, 2 * core))
    for name, t, fl in rows:
        print(f"{name:44s} {t * 1e3:8.4f} ms  {fl / t / 1e12:7.2f} TFLOP/s  {fl / t / 1e12 / PEAK:.3f} of f32 MFMA peak", flush=True)
