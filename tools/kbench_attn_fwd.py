"""Attention forward at the ViT-VQGAN layer shape: exact-f32 MFMA kernel vs the split-bf16 (bf16x6) kernel,
time and error against a double-precision reference of one (batch, head).

    python tools/kbench_attn_fwd.py [--batch 32] [--iters 30]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
sys.path.insert(0, ROOT)
from tools.kbench_moe import time_launches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--f32-only", action="store_true")
    a = ap.parse_args()
    from amk import ops

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, H, T, D = a.batch, 8, 1024, 64
    q, k, v = (torch.randn(B, T, H, D, device=dev).permute(0, 2, 1, 3) for _ in range(3))
    s = D ** -0.5
    qd, kd, vd = (t[0, 0].double().cpu() for t in (q, k, v))
    ref = torch.softmax(qd @ kd.t() * s, -1) @ vd
    fl = 4.0 * B * H * T * T * D
    for mode in (("f32",) if a.f32_only else ("f32", "bf16x6")):
        ops.ATTENTION_FORWARD = mode
        o = ops.attention(q, k, v, s)
        err = float((o[0, 0].double().cpu() - ref).abs().max() / ref.abs().max())
        t = time_launches(lambda: ops.attention(q, k, v, s), a.iters)
        print(f"{mode:7s} {t*1e3:7.4f} ms  {fl/t/1e12:6.1f} algorithmic TFLOP/s  max rel err vs float64 {err:.2e}")
    ops.ATTENTION_FORWARD = "f32"


if __name__ == "__main__":
    main()
