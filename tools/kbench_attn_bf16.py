"""bf16-MFMA attention core (csrc/attn_bf16.hip) at the ViT-VQGAN layer shape: forward and backward launch times,
algorithmic TFLOP/s against the dense bf16 MFMA peak (2500 TFLOP/s), next to the exact-f32 kernels.
    python tools/kbench_attn_bf16.py [--batch 32] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

from bench import time_launches  # noqa: E402

BF16_PEAK = 2500.0

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-f32 kernels' line")
    a = ap.parse_args()
    from amk import ops

    dev = torch.device("cuda:0")
    B, H, T, D = a.batch, 8, 1024, 64
    g = torch.Generator().manual_seed(0)
    q2 = torch.randn(B, T, H * D, generator=g).to(dev).bfloat16().requires_grad_(True)
    kv2 = torch.randn(B, T, 2 * H * D, generator=g).to(dev).bfloat16().requires_grad_(True)
    cot = torch.randn(B, T, H * D, generator=g).to(dev).bfloat16()
    core = 4.0 * B * H * T * T * D
    o = ops.attention_fused_kv(q2, kv2, H, D, D ** -0.5)
    t_f = time_launches(lambda: ops.attention_fused_kv(q2, kv2, H, D, D ** -0.5), a.iters)
    t_fb = time_launches(lambda: torch.autograd.grad(ops.attention_fused_kv(q2, kv2, H, D, D ** -0.5), [q2, kv2], cot), a.iters)
    t_b = t_fb - t_f
    print(f"bf16 attention B{B} h{H} T{T} d{D}: forward {t_f*1e6:.1f} us = {core/t_f/1e12:.1f} TFLOP/s ({core/t_f/1e12/BF16_PEAK:.3f} of bf16 peak); "
          f"backward (delta + fused + dq reduce) {t_b*1e6:.1f} us = {2.5*core/t_b/1e12:.1f} TFLOP/s on its five products "
          f"({2.5*core/t_b/1e12/BF16_PEAK:.3f})")
    if a.no_f32:
        sys.exit(0)
    qf, kvf = q2.detach().float().requires_grad_(True), kv2.detach().float().requires_grad_(True)
    t_f32 = time_launches(lambda: ops.attention_fused_kv(qf, kvf, H, D, D ** -0.5), a.iters)
    t_fb32 = time_launches(lambda: torch.autograd.grad(ops.attention_fused_kv(qf, kvf, H, D, D ** -0.5), [qf, kvf], cot.float()), a.iters)
    print(f"exact-f32 kernels, same shape: forward {t_f32*1e6:.1f} us, backward {(t_fb32-t_f32)*1e6:.1f} us")
