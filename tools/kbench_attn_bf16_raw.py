"""The bf16 attention forward through the C ABI alone (the autograd wrapper's host time, ~80 us per call, would hide a kernel
of that length), a fixed number of launches; also what the rocprofv3 --pmc passes run:
    OUT=r4_pmc_bf16 SCRIPT=tools/kbench_attn_bf16_raw.py ARGS="--iters 10" FILTER="attn_bf16" bash tools/pmc_dense.sh
    python tools/kbench_attn_bf16_raw.py [--iters 50] [--batch 32] [--tokens 1024]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def fwd_raw(q2, kv2, o2, stats, H):
    from amk import lib as _lib
    B, I, _ = q2.shape
    J, D = kv2.shape[1], 64
    qs, kvs = (I * H * D, H * D, D), (J * 2 * H * D, 2 * H * D, D)
    rc = _lib.load().amk_attn_bf16_fwd(ctypes.c_void_p(q2.data_ptr()), ctypes.c_void_p(kv2.data_ptr()), ctypes.c_void_p(kv2.data_ptr() + 2 * H * D),
                                       ctypes.c_void_p(o2.data_ptr()), ctypes.c_void_p(stats.data_ptr()), None, None,
                                       B, H, I, J, D, *qs, *kvs, *kvs, *qs, float(D ** -0.5), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0


def operands(B, T, H=8, D=64, dev="cuda:0"):
    g = torch.Generator().manual_seed(0)
    q2 = torch.randn(B, T, H * D, generator=g).to(dev).bfloat16()
    kv2 = torch.randn(B, T, 2 * H * D, generator=g).to(dev).bfloat16()
    return q2, kv2, torch.empty_like(q2), torch.empty(B, H, T, 2, device=dev)


if __name__ == "__main__":
    from tools.kbench_moe import time_launches
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    q2, kv2, o2, stats = operands(a.batch, a.tokens)
    t = time_launches(lambda: fwd_raw(q2, kv2, o2, stats, 8), a.iters)
    fl = 4.0 * a.batch * 8 * a.tokens * a.tokens * 64
    print(f"amk_attn_bf16_fwd B{a.batch} T{a.tokens}: {t*1e6:.1f} us  {fl/t/1e12:.0f} TFLOP/s  {fl/t/1e12/2500:.3f} of the bf16 MFMA peak")
