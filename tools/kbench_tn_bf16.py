"""Times amk_gemm_tn_bf16 against the vendor GEMM (+ the separate bias reduction) on the weight-gradient shapes of
the configs[2] ViT blocks.  python tools/kbench_tn_bf16.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "attention-models_amd"))
from amk import dense  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    M = a.batch * 1024
    dev = torch.device("cuda:0")
    for name, N, K, bias in [("dW12", 2736, 256, True), ("dW3", 256, 1368, True), ("dWkv", 1024, 256, False),
                             ("dWq", 512, 256, False), ("dWo", 256, 512, True)]:
        y = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        t_own = timeit(lambda: dense.gemm_tn_bf16(y, x, want_bias=bias))
        def lib():
            dw = y.t() @ x
            if bias:
                y.sum(0)
            return dw
        t_lib = timeit(lib)
        flops = 2.0 * M * N * K
        byts = (M * N + M * K) * 2 + N * K * 4
        print(f"{name:5s} N={N:5d} K={K:5d}: own {t_own:7.1f} us ({flops / t_own / 1e6:6.0f} TFLOP/s, {byts / t_own / 1e6:5.2f} TB/s)"
              f"   library {t_lib:7.1f} us ({flops / t_lib / 1e6:6.0f} TFLOP/s)", flush=True)


def forward_shapes(batch):
    """NT (forward, bias) and NN (input gradient) products of one ViT block against F.linear / mm in bf16."""
    import torch.nn.functional as F

    M = batch * 1024
    dev = torch.device("cuda:0")
    for name, N, K, bias in [("q", 512, 256, False), ("kv", 1024, 256, False), ("W_o", 256, 512, True), ("w12", 2736, 256, True),
                             ("w3", 256, 1368, True)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        b = torch.randn(N, device=dev) if bias else None
        b16 = b.bfloat16() if bias else None
        dy = torch.randn(M, N, device=dev).bfloat16()
        flops = 2.0 * M * N * K
        t_own, t_lib = timeit(lambda: dense.gemm_nt_bf16(a, w, b)), timeit(lambda: F.linear(a, w, b16))
        byts = (M * K + M * N + N * K) * 2
        print(f"NT {name:4s} N={N:5d} K={K:5d}: own {t_own:7.1f} us ({flops / t_own / 1e6:6.0f} TFLOP/s, {byts / t_own / 1e6:5.2f} TB/s)"
              f"   library {t_lib:7.1f} us ({flops / t_lib / 1e6:6.0f} TFLOP/s)", flush=True)
        t_own, t_lib = timeit(lambda: dense.gemm_nn_bf16(dy, w)), timeit(lambda: dy @ w)
        print(f"NN {name:4s} N={K:5d} K={N:5d}: own {t_own:7.1f} us ({flops / t_own / 1e6:6.0f} TFLOP/s, {byts / t_own / 1e6:5.2f} TB/s)"
              f"   library {t_lib:7.1f} us ({flops / t_lib / 1e6:6.0f} TFLOP/s)", flush=True)
    H, K = 1368, 256
    a = torch.randn(M, K, device=dev).bfloat16()
    w12 = (torch.randn(2 * H, K, device=dev) * K ** -0.5).bfloat16()
    b12 = torch.randn(2 * H, device=dev)
    from amk import ops
    t_own = timeit(lambda: dense.gemm_nt_swiglu_bf16(a, w12, b12))
    t_own2 = timeit(lambda: dense.gemm_nt_swiglu_bf16(a, w12, b12, keep_ab=False))
    b16 = b12.bfloat16()
    t_lib = timeit(lambda: ops.swiglu(F.linear(a, w12, b16)))
    print(f"w12 + SwiGLU: own {t_own:.1f} us (gate only: {t_own2:.1f} us)   library GEMM + amk_swiglu_bf16 {t_lib:.1f} us", flush=True)


if __name__ == "__main__":
    main()
    forward_shapes(32)
