"""The nn.Linear GEMM shapes of the ViT-VQGAN step (batch 32: M = 32768 rows): vendor exact-f32 GEMM
(TunableOp selection) against the split-bf16 kernel (amk_gemm_x6_nt), forward shapes and input-gradient shapes.
    python tools/kbench_gemm.py [--batch 32] [--iters 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))

import torch  # noqa: E402

from bench import time_launches  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from amk import ops, tuning

    tuning.enable_gemm_tuning()
    dev = torch.device("cuda:0")
    M = a.batch * 1024
    g = torch.Generator().manual_seed(0)
    shapes = [("q / pre-LN proj", 256, 512), ("kv", 256, 1024), ("W_o", 512, 256), ("ffn w12", 256, 2736),
              ("ffn w3", 1368, 256), ("patch embed", 192, 256), ("pre_quant", 256, 32), ("post_quant", 32, 256),
              ("dX of kv", 1024, 256), ("dX of w12", 2736, 256), ("dX of w3", 256, 1368)]
    tot_lib = tot_x6 = 0.0
    for name, K, N in shapes:
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        t_lib = time_launches(lambda: torch.nn.functional.linear(x, w, b), a.iters)
        t_x6 = time_launches(lambda: ops.gemm_x6_nt(x, w, b), a.iters)
        ref = x.double() @ w.double().t() + b.double()
        e_lib = float((torch.nn.functional.linear(x, w, b).double() - ref).abs().max() / ref.abs().max())
        e_x6 = float((ops.gemm_x6_nt(x, w, b).double() - ref).abs().max() / ref.abs().max())
        fl = 2.0 * M * N * K
        tot_lib += t_lib
        tot_x6 += t_x6
        print(f"{name:16s} M{M} K{K:5d} N{N:5d}  library f32 {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.1f} TF/s err {e_lib:.1e} | "
              f"bf16x6 {t_x6*1e6:7.1f} us {fl/t_x6/1e12:6.1f} TF/s err {e_x6:.1e}  x{t_lib/t_x6:.2f}", flush=True)
    print(f"sum: library {tot_lib*1e3:.3f} ms, bf16x6 {tot_x6*1e3:.3f} ms")
