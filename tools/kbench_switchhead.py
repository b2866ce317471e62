"""SwitchHeadAttention (ViTMoE layer shape: D 1024, h 8, E 32, top-2, T 65) forward / forward+backward.

    python tools/kbench_switchhead.py [--batch 64] [--iters 20]
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from amk.models import SwitchHeadAttention

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sh = SwitchHeadAttention(1024, 8, 64, num_experts=32, sel_experts=2).to(dev)
    x = torch.randn(a.batch, 65, 1024, device=dev, requires_grad=True)
    cot = torch.randn(a.batch, 65, 1024, device=dev)

    def fb():
        sh(x).backward(cot)
    t_f = time_launches(lambda: sh(x), a.iters)
    t_fb = time_launches(fb, a.iters)
    print(f"SwitchHead (B {a.batch}, T 65, D 1024, h 8, E 32, k 2) fwd {t_f*1e3:7.3f} ms  fwd+bwd {t_fb*1e3:7.3f} ms")


if __name__ == "__main__":
    main()
