"""Kernel-only micro-benchmark: the per-kernel roofline table of bench.py without the train step.
    python tools/kbench.py [--batch 32] [--iters 30]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))

import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for r in bench.kernel_rooflines(a.batch, dev, a.iters):
        print(f"{r['kernel']:40s} {r['avg_ms']:8.4f} ms  {r['tflops']:7.2f} TFLOP/s  {r['frac_of_f32_mfma_peak']:.3f} of peak")
