"""Residual+LayerNorm and column-sum kernels against torch's own, at the ViT-VQGAN layer shape.

    python tools/kbench_ln.py [--rows 32768] [--width 256]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
sys.path.insert(0, ROOT)
from tools.kbench_moe import time_launches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32768)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from amk import lib, ops

    L = lib.load()
    dev = torch.device("cuda:0")
    M, D = a.rows, a.width
    x, r, dy, dh_in = (torch.randn(M, D, device=dev) for _ in range(4))
    w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    h, y = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dh = torch.empty_like(x)
    part = torch.empty(L.amk_rowsum_num_partials(M), 2, D, device=dev)
    P, st = ops._ptr, ops._stream()
    mb = M * D * 4 / 1e6
    rows = [
        ("add_layernorm_fwd (x, res -> h, y)", 4 * mb,
         lambda: L.amk_add_layernorm_fwd(P(x), P(r), P(w), P(b), M, D, 1e-5, P(h), P(y), P(mean), P(rstd), st)),
        ("add_layernorm_bwd (dy, h, dh_in -> dh)", 4 * mb,
         lambda: L.amk_add_layernorm_bwd(P(dy), P(h), P(dh_in), P(w), P(mean), P(rstd), M, D, P(dh), P(part), st)),
        ("colsum", mb, lambda: L.amk_colsum(P(x), M, D, P(part), st)),
        ("torch: add + layer_norm", 4 * mb, lambda: F.layer_norm(x + r, (D,), w, b)),
        ("torch: sum(0)", mb, lambda: x.sum(0)),
    ]
    L.amk_add_layernorm_fwd(P(x), P(r), P(w), P(b), M, D, 1e-5, P(h), P(y), P(mean), P(rstd), st)
    for name, mbytes, fn in rows:
        t = time_launches(fn, a.iters)
        print(f"{name:42s} {t*1e6:8.1f} us  {mbytes/1e6/t:6.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
