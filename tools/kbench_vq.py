"""VQ lookup forward (prep + argmin + finalize) timed over the codebook-slice count.

    python tools/kbench_vq.py [--batch 32] [--iters 30]
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
    a = ap.parse_args()
    from amk import ops

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, K, C = a.batch * 1024, 8192, 32
    z = torch.randn(a.batch, 1024, C, device=dev)
    cb = torch.randn(K, C, device=dev)
    default = ops.vq_nsplit(N, K)
    ref = None
    for ns in (1, 2, 4, 8, 16):
        ops.vq_nsplit = lambda n, k, ns=ns: ns
        out, idx, loss = ops.vq_lookup(z, cb, 0.25)
        if ref is None:
            ref = idx
        same = bool(torch.equal(ref, idx))
        t = time_launches(lambda: ops.vq_lookup(z, cb, 0.25), a.iters)
        fl = 2.0 * N * K * C
        print(f"nsplit {ns:2d}{' (default)' if ns == default else '':10s} {t*1e3:7.4f} ms  {fl/t/1e12:6.1f} TFLOP/s  "
              f"{fl/t/157.3e12:.3f} of peak  idx equal to nsplit 1: {same}")


if __name__ == "__main__":
    main()
