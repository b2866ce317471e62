"""A/B inside ONE process: the unmasked attention forward on attn_fwd_plain_kernel (lazy reference) against the mask-capable
attn_fwd_kernel (AMK_ATTN_FWD_PLAIN is read per call), interleaved, with and without kept scores.  Separate processes are
not comparable at the 2 % level on these boxes (the first seconds of a process run at another clock).
    python tools/ab_attn_fwd.py [--batch 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tools.kbench_moe import time_launches  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
from amk import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, T, D = a.batch, 8, 1024, 64
q, k, v = (torch.randn(B, T, H, D, device=dev).permute(0, 2, 1, 3) for _ in range(3))
s = D ** -0.5
fl = 4.0 * B * H * T * T * D
for rnd in range(4):
    for plain in ("1", "0"):
        os.environ["AMK_ATTN_FWD_PLAIN"] = plain
        for keep in (False, True):
            t = time_launches(lambda: ops._attn_forward(q, k, v, None, None, s, keep_scores=keep), a.iters)
            print(f"round {rnd} {'plain kernel ' if plain == '1' else 'masked-capable'} keep={keep!s:5}: {t*1e3:.4f} ms  {fl/t/1e12/157.3:.3f} of peak")
