"""The hand-written attention / VQ kernels of the ViT-VQGAN step, back to back, interleaved for several rounds inside ONE process:
the last rounds are the steady state of a warm chip.  (Single-shot timings of a fresh process, and timings taken after minutes of
full load, differ from these by up to 10 %: the chip's clock follows its power / thermal state.)
    python tools/kbench_steady.py [--batch 32] [--rounds 4]"""
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
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
from amk import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, T, D = a.batch, 8, 1024, 64
q, k, v, d_o = (torch.randn(B, T, H, D, device=dev).permute(0, 2, 1, 3) for _ in range(4))
s = D ** -0.5
core = 4.0 * B * H * T * T * D
q, k, v, o, stats, scores = ops._attn_forward(q, k, v, None, None, s, keep_scores=True)
dq, dk, dv = (torch.empty_like(q) for _ in range(3))
delta = torch.empty(lib.load().amk_attn_bwd_ws_floats(B, H, T, T, 72), device=dev)
bwd = lambda st, sc=None: ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, s, stages=st, delta=delta, scores=sc)
N, K, C = B * T, 8192, 32
z = torch.randn(N, C, device=dev)
cb = torch.randn(K, C, device=dev)
rows = [("attn_fwd (plain kernel)", lambda: ops._attn_forward(q, k, v, None, None, s), core),
        ("attn_fwd keeping scores", lambda: ops._attn_forward(q, k, v, None, None, s, keep_scores=True), core),
        ("attn_bwd fused, kept scores (+ dq memset)", lambda: bwd(8, scores), 2 * core),
        ("attn_bwd fused, recomputing", lambda: bwd(8), 2 * core),
        ("vq_lookup_fwd (prep + argmin + finalize)", lambda: ops.vq_lookup(z, cb, 0.25), 2.0 * N * K * C)]
for rnd in range(a.rounds):
    for name, fn, fl in rows:
        t = time_launches(fn, a.iters)
        print(f"round {rnd} {name:44s} {t*1e3:.4f} ms  {fl/t/1e12:6.1f} TFLOP/s  {fl/t/1e12/157.3:.3f} of the f32 MFMA peak")
