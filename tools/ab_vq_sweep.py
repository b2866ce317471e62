"""A/B inside ONE process: the VQ lookup with the argmin sweep's variants (AMK_VQ_VAR, read per call): 1 = running maximum folded
into the group's max tree, 2 = sub-tiles software-pipelined past the MFMA-result hazard, 3 = both.  Indices must be identical.
    python tools/ab_vq_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tools.kbench_moe import time_launches  # noqa: E402
from amk import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, K, C = 32 * 1024, 8192, 32
z = torch.randn(N, C, device=dev)
cb = torch.randn(K, C, device=dev)
ref = None
for var in ("0", "3", "5", "7"):
    os.environ["AMK_VQ_VAR"] = var
    out = ops.vq_lookup(z, cb, 0.25)
    idx = [t for t in out if t.dtype == torch.int64][0] if isinstance(out, (tuple, list)) else out
    if ref is None:
        ref = idx.clone()
    print(f"var {var}: indices differing from var 0: {(idx != ref).sum().item()} of {idx.numel()}")
for rnd in range(4):
    for var in ("0", "3", "5", "7"):
        os.environ["AMK_VQ_VAR"] = var
        t = time_launches(lambda: ops.vq_lookup(z, cb, 0.25), 50)
        print(f"round {rnd} var {var}: vq_lookup {t*1e6:.1f} us  {2.0*N*K*C/t/1e12/157.3:.3f} of the f32 MFMA peak")
