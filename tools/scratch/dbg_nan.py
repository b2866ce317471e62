import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch
from amk import ops
from oracle.fixture_recipe import seeded
dev = torch.device("cuda:0")
B, H, I, J, D = 2, 2, 200, 130, 64
mk = lambda seed, T: seeded((B, T, H, D), seed).to(dev).permute(0, 2, 1, 3)
q, k, v, d_o = mk(1, I), mk(2, J), mk(3, J), mk(4, I)
km = torch.ones(B, J, dtype=torch.uint8); km[0, 5::7] = 0; km = km.to(dev)
scale = D ** -0.5
q, k, v, o, stats, scores = ops._attn_forward(q, k, v, km, None, scale, keep_scores=True)
print("o nan", torch.isnan(o).sum().item(), "stats nan", torch.isnan(stats).sum().item(), "scores nan", torch.isnan(scores).sum().item())
for keys in (16, 32):
    for kept in (False, True):
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
        ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, km, None, scale, stages=9 | keys, scores=scores if kept else None)
        torch.cuda.synchronize()
        for name, t in (("dq", dq), ("dk", dk), ("dv", dv)):
            nan = torch.isnan(t)
            print(keys, kept, name, tuple(t.shape), t.stride(), "nan elems", nan.sum().item(), "rows", nan.any(-1).nonzero()[:8].tolist())
