"""Eager vs HIP-graph replay of a ViTMoE training step at batch 2 (launch-bound regime)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

from amk.graphs import GraphedStep  # noqa: E402
from amk.models import ViTMoE  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = ViTMoE(dim=1024, patch_size=32, n_heads=8, depth=6, n_experts=32, sel_experts=2).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True, foreach=True)
imgs = torch.randn(B, 3, 256, 256, device=dev)
labels = torch.randint(0, 1000, (B,), device=dev)


def step(x, y):
    loss = torch.nn.functional.cross_entropy(model(x), y)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=False)
    return loss


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_eager = timeit(lambda: step(imgs, labels))
g = GraphedStep(step, [imgs, labels])
t_graph = timeit(lambda: g.replay(imgs, labels))
l_eager = float(step(imgs, labels))
l_graph = float(g.replay(imgs, labels))
print(f"ViTMoE train step, batch {B}: eager {t_eager*1e3:.2f} ms, HIP-graph replay {t_graph*1e3:.2f} ms "
      f"({t_eager/t_graph:.2f}x); loss eager {l_eager:.4f} graph {l_graph:.4f}")
