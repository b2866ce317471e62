"""Soak: ViT, ViTMoE and the Muse decoder train for a few steps in f32 and under bf16 autocast: finite, loss going down."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch
import torch.nn.functional as F
from amk.models import ViT, ViTMoE

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
imgs = torch.randn(16, 3, 64, 64, generator=g).to(dev)
labels = torch.randint(0, 10, (16,), generator=g).to(dev)
cases = {
    "ViT": lambda: ViT(dim=128, image_size=64, patch_size=8, n_heads=2, d_head=64, depth=2, mlp_dim=256, num_classes=16),
    "ViTMoE": lambda: ViTMoE(dim=128, image_size=64, patch_size=8, n_heads=2, d_head=64, depth=2, n_experts=4, sel_experts=2, num_classes=16),
}
for name, make in cases.items():
    for amp in (None, torch.bfloat16):
        torch.manual_seed(0)
        m = make().to(dev)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
        hist = []
        for step in range(30):
            with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                loss = F.cross_entropy(m(imgs).float(), labels)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            hist.append(float(loss))
            assert hist[-1] == hist[-1], (name, amp, step)
        ok = all(bool(torch.isfinite(p).all()) for p in m.parameters())
        print(f"{name} {'bf16' if amp else 'f32'}: loss {hist[0]:.3f} -> {hist[-1]:.3f}, parameters finite: {ok}", flush=True)
