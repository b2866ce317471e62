"""BASELINE.json configs[4] at its own size: Muse decoder D 1024, h 16, d 64, depth 22, mult 6,
vocab 8192, 1024 image tokens, 77 text positions; training forward+backward and the 18-step
parallel decode (2 decoder passes per step).  Synthetic text states (the CLIP tower needs a download).
    python tools/kbench_muse.py [--batch 8]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

from amk import tuning  # noqa: E402
from amk.models import MUSE, ViTVQGAN  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
tuning.enable_gemm_tuning(results_csv=None)
vq = ViTVQGAN(dict(dim=256, img_size=256, patch_size=8, n_heads=8, d_head=64, depth=6, mlp_dim=2048, dropout=0.0),
              dict(codebook_size=8192, codebook_dim=32))
muse = MUSE(dim=1024, vq=vq, n_heads=16, d_head=64, depth=22, mult=6).to(dev)
print(f"decoder parameters: {sum(p.numel() for p in muse.decoder.parameters())/1e6:.1f} M")
B = a.batch
text = torch.randn(B, 77, 768, device=dev)
imgs = torch.rand(B, 3, 256, 256, device=dev)
opt = torch.optim.AdamW([p for p in muse.parameters() if p.requires_grad], lr=1e-4, fused=True)


def train_step():
    loss = muse(text, imgs)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


for _ in range(3):
    train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    train_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"train step (encode_imgs + decoder fwd+bwd + AdamW), batch {B}: {dt*1e3:.1f} ms = {B/dt:.1f} images/s")
muse.generate(text, timesteps=18)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = muse.generate(text, timesteps=18)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"generate 18 steps x 2 passes, batch {B}: {dt*1e3:.1f} ms = {B/dt:.2f} images/s, output {tuple(out.shape)}")
