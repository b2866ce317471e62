"""Sanity: over-fit a few synthetic images with the full GAN train step (the reference's own
'does the loss go down' check, cfg_exp/*.yaml) on the HIP path; prints losses every 20 steps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

from amk.models import ViTVQGAN  # noqa: E402
from amk.models.discriminator import NLayerDiscriminator  # noqa: E402
from amk.train import VQGANTrainStep  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = dict(dim=128, img_size=64, patch_size=8, n_heads=2, d_head=64, depth=2, mlp_dim=256, dropout=0.0)
model = ViTVQGAN(cfg, dict(codebook_size=512, codebook_dim=32)).to(dev)
discr = NLayerDiscriminator(3, 32, 3).to(dev)
# python tools/train_sanity.py [bf16] [graph]: the same under bf16 autocast / with the step replayed as one HIP graph
amp = torch.bfloat16 if "bf16" in sys.argv[1:] else None
graph = "graph" in sys.argv[1:]
tr = VQGANTrainStep(model, discr, lr=1e-3, warmup_steps=10, decay_steps=400, autocast=amp, capturable=graph)
g = torch.Generator().manual_seed(1)
base = torch.rand(8, 3, 8, 8, generator=g)
imgs = torch.nn.functional.interpolate(base, size=64, mode="bilinear").to(dev)  # smooth images
first = None
if graph:
    for _ in range(2):
        tr.step(imgs)
    tr.capture(imgs, warmup=0)
for step in range(301):
    logs = tr.step(imgs)
    if step % 50 == 0:
        vals = {k: round(float(v), 4) for k, v in logs.items()}
        print(step, vals, flush=True)
        if first is None:
            first = vals
assert all(torch.isfinite(p).all() for p in model.parameters())
assert vals["l2"] < 0.5 * first["l2"], "reconstruction loss did not go down"
print("ok: l2", first["l2"], "->", vals["l2"], " codes used:", int(model.encode_imgs(imgs).unique().numel()))
