"""Soak: the configs[2] model for a few dozen steps in f32 / bf16 / bf16 + graph / accumulated: finite losses, l2 going down."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch
import bench
from amk import tuning
from amk.models import ViTVQGAN
from amk.models.discriminator import NLayerDiscriminator
from amk.train import VQGANTrainStep

tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
B = 8
g = torch.Generator().manual_seed(1)
imgs = torch.nn.functional.interpolate(torch.rand(B, 3, 16, 16, generator=g), size=256, mode="bilinear").to(dev)
for mode in sys.argv[1:] or ["f32", "bf16", "bf16graph", "bf16accum"]:
    torch.manual_seed(0)
    model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev)
    discr = NLayerDiscriminator(3, 64, 3).to(dev)
    amp = torch.bfloat16 if mode.startswith("bf16") else None
    tr = VQGANTrainStep(model, discr, lr=3e-4, warmup_steps=5, decay_steps=400, autocast=amp, capturable="graph" in mode)
    if "graph" in mode:
        for _ in range(2):
            tr.step(imgs)
        tr.capture(imgs, warmup=0)
    hist = []
    for step in range(40):
        logs = tr.step_accumulated([imgs[:4], imgs[4:]]) if "accum" in mode else tr.step(imgs)
        vals = {k: float(v) for k, v in logs.items()}
        assert all(v == v and abs(v) < 1e30 for v in vals.values()), (mode, step, vals)
        hist.append(vals["l2"])
    ok = all(bool(torch.isfinite(p).all()) for p in list(model.parameters()) + list(discr.parameters()))
    print(f"{mode}: l2 {hist[0]:.4f} -> {hist[-1]:.4f}, d_loss {vals['d_loss']:.3f}, g_loss {vals['g_loss']:.3f}, parameters finite: {ok}", flush=True)
    del model, discr, tr
    torch.cuda.empty_cache()
