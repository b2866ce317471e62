"""ViT-VQGAN train step (configs[2]) at small batches: own f32 Linear GEMMs (AMK_DENSE=auto / amk) against the library
(AMK_DENSE=lib) -- where the row count M = batch * 1024 stops favouring the own kernels."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch
import bench
from amk import tuning
from amk.models import ViTVQGAN
from amk.models.discriminator import NLayerDiscriminator
from amk.train import VQGANTrainStep
tuning.enable_conv_autotune(True); tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
for B in [int(b) for b in (sys.argv[1:] or ["4", "8", "16"])]:
    torch.manual_seed(0)
    model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev)
    discr = NLayerDiscriminator(3, 64, 3).to(dev)
    tr = VQGANTrainStep(model, discr)
    img = torch.rand(B, 3, 256, 256, device=dev)
    for _ in range(4): tr.step(img)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): tr.step(img)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"AMK_DENSE={os.environ.get('AMK_DENSE','auto')} batch {B}: {dt*1e3:.1f} ms = {B/dt:.0f} images/s", flush=True)
    del tr, model, discr
    torch.cuda.empty_cache()
