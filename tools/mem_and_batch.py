import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch, bench
from amk import tuning
from amk.models import ViTVQGAN
from amk.models.discriminator import NLayerDiscriminator
from amk.train import VQGANTrainStep
tuning.enable_conv_autotune(True); tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
for B in (32, 64):
    torch.manual_seed(0)
    model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev); discr = NLayerDiscriminator(3, 64, 3).to(dev)
    tr = VQGANTrainStep(model, discr)
    base = torch.cuda.memory_allocated()
    imgs = torch.rand(B, 3, 256, 256, device=dev)
    for _ in range(3): tr.step(imgs)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(8): tr.step(imgs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"B {B}: {B/dt:.1f} images/s ({dt*1e3:.1f} ms/step), params+optimizer+grads resident {base/2**30:.2f} GiB, peak allocated {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
    del tr, model, discr, imgs
    torch.cuda.empty_cache()
