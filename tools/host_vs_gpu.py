"""Is the train step host-bound?  Host time to ENQUEUE a step (no sync) against the GPU time it takes.
    python tools/host_vs_gpu.py [--gemm bf16x6]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from amk import ops, tuning  # noqa: E402
from amk.models import ViTVQGAN  # noqa: E402
from amk.models.discriminator import NLayerDiscriminator  # noqa: E402
from amk.train import VQGANTrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gemm", default="f32")
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
ops.GEMM_MODE = a.gemm
tuning.enable_conv_autotune(True)
tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev)
discr = NLayerDiscriminator(3, 64, 3).to(dev)
tr = VQGANTrainStep(model, discr)
imgs = torch.rand(32, 3, 256, 256, device=dev)
for _ in range(4):
    tr.step(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    tr.step(imgs)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"gemm {a.gemm}: host enqueue {t_host / a.steps * 1e3:.1f} ms per step, wall {t_all / a.steps * 1e3:.1f} ms per step "
      f"({'HOST-bound' if t_host > 0.95 * t_all else 'GPU-bound'})")
