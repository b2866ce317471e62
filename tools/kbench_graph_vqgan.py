"""Eager vs HIP-graph replay of the ViT-VQGAN GAN train step (batch 32): is the step CPU-launch-bound?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from amk import tuning  # noqa: E402
from amk.graphs import GraphedStep  # noqa: E402
from amk.models import ViTVQGAN  # noqa: E402
from amk.models.discriminator import NLayerDiscriminator  # noqa: E402
from amk.train import VQGANTrainStep  # noqa: E402

tuning.enable_conv_autotune(True)
tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev)
discr = NLayerDiscriminator(3, 64, 3).to(dev)
tr = VQGANTrainStep(model, discr, capturable=True)
imgs = torch.rand(B, 3, 256, 256, device=dev)


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n, (t1 - t0) / n


t_eager, q_eager = timeit(lambda: tr.step(imgs))
g = GraphedStep(lambda x: tr.step_body(x), [imgs])
t_graph, q_graph = timeit(lambda: g.replay(imgs))
print(f"ViT-VQGAN train step, batch {B}: eager {t_eager*1e3:.1f} ms (CPU enqueue {q_eager*1e3:.1f} ms), "
      f"HIP-graph replay {t_graph*1e3:.1f} ms (CPU {q_graph*1e3:.2f} ms)")
