"""cProfile of the host side of the ViT-VQGAN train step (where do the ~97 ms of launch work per step go?)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from amk import tuning  # noqa: E402
from amk.models import ViTVQGAN  # noqa: E402
from amk.models.discriminator import NLayerDiscriminator  # noqa: E402
from amk.train import VQGANTrainStep  # noqa: E402

tuning.enable_conv_autotune(True)
tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev)
discr = NLayerDiscriminator(3, 64, 3).to(dev)
tr = VQGANTrainStep(model, discr)
imgs = torch.rand(32, 3, 256, 256, device=dev)
for _ in range(3):
    tr.step(imgs)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(imgs)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
