"""torch.profiler view of one ViT-VQGAN train step under bf16 autocast (the variant_bf16_autocast path)."""
import os
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
tr = VQGANTrainStep(model, discr, autocast=torch.bfloat16)
imgs = torch.rand(32, 3, 256, 256, device=dev)
for _ in range(3):
    tr.step(imgs)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        tr.step(imgs)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=60))
