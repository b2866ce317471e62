"""torch.profiler view of MUSE.generate (configs[4] size, batch 8): decoder passes vs sampling ops."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

from amk import tuning  # noqa: E402
from amk.models import MUSE, ViTVQGAN  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
tuning.enable_gemm_tuning(results_csv=None)
vq = ViTVQGAN(dict(dim=256, img_size=256, patch_size=8, n_heads=8, d_head=64, depth=6, mlp_dim=2048, dropout=0.0),
              dict(codebook_size=8192, codebook_dim=32))
muse = MUSE(dim=1024, vq=vq, n_heads=16, d_head=64, depth=22, mult=6).to(dev)
text = torch.randn(8, 77, 768, device=dev)
muse.generate(text, timesteps=4)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    muse.generate(text, timesteps=6)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=32, max_name_column_width=56))
