"""ViT-VQGAN generator forward + backward (bench configuration, batch 32): f32 against torch.autocast(bf16) -- the
reference's shipped mixed-precision setting (cfg/vitvqgan.yaml:73).  The attention / VQ / LayerNorm / gate kernels
stay f32 under autocast (amk.ops upcasts their inputs); the nn.Linear GEMMs run in bf16.
    python tools/kbench_autocast.py [--batch 32] [--step]
--step: the whole GAN train step of bench.py with both phases' forwards under bf16 autocast, as the reference's
accelerator.autocast() blocks run them (about a minute of MIOpen searches for the bf16 convolutions first, which is why
this is a tool and not a variant of the default bench run).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from bench import time_launches  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--step", action="store_true")
    a = ap.parse_args()
    from amk import tuning
    from amk.models import ViTVQGAN

    tuning.enable_gemm_tuning()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev)
    img = torch.rand(a.batch, 3, 256, 256, device=dev)

    def step(amp):
        def run():
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                rec, loss = model(img)
                total = (rec.float() - img).abs().mean() + loss
            total.backward()
            return total
        return run

    t32 = time_launches(step(False), a.iters)
    t16 = time_launches(step(True), a.iters)
    l32, l16 = float(step(False)()), float(step(True)())
    if a.step:
        import time

        from amk.models.discriminator import NLayerDiscriminator
        from amk.train import VQGANTrainStep

        tuning.enable_conv_autotune(True)
        tr = VQGANTrainStep(model, NLayerDiscriminator(3, 64, 3).to(dev), autocast=torch.bfloat16)
        t0 = time.time()
        for _ in range(3):
            logs = tr.step(img)
        torch.cuda.synchronize()
        print(f"train step under bf16 autocast: warm-up {time.time() - t0:.0f} s, losses "
              + ", ".join(f"{k} {float(v):.4f}" for k, v in logs.items()), flush=True)
        t0 = time.time()
        for _ in range(a.iters):
            tr.step(img)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.iters
        print(f"train step under bf16 autocast, batch {a.batch}: {dt * 1e3:.1f} ms = {a.batch / dt:.0f} images/s")
    print(f"generator fwd+bwd, batch {a.batch}: f32 {t32*1e3:.1f} ms ({a.batch/t32:.0f} images/s), bf16 autocast {t16*1e3:.1f} ms "
          f"({a.batch/t16:.0f} images/s); loss {l32:.5f} vs {l16:.5f}")
