"""ViT-VQGAN inference throughput (configs[2] size): encode_imgs (encoder + VQ lookup -> codes) and
decode_indices (codes -> image), eager and as HIP-graph replays, fp32.

    python tools/kbench_infer.py [--batch 32]
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
from amk import tuning  # noqa: E402
from amk import ops  # noqa: E402
from amk.graphs import GraphedStep  # noqa: E402

ops.CHECK_INDICES = False  # the index check of decode_indices reads the device: not inside a captured graph
from amk.models import ViTVQGAN  # noqa: E402


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    tuning.enable_gemm_tuning()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(dev).eval()
    imgs = torch.rand(a.batch, 3, 256, 256, device=dev)
    with torch.no_grad():
        idx = model.encode_imgs(imgs)
        t_enc = timeit(lambda: model.encode_imgs(imgs))
        t_dec = timeit(lambda: model.decode_indices(idx))
        g_enc = GraphedStep(lambda x: model.encode_imgs(x), [imgs])
        g_dec = GraphedStep(lambda i: model.decode_indices(i), [idx])
        t_enc_g = timeit(lambda: g_enc.replay(imgs))
        t_dec_g = timeit(lambda: g_dec.replay(idx))
        same = bool(torch.equal(g_enc.replay(imgs), idx))
    B = a.batch
    print(f"batch {B}: encode_imgs {t_enc*1e3:.2f} ms = {B/t_enc:.0f} images/s (graph replay {t_enc_g*1e3:.2f} ms = {B/t_enc_g:.0f}); "
          f"decode_indices {t_dec*1e3:.2f} ms = {B/t_dec:.0f} images/s (graph replay {t_dec_g*1e3:.2f} ms = {B/t_dec_g:.0f}); "
          f"replayed codes equal eager codes: {same}")


if __name__ == "__main__":
    main()
