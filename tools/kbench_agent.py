"""AgentAttention module timings (B 2 = the BASELINE.md row, B 64 = the chip-filling case).

    python tools/kbench_agent.py [--iters 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
sys.path.insert(0, ROOT)
from tools.kbench_moe import time_launches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from amk import ops
    from amk.models import AgentAttention

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ag = AgentAttention(384, 6, 64).to(dev)
    for B in (2, 8, 64):
        x = torch.randn(B, 1024, 384, device=dev, requires_grad=True)
        cot = torch.randn(B, 1024, 384, device=dev)
        qkv = ag.qkv(x).detach().requires_grad_(True)
        cw, cb = ag.dwc[1].weight, ag.dwc[1].bias
        core = lambda: ops.agent_attention(qkv, cw, cb, 6, 64, ag.pool_size, ag.scale)
        co = torch.randn(B, 1024, 384, device=dev)

        def core_fb():
            core().backward(co)

        def fb():
            ag(x).backward(cot)
        t_c = time_launches(core, a.iters)
        t_cfb = time_launches(core_fb, a.iters)
        t_f = time_launches(lambda: ag(x), a.iters)
        t_fb = time_launches(fb, a.iters)
        byt_f = 4.0 * B * 1024 * 384 * 4   # q, k, v read + o written
        print(f"B {B:3d}: core fwd {t_c*1e3:7.3f} ms ({byt_f/t_c/1e9:7.1f} GB/s algorithmic)  core fwd+bwd {t_cfb*1e3:7.3f} ms"
              f" | module fwd {t_f*1e3:7.3f} ms  fwd+bwd {t_fb*1e3:7.3f} ms")


if __name__ == "__main__":
    main()
