"""Micro-benchmark of the routed-expert kernels at the ViTMoE layer shapes (BASELINE.json configs[3]):
MoELayer(D 1024, E 32, top-2) and SwitchHeadAttention(D 1024, h 8, E 32), T = 65 tokens.
    python tools/kbench_moe.py [--batch 64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))

import torch  # noqa: E402

from bench import time_launches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from amk import ops
    from amk.models import AgentAttention, MoELayer, SwitchHeadAttention, ViTMoE

    dev = torch.device("cuda:0")
    B, T, D, E, k, h = a.batch, 65, 1024, 32, 2, 8
    torch.manual_seed(0)
    x = torch.randn(B, T, D, device=dev, requires_grad=True)
    cot = torch.randn(B, T, D, device=dev)

    def fb(m):
        def run():
            out = m(x)
            out.backward(cot)
            return out
        return run

    R = B * T
    moe = MoELayer(D, D, E, k).to(dev)
    t_f = time_launches(lambda: moe(x), a.iters)
    t_fb = time_launches(fb(moe), a.iters)
    fl = 2.0 * k * R * D * D
    print(f"MoELayer       fwd {t_f*1e3:8.3f} ms ({fl/t_f/1e12:6.1f} TFLOP/s algorithmic)  fwd+bwd {t_fb*1e3:8.3f} ms ({3*fl/t_fb/1e12:6.1f} TFLOP/s)  "
          f"weights {E*D*D*4/1e6:.0f} MB -> fwd weight-stream floor {E*D*D*4/8e12*1e3:.3f} ms")
    # stage breakdown of the forward
    logits = moe.gate(x).reshape(R, E)
    x2 = x.detach().reshape(R, D)
    t_route = time_launches(lambda: ops.moe_route(logits.detach(), k), a.iters)
    print(f"   route (3 launches) {t_route*1e3:.3f} ms")
    # individual ABI calls of the MoE layer
    import ctypes
    from amk import lib as L_
    L = L_.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    r = ops.moe_route(logits.detach(), k)
    W = moe.experts_weight.detach(); bias = moe.experts_bias.detach()
    P = R * k
    Y = torch.empty(P, D, device=dev); dxp = torch.empty(P, D, device=dev); dW = torch.empty_like(W); db = torch.empty_like(bias)
    d_out = cot.reshape(R, D).contiguous()
    calls = {
        "grouped_nt": lambda: L.amk_grouped_gemm_nt(P_(x2), D, k, P_(W), P_(bias), P_(r["offsets"]), P_(r["perm"]), P, E, D, D, P_(Y), st),
        "grouped_nn": lambda: L.amk_grouped_gemm_nn(P_(d_out), D, k, P_(W), P_(r["gate"]), P_(r["offsets"]), P_(r["perm"]), P, E, D, D, P_(dxp), st),
        "grouped_wgrad": lambda: L.amk_grouped_gemm_wgrad(P_(d_out), D, k, P_(x2), D, k, P_(r["gate"]), P_(r["offsets"]), P_(r["perm"]), P, E, D, D, P_(dW), P_(db), st),
        "combine": lambda: L.amk_moe_combine(P_(Y), P_(r["ids"]), P_(r["gate"]), R, 1, k, D, P_(dxp), st),
    }
    for name, fn in calls.items():
        t = time_launches(fn, a.iters)
        print(f"   {name:14s} {t*1e3:8.3f} ms  {fl/t/1e12 if name.startswith('grouped') else 0:6.1f} TFLOP/s")
    sh = SwitchHeadAttention(D, h, 64, num_experts=E, sel_experts=k).to(dev)
    t_f = time_launches(lambda: sh(x), a.iters)
    t_fb = time_launches(fb(sh), a.iters)
    fl = 2.0 * k * R * h * D * 64 * 2 + 4.0 * B * h * T * T * 64
    print(f"SwitchHead     fwd {t_f*1e3:8.3f} ms ({fl/t_f/1e12:6.1f} TFLOP/s algorithmic)  fwd+bwd {t_fb*1e3:8.3f} ms")
    ag = AgentAttention(384, 6, 64).to(dev)
    xa = torch.randn(B, 1024, 384, device=dev, requires_grad=True)
    cota = torch.randn(B, 1024, 384, device=dev)
    t_f = time_launches(lambda: ag(xa), a.iters)

    def fba():
        ag(xa).backward(cota)
    t_fb = time_launches(fba, a.iters)
    print(f"AgentAttention (B {B}, T 1024, D 384, h 6) fwd {t_f*1e3:8.3f} ms  fwd+bwd {t_fb*1e3:8.3f} ms")
    vm = ViTMoE(dim=1024, patch_size=32, n_heads=8, depth=6, n_experts=32, sel_experts=2).to(dev)
    imgs = torch.randn(B, 3, 256, 256, device=dev)
    labels = torch.randint(0, 1000, (B,), device=dev)

    def step():
        loss = torch.nn.functional.cross_entropy(vm(imgs), labels)
        loss.backward()
    t = time_launches(step, 5, warm=2)
    print(f"ViTMoE (240.6 M params) fwd+bwd, batch {B}: {t*1e3:.1f} ms = {B/t:.1f} images/s")


if __name__ == "__main__":
    main()
