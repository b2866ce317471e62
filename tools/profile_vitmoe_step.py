"""ViTMoE (BASELINE.json configs[3]) forward + cross-entropy + backward at batch 64, a few steps: the workload
`rocprofv3 --kernel-trace --stats` is pointed at to see where a step's time goes.
    python tools/profile_vitmoe_step.py [--steps 5] [--graph]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--events", action="store_true", help="HIP events around the routed-expert launches: per-launch table")
    ap.add_argument("--no-tune", action="store_true", help="library GEMMs without TunableOp (keeps its trial kernels out of a profile)")
    a = ap.parse_args()
    from amk import tuning
    from amk.models import ViTMoE

    if not a.no_tune:
        tuning.enable_gemm_tuning()   # as bench.py: TunableOp's selection for the GEMMs left on the library
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vm = ViTMoE(dim=1024, image_size=256, patch_size=32, n_heads=8, d_head=64, depth=6, n_experts=32, sel_experts=2,
                dropout=0.0, num_classes=1000).to(dev)
    imgs = torch.randn(a.batch, 3, 256, 256, device=dev)
    labels = torch.randint(0, 1000, (a.batch,), device=dev)

    def step():
        vm.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(vm(imgs), labels).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"ViTMoE batch {a.batch}: {dt*1e3:.2f} ms per step (host enqueue {t_host/a.steps*1e3:.2f} ms)")
    if a.events:
        from amk import ops

        ops.KERNEL_EVENTS = {}
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ev = ops.kernel_event_summary(ops.KERNEL_EVENTS)
        ops.KERNEL_EVENTS = None
        for name, (n, ms) in sorted(ev.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            print(f"  {name:60s} {n / a.steps:5.1f} per step  {ms * 1e3:8.1f} us  {ms * n / a.steps:7.3f} ms per step")


if __name__ == "__main__":
    main()
