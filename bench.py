"""ViT-VQGAN 256px training-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...          (no launcher: this process starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; synthetic U[0,1) images seeded 1234+rank, random-init weights; a step is
one pass of the reference's train step (trainers/vitgqgan.py:139-206, per_loss_weight 0) over
one batch per rank: 2 generator forwards + 1 backward through the HIP attention / VQ kernels,
the PatchGAN discriminator + gradient penalty, two Adam updates, gradients all-reduced over
RCCL by amk.dp.GradReducer.  Rank 0 prints ONE JSON line.

Besides the step throughput the line carries
  roofline     the dominant HIP kernel at the layer shape of this workload, timed live with HIP
               events on the launch stream: algorithmic FLOP per launch / average duration
               against the 157.3 TFLOP/s exact-f32 MFMA peak (MI355X_MICROARCH.md)
  kernels      the same figure for every hand-written kernel of the step
  cpu_baseline the CPU oracle's train step on this box's host cores (bounded sample), N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "attention-models_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense
HBM_PEAK_GBS = 8000.0

VIT = dict(dim=256, img_size=256, patch_size=8, n_heads=8, d_head=64, depth=6, mlp_dim=2048, dropout=0.0)
CODEBOOK = dict(codebook_size=8192, codebook_dim=32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the informational shared-forward variant")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step launch by launch instead of replaying one HIP graph")
    ap.add_argument("--model", default="vqgan", choices=["vqgan", "vit", "vitmoe", "muse"],
                    help="vqgan (default): the headline, BASELINE.json configs[2].  vit / vitmoe / muse: a SECONDARY line -- the "
                         "data-parallel classifier step of configs[1] / configs[3] (trainers/vit.py) or the masked-token decoder step "
                         "of configs[4] (trainers/muse.py) through the same reducer, optimizer and graph capture")
    ap.add_argument("--autocast", default="none", choices=["none", "bf16"],
                    help="bf16: the reference's shipped precision (cfg/vitvqgan.yaml:73); makes the line a SECONDARY one (the headline is f32)")
    ap.add_argument("--dp-overlap", default="auto", choices=["auto", "on", "off"],
                    help="all-reduces on a side stream under backward (on) or on the compute stream (off).  auto: on, except for a "
                         "graph-captured step with less than 256 MB of gradients (a forked graph is launched node by node by the host)")
    return ap.parse_args()


T_START = time.perf_counter()


def note(msg):
    """Progress to stderr (stdout carries only the JSON line)."""
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def time_launches(fn, iters, warm=3):
    """Average duration (s) of fn()'s launches, HIP events on the current (launch) stream."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def kernel_rooflines(B, dev, iters):
    """Time each hand-written kernel at this workload's layer shape; return per-kernel dicts."""
    from amk import ops

    # a full untimed pass in front of every timed one: the first launches of a kernel after other work run at another
    # clock and with cold caches (tools/kbench_steady.py: round 0 against rounds 1+); what is reported is the steady state
    def time_steady(fn, n):
        return time_launches(fn, n, warm=max(3, n))

    H, T, D = VIT["n_heads"], (VIT["img_size"] // VIT["patch_size"]) ** 2, VIT["d_head"]
    g = torch.Generator().manual_seed(99)
    mk = lambda: torch.randn(B, T, H, D, generator=g).to(dev).permute(0, 2, 1, 3)  # (B,T,h*d) storage
    q, k, v, d_o = mk(), mk(), mk(), mk()
    scale = D ** -0.5
    q, k, v, o, stats, scores = ops._attn_forward(q, k, v, None, None, scale, keep_scores=True)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    from amk import lib as amk_lib
    delta = torch.empty(amk_lib.load().amk_attn_bwd_ws_floats(B, H, T, T, 72), device=dev)  # deltas + dq partials
    bwd = lambda st, sc=None: ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=st,
                                                 delta=delta, scores=sc)
    bwd(1)
    main = 72 if ops.DETERMINISTIC_ATTENTION_BACKWARD else 8  # the default path: fused pass (+ reproducible dq)
    core = 4.0 * B * H * T * T * D  # algorithmic FLOP of the forward
    layers_f, layers_b = 4 * VIT["depth"], 2 * VIT["depth"]  # per step: 2 model fwd + 1 bwd, enc+dec
    out = []
    t = time_steady(lambda: ops._attn_forward(q, k, v, None, None, scale), iters)
    out.append(dict(kernel="attn_fwd_kernel", launches_per_step=layers_f - layers_b, avg_ms=t * 1e3, flop=core,
                    note="4*B*h*I*J*d"))
    if scores is not None:
        t = time_steady(lambda: ops._attn_forward(q, k, v, None, None, scale, keep_scores=True), iters)
        out.append(dict(kernel="attn_fwd_keep_kernel", launches_per_step=layers_b, avg_ms=t * 1e3, flop=core,
                        note="4*B*h*I*J*d; also writes the raw scores (4*B*h*I*J bytes) for the backward"))
    old_mode, ops.ATTENTION_FORWARD = ops.ATTENTION_FORWARD, "bf16x6"
    t = time_steady(lambda: ops._attn_forward(q, k, v, None, None, scale), iters)
    ops.ATTENTION_FORWARD = old_mode
    out.append(dict(kernel="attn_fwd_x6 (split pre-pass + kernel)", launches_per_step=0, avg_ms=t * 1e3, flop=core,
                    note="the forward with split-bf16 products (amk_attn_fwd_x6); its bound is the bf16 MFMA peak / 6, "
                         "not the f32 MFMA peak the fraction below is taken against"))
    if scores is not None:
        t = time_steady(lambda: bwd(main, scores), iters)
        out.append(dict(kernel="attn_bwd_fused_kernel(kept scores)", launches_per_step=layers_b, avg_ms=t * 1e3,
                        flop=2 * core, note="8*B*h*I*J*d (dV,dP,dQ,dK: the four products it runs; S is read back from "
                                            "the forward's score tiles); includes the dq memset"))
        t = time_steady(lambda: bwd(72, scores), iters)
        out.append(dict(kernel="attn_bwd_fused_kernel(kept scores, reproducible dq)", launches_per_step=0, avg_ms=t * 1e3,
                        flop=2 * core, note="the same pass with dq as per-key-block partials + an ordered sum launch "
                                            "(no atomics, no memset; AMK_DETERMINISTIC=1)"))
    t = time_steady(lambda: bwd(main), iters)
    out.append(dict(kernel="attn_bwd_fused_kernel", launches_per_step=0 if scores is not None else layers_b,
                    avg_ms=t * 1e3, flop=2 * core,
                    note="8*B*h*I*J*d (dV,dP,dQ,dK; recomputed S not credited)"))
    t = time_steady(lambda: bwd(2), iters)
    out.append(dict(kernel="attn_bwd_dkdv_kernel", launches_per_step=0, avg_ms=t * 1e3, flop=core,
                    note="credited dV,dK products: 4*B*h*I*J*d (recomputed S, dP not credited)"))
    t = time_steady(lambda: bwd(4), iters)
    out.append(dict(kernel="attn_bwd_dq_kernel", launches_per_step=0, avg_ms=t * 1e3, flop=core,
                    note="credited dP,dQ products: 4*B*h*I*J*d (recomputed S not credited)"))
    N, K, C = B * T, CODEBOOK["codebook_size"], CODEBOOK["codebook_dim"]
    z = torch.randn(N, C, generator=g).to(dev)
    E = torch.randn(K, C, generator=g).to(dev)
    t = time_steady(lambda: ops.vq_lookup(z, E, 0.25), iters)
    out.append(dict(kernel="vq_lookup_fwd (prep+argmin+finalize)", launches_per_step=2, avg_ms=t * 1e3,
                    flop=2.0 * N * K * C, note="2*N*K*C"))
    # head dims 32 and 128 (csrc/attn_generic.hip: plain forward, two recompute kernels backward) at the same token
    # count and h * d = 512: not on the ViT-VQGAN path, reported so that every shipped kernel has a number
    for Dg in (32, 128):
        Hg = 512 // Dg
        mkg = lambda: torch.randn(B, T, Hg, Dg, generator=g).to(dev).permute(0, 2, 1, 3)
        qg, kg, vg, dog = mkg(), mkg(), mkg(), mkg()
        sg = Dg ** -0.5
        qg, kg, vg, og, stg, scg = ops._attn_forward(qg, kg, vg, None, None, sg, keep_scores=True)   # (kept for head dim 128 only)
        dqg, dkg, dvg = (torch.empty_like(qg) for _ in range(3))
        coreg = 4.0 * B * Hg * T * T * Dg
        t = time_steady(lambda: ops._attn_forward(qg, kg, vg, None, None, sg), iters)
        out.append(dict(kernel=f"attn_fwd_gen_kernel<{Dg}>", launches_per_step=0, avg_ms=t * 1e3, flop=coreg,
                        note=f"4*B*h*I*J*d at head dim {Dg}, {Hg} heads (unmasked: attn_fwd_gen_plain_kernel)"))
        if scg is not None:
            t = time_steady(lambda: ops._attn_backward(qg, kg, vg, og, stg, dog, dqg, dkg, dvg, None, None, sg, stages=9, scores=scg), iters)
            out.append(dict(kernel=f"attn_bwd_fused_gen_kernel<{Dg}>(kept scores) + delta", launches_per_step=0, avg_ms=t * 1e3,
                            flop=2 * coreg, note="8*B*h*I*J*d: the one-pass backward of csrc/attn_bwd_fused_gen.hip (four products)"))
        t = time_steady(lambda: ops._attn_backward(qg, kg, vg, og, stg, dog, dqg, dkg, dvg, None, None, sg, stages=9), iters)
        out.append(dict(kernel=f"attn_bwd_fused_gen_kernel<{Dg}> + delta", launches_per_step=0, avg_ms=t * 1e3,
                        flop=2 * coreg, note="8*B*h*I*J*d credited (five products: recomputed S not credited)"))
        t = time_steady(lambda: ops._attn_backward(qg, kg, vg, og, stg, dog, dqg, dkg, dvg, None, None, sg, stages=7), iters)
        out.append(dict(kernel=f"attn_bwd_gen_kernels<{Dg}> (delta + dkdv + dq)", launches_per_step=0, avg_ms=t * 1e3,
                        flop=2 * coreg, note="8*B*h*I*J*d credited (the reproducible recompute pair: seven products)"))
    for r in out:
        r["tflops"] = r["flop"] / (r["avg_ms"] * 1e-3) / 1e12
        r["frac_of_f32_mfma_peak"] = r["tflops"] / F32_MFMA_PEAK_TFLOPS
        r["ms_per_step"] = r["avg_ms"] * r["launches_per_step"]
    return out


def bf16_attention_block(B, dev, iters):
    """The bf16-MFMA attention kernels at the workload's layer shape: launch times, algorithmic TFLOP/s against the
    dense bf16 MFMA peak, and their error against the exact-f32 kernels on the same (bf16-rounded) operands."""
    from amk import ops

    H, T, D = VIT["n_heads"], (VIT["img_size"] // VIT["patch_size"]) ** 2, VIT["d_head"]
    g = torch.Generator().manual_seed(77)
    q2 = torch.randn(B, T, H * D, generator=g).to(dev).bfloat16().requires_grad_(True)
    kv2 = torch.randn(B, T, 2 * H * D, generator=g).to(dev).bfloat16().requires_grad_(True)
    cot = torch.randn(B, T, H * D, generator=g).to(dev).bfloat16()
    f = lambda: ops.attention_fused_kv(q2, kv2, H, D, D ** -0.5)
    t_f = time_launches(f, iters)
    t_fb = time_launches(lambda: torch.autograd.grad(f(), [q2, kv2], cot), iters)
    o16 = f()
    dq16, dkv16 = torch.autograd.grad(o16, [q2, kv2], cot)
    qf, kvf = q2.detach().float().requires_grad_(True), kv2.detach().float().requires_grad_(True)
    o32 = ops.attention_fused_kv(qf, kvf, H, D, D ** -0.5)
    dq32, dkv32 = torch.autograd.grad(o32, [qf, kvf], cot.float())
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())
    core = 4.0 * B * H * T * T * D
    t_b = t_fb - t_f
    # the FFN's bf16 GEMMs (csrc/gemm_bf16.hip) at the layer shape: HBM-bound (tiny weights, every activation byte once)
    from amk import dense

    M, Dm, Hf = B * T, VIT["dim"], (int(VIT["mlp_dim"] * 2 / 3) + 7) // 8 * 8   # (the SwiGLU hidden width: 1368)
    x = torch.randn(M, Dm, generator=g).to(dev).bfloat16()
    w12 = (torch.randn(2 * Hf, Dm, generator=g) * Dm ** -0.5).to(dev).bfloat16()
    b12 = torch.randn(2 * Hf, generator=g).to(dev)
    w3 = (torch.randn(Dm, Hf, generator=g) * Hf ** -0.5).to(dev).bfloat16()
    dy = torch.randn(M, Dm, generator=g).to(dev).bfloat16()
    gg, ab = dense.gemm_nt_swiglu_bf16(x, w12, b12)
    dab = dense.gemm_nn_swiglu_bwd_bf16(dy, w3, ab)
    gemm_rows = []
    for name, fn, nbytes, flop, per_step in [
            ("gemm_bf16_kernel<NT, SwiGLU> (w12 + gate)", lambda: dense.gemm_nt_swiglu_bf16(x, w12, b12),
             2.0 * (M * Dm + 2 * Hf * Dm + 3 * M * Hf), 4.0 * M * Dm * Hf, VIT["depth"]),
            ("gemm_bf16_kernel<NN, SwiGLU bwd> (dY W3 + gate backward)", lambda: dense.gemm_nn_swiglu_bwd_bf16(dy, w3, ab),
             2.0 * (M * Dm + Dm * Hf + 4 * M * Hf), 2.0 * M * Dm * Hf, VIT["depth"]),
            ("gemm_tn_bf16 (dW12, db12)", lambda: dense.gemm_tn_bf16(dab, x, want_bias=True),
             2.0 * (2 * M * Hf + M * Dm) + 4.0 * 2 * Hf * Dm, 4.0 * M * Dm * Hf, VIT["depth"]),
            ("gemm_tn_bf16 (dW3, db3)", lambda: dense.gemm_tn_bf16(dy, gg, want_bias=True),
             2.0 * (M * Hf + M * Dm) + 4.0 * Hf * Dm, 2.0 * M * Dm * Hf, VIT["depth"])]:
        t = time_launches(fn, iters)
        gemm_rows.append(dict(kernel=name, avg_launch_ms=t * 1e3, bound="hbm", achieved=nbytes / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=nbytes / t / 1e9 / HBM_PEAK_GBS, bytes_per_launch=nbytes, tflops=flop / t / 1e12, launches_per_step=per_step))
    del x, w12, b12, w3, dy, gg, ab, dab
    return {"roofline_bf16": {
        "bound": "mfma", "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "gemm_kernels": gemm_rows,
        "kernels": [dict(kernel="attn_bf16_fwd_kernel", avg_launch_ms=t_f * 1e3, achieved=core / t_f / 1e12,
                         frac=core / t_f / 1e12 / BF16_MFMA_PEAK_TFLOPS, flop_per_launch=core, launches_per_step=4 * VIT["depth"]),
                    dict(kernel="attn_bf16_bwd (delta + fused + dq reduce)", avg_launch_ms=t_b * 1e3,
                         achieved=2.5 * core / t_b / 1e12, frac=2.5 * core / t_b / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                         flop_per_launch=2.5 * core, launches_per_step=2 * VIT["depth"],
                         note="10*B*h*I*J*d: five products (S recomputed from the statistics)")]},
        "error_vs_f32_kernels": {"out": rel(o16, o32), "dq": rel(dq16, dq32), "dkv": rel(dkv16, dkv32),
                                 "what": "max |bf16 - f32| / max |f32| of the attention core on the same bf16-rounded operands"}}


def agent_block(dev, iters, batch=64):
    """Secondary, informational: the AgentAttention core (models/agent_attention.py:55-73; dim 384, 6 heads x 64, 1024
    tokens, 6 agents per head) forward and backward at batch 64, HBM-bound: algorithmic bytes 16*B*h*T*d forward (q, k,
    v read, o written) and 28*B*h*T*d backward (q, k, v, dO read, dq, dk, dv written), against the HBM peak and
    against the copy bandwidth measured on this box by a device-to-device copy of the same size."""
    from amk import ops
    from amk.models import AgentAttention

    torch.manual_seed(0)
    h, d, T = 6, 64, 1024
    ag = AgentAttention(h * d, h, d).to(dev)
    qkv = torch.randn(batch, T, 3 * h * d, device=dev, requires_grad=True)
    co = torch.randn(batch, T, h * d, device=dev)
    cw, cb = ag.dwc[1].weight, ag.dwc[1].bias
    core = lambda: ops.agent_attention(qkv, cw, cb, h, d, ag.pool_size, ag.scale)
    t_f = time_launches(core, iters)
    t_fb = time_launches(lambda: torch.autograd.grad(core(), [qkv], co), iters)
    t_b = t_fb - t_f
    src = torch.empty(batch * T * h * d * 2, device=dev)   # 8 B per element moved: half of the forward's bytes
    dst = torch.empty_like(src)
    t_c = time_launches(lambda: dst.copy_(src), iters)
    copy_gbs = 2.0 * src.numel() * 4 / t_c / 1e9
    unit = float(batch * h * T * d)
    rows = []
    for name, t, nbytes in [("agent forward (pool, s1 partial, s1 combine, s2)", t_f, 16.0 * unit),
                            ("agent backward (s2 bwd, mid, s1 bwd, pool bwd)", t_b, 28.0 * unit)]:
        rows.append(dict(kernel=name, avg_ms=t * 1e3, bytes=nbytes, achieved=nbytes / t / 1e9, unit="GB/s", bound="hbm",
                         frac_of_hbm_peak=nbytes / t / 1e9 / HBM_PEAK_GBS, frac_of_measured_copy=nbytes / t / 1e9 / copy_gbs))
    del ag, qkv, co, src, dst
    return {"workload": f"AgentAttention core, batch {batch}, 6 heads x 64, 1024 tokens, 6 agents, f32",
            "measured_copy_gbs": copy_gbs, "kernels": rows}


def vitmoe_block(dev, batch=64, steps=10):
    """Secondary, informational: BASELINE.json configs[3] (ViTMoE dim 1024, patch 32, depth 6, 32 experts top-2,
    SwitchHead h 8) forward + backward at batch 64, with HIP events around every routed-expert launch.
    Grouped expert GEMMs are credited 2*P*N*K FLOP (P routed pairs: the reference's work, also where a launch marked
    "(distinct rows)" forms an expert's product once per distinct (token, expert)), against the f32 MFMA peak."""
    import re

    from amk import ops as amk_ops
    from amk.models import ViTMoE

    torch.manual_seed(0)
    vm = ViTMoE(dim=1024, image_size=256, patch_size=32, n_heads=8, d_head=64, depth=6, n_experts=32, sel_experts=2,
                dropout=0.0, num_classes=1000).to(dev)
    g = torch.Generator().manual_seed(4321)
    imgs = torch.randn(batch, 3, 256, 256, generator=g).to(dev)
    labels = torch.randint(0, 1000, (batch,), generator=g).to(dev)

    def step():
        vm.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(vm(imgs), labels).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    amk_ops.KERNEL_EVENTS = {}
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ev = amk_ops.kernel_event_summary(amk_ops.KERNEL_EVENTS)
    amk_ops.KERNEL_EVENTS = None
    rows = []
    for name, (n, ms) in sorted(ev.items()):
        r = dict(kernel=name, launches_per_step=n / steps, avg_ms=ms, ms_per_step=ms * n / steps)
        m = re.match(r"grouped_\w+ P(\d+) N(\d+) K(\d+)", name)
        if m:
            fl = 2.0 * int(m.group(1)) * int(m.group(2)) * int(m.group(3))
            r.update(flop=fl, tflops=fl / (ms * 1e-3) / 1e12, frac_of_f32_mfma_peak=fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS)
        m = re.match(r"dense_z_gemm M(\d+) N(\d+) K(\d+)", name)
        if m:
            # SwitchHead's head / slot sums as one library GEMM over per-expert sums: the FLOPs it EXECUTES (E / pairs-per-row
            # = 2x the routed ones at this layer), so the fraction is the GEMM's own efficiency
            fl = 2.0 * int(m.group(1)) * int(m.group(2)) * int(m.group(3))
            r.update(flop=fl, note="executed FLOPs of the dense form (2x the routed ones); vendor GEMM",
                     tflops=fl / (ms * 1e-3) / 1e12, frac_of_f32_mfma_peak=fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS)
        rows.append(r)
    del vm
    torch.cuda.empty_cache()
    return {"workload": "BASELINE.json configs[3]: ViTMoE dim=1024 patch=32 depth=6 n_experts=32 top-2 + SwitchHead h=8, "
                        f"forward + cross-entropy + backward, batch {batch}, f32",
            "ms_per_step": dt * 1e3, "images_per_s": batch / dt, "kernels": rows}


def dp_graph_ok(dev, world):
    """Can this process group's collectives be captured into a HIP graph?  A pre-flight on a bucket-sized buffer: an all-reduce
    on a side stream inside a capture, two replays, the result checked; every rank must agree (MIN over ranks), else the
    data-parallel step runs eagerly.  World of one: nothing to capture."""
    if world == 1 and not dist.is_initialized():
        return True, "one rank"
    if os.environ.get("AMK_DP_GRAPH", "auto") == "0":
        return False, "AMK_DP_GRAPH=0"
    if dist.get_backend() != "nccl":
        return False, f"backend {dist.get_backend()} cannot be captured"
    ok, why = 1.0, "probe passed"
    try:
        buf = torch.ones(8 << 20, device=dev)   # 32 MiB: one gradient bucket (the protocol RCCL picks depends on the size)
        side = torch.cuda.Stream(device=dev)
        dist.all_reduce(buf, op=dist.ReduceOp.AVG)   # communicator warm-up outside the capture
        torch.cuda.synchronize()
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        time.sleep(0.5)   # (the watchdog drops the finished warm-up collective: see amk/graphs.py)
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap, capture_error_mode="thread_local"):
                buf.mul_(2.0)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    dist.all_reduce(buf, op=dist.ReduceOp.AVG)
                torch.cuda.current_stream().wait_stream(side)
                buf.add_(1.0)
        graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        if abs(float(buf[0]) - 7.0) > 1e-6 or abs(float(buf[-1]) - 7.0) > 1e-6:   # ((1*2+1)*2+1)
            ok, why = 0.0, f"probe replay gave {float(buf[0])}, expected 7"
    except Exception as e:  # noqa: BLE001 -- any failure means: run eagerly
        ok, why = 0.0, f"probe raised {type(e).__name__}: {e}"
    t = torch.tensor([ok], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if float(t.item()) < 1.0 and ok == 1.0:
        why = "probe failed on another rank"
    return float(t.item()) >= 1.0, why


def guarded_graph_attempt(attempt, fallback_line, rank):
    """Run `attempt()` (pre-flight, capture and timing of the data-parallel step with its RCCL collectives inside a HIP graph)
    under a deadline.  A captured collective that never completes cannot be caught as an exception: when the deadline passes,
    rank 0 prints the already measured eager line and every rank leaves with os._exit(0) (each rank runs its own timer; a rank
    is past the attempt only when the attempt's last collective -- the MAX over ranks of the timed steps -- has returned,
    i.e. when every rank has finished it).  Returns attempt()'s result."""
    import threading

    deadline = float(os.environ.get("AMK_DP_GRAPH_DEADLINE", "150"))
    done = threading.Event()

    def watchdog():
        if done.wait(deadline):
            return
        note(f"captured data-parallel step did not come back within {deadline:.0f} s: reporting the eager step")
        if rank == 0:
            fallback_line["graph_decision"] = f"captured-collectives attempt did not finish within {deadline:.0f} s; eager step reported"
            emit(fallback_line)
        os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return attempt()
    finally:
        done.set()


def capture_or_eager(capture, release, reducers, world, dev):
    """Capture the step; with several ranks every rank must have succeeded (MIN over ranks, an eager collective: the ones
    inside a capture were recorded, not run) -- otherwise every rank drops its graph and the step runs eagerly, with the
    all-reduces back on the side stream.  Returns (graphed, reason)."""
    ok, why = 1.0, "captured"
    try:
        capture()
    except Exception as e:  # noqa: BLE001
        ok, why = 0.0, f"capture raised {type(e).__name__}: {e}"
        note(why)
    if dist.is_initialized():
        t = torch.tensor([ok], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if float(t.item()) < 1.0 and ok == 1.0:
            why = "capture failed on another rank"
        ok = float(t.item())
    if ok < 1.0:
        release()
        for red in reducers:
            red.overlap = True
        return False, why
    return True, why


SECONDARY = {
    "vit": dict(metric="ViT 256px classifier train-step images/sec", batch=64,
                workload="BASELINE.json configs[1]: ViT dim=1024 patch=32 img=256 depth=6 h=16 d=64 (33.6 M parameters); "
                         "trainers/vit.py step: CE, AdamW (wd 0.01), clip 1.0, cosine schedule with warm-up"),
    "vitmoe": dict(metric="ViTMoE 256px classifier train-step images/sec", batch=64,
                   workload="BASELINE.json configs[3]: ViTMoE dim=1024 patch=32 depth=6 n_experts=32 top-2 + SwitchHeadAttention "
                            "h=8 (240.6 M parameters, 962 MB of gradients per all-reduce); trainers/vit.py step: CE, AdamW, clip, cosine"),
    "muse": dict(metric="Muse masked-token decoder train-step images/sec", batch=8,
                 workload="BASELINE.json configs[4]: Muse decoder dim=1024 h=16 depth=22 mult=6 over FROZEN ViTVQGAN codes "
                          "(1024 tokens, 77 synthetic text positions); trainers/muse.py step: masked-token CE, AdamW, clip"),
}


def _overlap(choice, graphed, communicating, model):
    if choice != "auto":
        return choice == "on"
    nbytes = 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
    return not (graphed and communicating and nbytes < (256 << 20))


def secondary_main(args, world, rank, dev, n_ranks_seen):
    """--model vit | vitmoe | muse: the data-parallel single-model step (amk.train.ClassifierTrainStep /
    MaskedTokenTrainStep) -- a secondary line, never the headline."""
    from amk import lib, tuning
    from amk.train import ClassifierTrainStep, MaskedTokenTrainStep

    lib.load()
    if os.environ.get("AMK_TUNABLEOP", "1") == "1":
        tuning.enable_gemm_tuning()
    spec = SECONDARY[args.model]
    batch = args.batch if args.batch != 32 else spec["batch"]
    amp = torch.bfloat16 if args.autocast == "bf16" else None
    can_graph, graph_why = (False, "--no-graph") if args.no_graph else dp_graph_ok(dev, world)
    alone_rccl = os.environ.get("AMK_BENCH_RCCL_ALONE", "0") == "1"
    kw = dict(capturable=can_graph, autocast=amp, communicate_when_alone=alone_rccl, max_grad_norm=1.0)
    overlap_arg = args.dp_overlap
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234 + rank)
    if args.model == "muse":
        from amk.models import MUSE, ViTVQGAN

        vq = ViTVQGAN(VIT, CODEBOOK)
        model = MUSE(dim=1024, vq=vq, n_heads=16, d_head=64, depth=22, mult=6).to(dev)
        kw["overlap"] = _overlap(overlap_arg, can_graph, world > 1 or alone_rccl, model)
        step = MaskedTokenTrainStep(model, lr=1e-4, weight_decay=0.0, warmup_steps=1000, **kw)
        data = (torch.randn(batch, 77, 768, generator=g).to(dev), torch.rand(batch, 3, 256, 256, generator=g).to(dev))
    else:
        from amk.models import ViT, ViTMoE

        if args.model == "vit":
            model = ViT(dim=1024, image_size=256, patch_size=32, n_heads=16, d_head=64, depth=6, mlp_dim=2048, dropout=0.0,
                        num_classes=1000).to(dev)
        else:
            model = ViTMoE(dim=1024, image_size=256, patch_size=32, n_heads=8, d_head=64, depth=6, n_experts=32, sel_experts=2,
                           dropout=0.0, num_classes=1000).to(dev)
        kw["overlap"] = _overlap(overlap_arg, can_graph, world > 1 or alone_rccl, model)
        step = ClassifierTrainStep(model, lr=1e-4, warmup_steps=1000, total_steps=100000, **kw)
        data = (torch.randn(batch, 3, 256, 256, generator=g).to(dev), torch.randint(0, 1000, (batch,), generator=g).to(dev))
    nparam = sum(p.numel() for p in step.red.params)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"rank {rank}/{world}: {args.model} built ({nparam / 1e6:.1f} M trainable parameters), batch {batch}/GPU")
    for _ in range(max(args.warmup, 2)):   # (step 0 records the static-unused parameters; overlap starts with step 1)
        step.step(*data)
    graph = "eager"
    if can_graph:
        done, cap_why = capture_or_eager(lambda: step.capture(*data), step.release_graph, [step.red], world, dev)
        if done:
            graph = "one HIP-graph replay per step" + (" (RCCL all-reduces captured inside)" if not step.red.alone else "")
        else:
            graph_why = cap_why
    note(f"warm-up done ({graph}; {graph_why})")
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step.step(*data)
        if i == min(args.steps, 3) - 1:
            t_host = (time.perf_counter() - t0) / (i + 1) * args.steps   # (first three: later ones can block on a full device queue)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        emit({
            "metric": spec["metric"], "value": batch * world * args.steps / dt, "unit": "images/s", "n_gpus": world,
            "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "host_enqueue_ms": t_host / args.steps * 1e3, "step_launch": graph, "graph_decision": graph_why,
            "dp_allreduce": ("side stream, overlapped with backward" if step.red.overlap else "compute stream, at bucket completion")
                            if not step.red.alone else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if amp is None else "bf16 (autocast)", "data": "synthetic N(0,1) images, random-init weights",
            "secondary": True,
            "config": {"workload": spec["workload"], "global_batch": batch * world, "batch_per_gpu": batch,
                       "parallelism": f"dp{world}", "final_loss": float(loss),
                       "gradient_bytes_per_allreduce": step.red.grads_nbytes(), "buckets": len(step.red.buckets),
                       "static_unused_parameters": len(step.red.static_unused_parameters())}})
    if world > 1 or dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def rccl_world_of_one_block(args, dev, init_state, imgs):
    """Informational, one-GPU runs only: the data-parallel form of the headline step on THIS box -- a world of one rank that
    issues its RCCL all-reduces anyway (GradReducer(communicate_when_alone=True)), captured into one HIP graph with the
    collectives inside (compute-stream all-reduces, as `--dp-overlap auto` picks for this model) and, for comparison, eager
    with the all-reduces on the side stream.  What N > 1 ranks execute, minus the wire."""
    import socket

    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    out = {"what": "the headline step as its data-parallel form on one GPU: RCCL world of one rank, all-reduces issued "
                   "(AMK_BENCH_RCCL_ALONE=1 runs the whole bench this way); not the headline value"}
    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:  # noqa: BLE001
        out["error"] = f"init_process_group: {type(e).__name__}: {e}"
        return out
    try:
        ok, why = dp_graph_ok(dev, 1)
        out["capture_preflight"] = why
        for mode in (("graph", False), ("eager", True)):
            if mode[0] == "graph" and not ok:
                continue
            torch.manual_seed(0)
            model = ViTVQGAN(VIT, CODEBOOK)
            model.load_state_dict(init_state)
            model = model.to(dev)
            discr = NLayerDiscriminator(3, 64, 3).to(dev)
            tr = VQGANTrainStep(model, discr, capturable=mode[0] == "graph", communicate_when_alone=True, overlap=mode[1])
            for _ in range(2):
                tr.step(imgs)
            if mode[0] == "graph":
                done, why2 = capture_or_eager(lambda: tr.capture(imgs), tr.release_graph, [tr.g_red, tr.d_red], 1, dev)
                if not done:
                    out["graph"] = {"error": why2}
                    continue
            n = max(3, min(args.steps, 10))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                tr.step(imgs)
                if i == 2:
                    t_host = (time.perf_counter() - t0) / 3
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            out[mode[0]] = {"ms_per_step": dt * 1e3, "host_enqueue_ms": t_host * 1e3, "images_per_s": args.batch / dt,
                            "dp_allreduce": "side stream, overlapped with backward" if mode[1] else "compute stream, at bucket completion",
                            "buckets": len(tr.g_red.buckets) + len(tr.d_red.buckets),
                            "gradient_bytes": tr.g_red.grads_nbytes() + tr.d_red.grads_nbytes()}
            tr.release_graph()
            del tr, model, discr
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()
    return out


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: start the N rank processes from here.  The parent
    never touches HIP (no torch.cuda call), passes its own flags through, relays rank 0's JSON line and
    fails if any rank fails.  (The driver's torch.distributed.run launch sets WORLD_SIZE and skips this.)"""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    os.write(REAL_STDOUT, out)
    if any(rcs):
        raise SystemExit(f"bench.py: rank exit codes {rcs}")


def latest_pmc_digest():
    """Newest committed PMC digest (profiles/rNN_pmc_kernels_b32.json) or (None, None)."""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_kernels_b*.json")))
    if not paths:
        return None, None
    return json.load(open(paths[-1])), os.path.relpath(paths[-1], ROOT)


def emit(line):
    """The ONE JSON line, on the process's real stdout."""
    os.write(REAL_STDOUT, (json.dumps(line) + "\n").encode())


REAL_STDOUT = 1


def main():
    global REAL_STDOUT
    args = parse()
    # stdout carries the one JSON line and nothing else: native libraries write to fd 1 behind Python's back (RCCL prints
    # a version banner there when a communicator is created), so fd 1 is pointed at stderr for the life of the process and
    # the line goes to a duplicate of the original
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    # dmabuf IPC is what RCCL needs on this pool; set before the first HIP call, for BOTH launch modes (the driver's
    # torch.distributed.run launch never passes through self_launch)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run (see docstring)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible (there is no CPU path)")
    # rehearsal switch for a 1-GPU box: AMK_REHEARSE_SHARED_GPU=1 puts every rank on cuda:0 over gloo
    # (RCCL refuses two ranks on one device); the timed numbers of such a run mean nothing.
    rehearse = os.environ.get("AMK_REHEARSE_SHARED_GPU", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    alone_rccl = world == 1 and os.environ.get("AMK_BENCH_RCCL_ALONE", "0") == "1"
    if alone_rccl:
        # one rank that communicates anyway: the RCCL path (eager or captured) on a one-GPU box
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n_ranks_seen = dist.get_world_size() if world > 1 else 1
    if args.model != "vqgan":
        return secondary_main(args, world, rank, dev, n_ranks_seen)

    from amk import lib
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    lib.load()
    # vendor-library kernel selection for the non-hot-path GEMMs / convolutions (amk/tuning)
    from amk import tuning
    tuning.enable_conv_autotune(os.environ.get("AMK_MIOPEN_FIND", "1") == "1")
    if os.environ.get("AMK_TUNABLEOP", "1") == "1":
        tuning.enable_gemm_tuning()
    torch.manual_seed(0)  # identical init on every rank (and broadcast from rank 0 anyway)
    model = ViTVQGAN(VIT, CODEBOOK)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev)
    discr = NLayerDiscriminator(3, 64, 3).to(dev)
    # one rank: the device side of a step (both phases, both optimizers) is captured into a HIP graph after the
    # warm-up and replayed -- the host then enqueues one launch per step instead of ~2500 (tools/host_vs_gpu.py:
    # 92.7 ms of host work per step in round 2); N ranks: eager steps (the reducers' RCCL collectives run on a side stream)
    # N ranks over RCCL: the same single replay, with the reducers' all-reduces (side stream) captured inside it -- after
    # a pre-flight that captures and replays one small all-reduce on every rank (dp_graph_ok); over gloo (rehearsal): eager
    communicating = world > 1 or alone_rccl
    # Ranks that communicate over RCCL: the eager step (all-reduces on a side stream) is timed FIRST -- it is a valid headline on
    # its own -- and the captured step (RCCL all-reduces inside the graph: never run on more than one GPU before the driver's
    # scaling run) is attempted afterwards under a deadline: if the attempt does not come back, rank 0 prints the eager line
    # and every rank leaves (guarded_graph_attempt).  One rank without communication: the graph directly, as before.
    # AMK_DP_GRAPH: "auto" (default) attempts the captured step only where it can matter -- when the eager step is HOST-bound
    # (host enqueue >= 0.6 of the step on some rank: the bf16 step); the f32 step is GPU-bound (46 of 108 ms) and a native
    # crash inside a captured collective, unlike a hang, could not be turned into a line.  "1": always attempt; "0": never.
    dp_graph_env = os.environ.get("AMK_DP_GRAPH", "auto")
    eager_first = communicating and not rehearse and not args.no_graph and dp_graph_env in ("auto", "1")
    if args.no_graph:
        use_graph, graph_why = False, "--no-graph"
    elif eager_first:
        use_graph, graph_why = False, "eager first"
    else:
        use_graph, graph_why = dp_graph_ok(dev, world)
    overlap = {"on": True, "off": False, "auto": not (use_graph and communicating)}[args.dp_overlap]
    bf16 = args.autocast == "bf16"
    if bf16:
        args.no_kernels = args.no_variants = args.no_cpu_baseline = True
    trainer = VQGANTrainStep(model, discr, capturable=use_graph or eager_first, communicate_when_alone=alone_rccl,
                             overlap=True if eager_first else overlap, autocast=torch.bfloat16 if bf16 else None)
    g = torch.Generator().manual_seed(1234 + rank)
    imgs = torch.rand(args.batch, 3, VIT["img_size"], VIT["img_size"], generator=g).to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from amk import ops as amk_ops

    host_enqueue = [None]

    def timed_steps(n):
        sync()
        t0 = time.perf_counter()
        for i in range(n):
            logs = trainer.step(imgs)
            if i == min(n, 3) - 1:
                # host time to ENQUEUE a step, over the first three (later ones can block on a full device queue: that
                # is the device's pace, not the host's)
                host_enqueue[0] = (time.perf_counter() - t0) / (i + 1)
        sync()
        return time.perf_counter() - t0, logs

    def max_over_ranks(x):
        if world > 1:
            tt = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return x

    def headline(dt_, host_ms, graphed, why, ovl, loss_):
        return {
            "metric": "ViTVQGAN 256px train-step images/sec",
            "value": args.batch * world * args.steps / dt_,
            "unit": "images/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_ / args.steps * 1e3,
            "host_enqueue_ms": host_ms,
            "step_launch": ("one HIP-graph replay per step" + (" (RCCL all-reduces captured inside)" if communicating else ""))
                           if graphed else "eager",
            "graph_decision": why,
            "dp_allreduce": ("side stream, overlapped with backward" if ovl else "compute stream, at bucket completion") if communicating else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 (autocast)" if bf16 else "f32",
            **({"secondary": True} if bf16 else {}),
            "data": "synthetic U[0,1) images, random-init weights",
            "config": {
                "workload": "BASELINE.json configs[2]: ViTVQGAN dim=256 patch=8 img=256 depth=6+6 h=8 d=64 "
                            "mlp=2048 (SwiGLU), codebook 8192x32; GAN train step (D phase + G phase, "
                            "gradient penalty, per_loss_weight=0), Adam, clip 1.0",
                "global_batch": args.batch * world,
                "batch_per_gpu": args.batch,
                "tokens_per_image": (VIT["img_size"] // VIT["patch_size"]) ** 2,
                "parallelism": f"dp{world}",
                "final_loss": loss_,
            },
        }

    note(f"rank {rank}/{world}: model built, batch {args.batch}/GPU")
    for _ in range(args.warmup):
        trainer.step(imgs)
    if use_graph:
        use_graph, cap_why = capture_or_eager(lambda: trainer.capture(imgs), trainer.release_graph, [trainer.g_red, trainer.d_red], world, dev)
        if not use_graph:
            graph_why, overlap = cap_why, True
    note("warm-up done" + (" (step captured into a HIP graph)" if use_graph else ""))
    # ---- the headline: EXACTLY --steps steps, no instrumentation, barrier + synchronize on both sides
    dt, logs = timed_steps(args.steps)
    host_enqueue_ms = host_enqueue[0] * 1e3
    dt = max_over_ranks(dt)
    loss = float(logs["loss"])
    dp_modes = None
    if eager_first:
        overlap = True
        dp_modes = {"eager": {"ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms": host_enqueue_ms, "dp_allreduce": "side stream, overlapped with backward"}}
        note(f"eager data-parallel step timed: {dt / args.steps * 1e3:.2f} ms; attempting the captured step under a deadline")
        fallback = headline(dt, host_enqueue_ms, False, "placeholder", True, loss)
        host_bound = max_over_ranks(host_enqueue_ms / (dt / args.steps * 1e3)) >= 0.6

        def attempt():
            ok, why = dp_graph_ok(dev, world)
            if os.environ.get("AMK_BENCH_FAKE_HANG") == "1":   # (test hook: the deadline path)
                time.sleep(1e6)
            if not ok:
                return None, why
            inline = {"on": False, "off": True, "auto": True}[args.dp_overlap]   # captured: all-reduces on the compute stream unless asked
            for red in (trainer.g_red, trainer.d_red):
                red.overlap = not inline
            ok, why = capture_or_eager(lambda: trainer.capture(imgs), trainer.release_graph, [trainer.g_red, trainer.d_red], world, dev)
            if not ok:
                return None, why
            dtg, lg = timed_steps(args.steps)
            return (max_over_ranks(dtg), host_enqueue[0] * 1e3, float(lg["loss"]), not inline), why

        if dp_graph_env == "auto" and not host_bound:
            res, why = None, (f"eager: the step is GPU-bound (host enqueue {host_enqueue_ms:.0f} of {dt / args.steps * 1e3:.0f} ms per step), so the "
                              "captured-collectives step was not attempted (AMK_DP_GRAPH=1 attempts it; measured with a world of one "
                              "rank in dp_step_rccl_world_of_one of the one-GPU line)")
        else:
            res, why = guarded_graph_attempt(attempt, fallback, rank)
        if res is not None:
            dtg, hg, lossg, ovl_g = res
            dp_modes["graph"] = {"ms_per_step": dtg / args.steps * 1e3, "host_enqueue_ms": hg,
                                 "dp_allreduce": "side stream, overlapped with backward" if ovl_g else "compute stream, at bucket completion"}
            # every rank takes the same branch: both times are MAX over ranks
            if dtg <= dt:
                dt, host_enqueue_ms, loss, use_graph, overlap = dtg, hg, lossg, True, ovl_g
                graph_why = "captured step at least as fast as the eager one (both timed: dp_step_modes)"
            else:
                trainer.release_graph()
                for red in (trainer.g_red, trainer.d_red):
                    red.overlap = True
                graph_why = "captured step slower than the eager one (both timed: dp_step_modes)"
        else:
            graph_why = why
        note(f"data-parallel step: {graph_why}")
    note(f"timed {args.steps} steps in {dt:.3f}s")

    # ---- a separate short instrumented pass (rank 0): HIP events on the launch stream around every
    # hot-path launch of real train steps -> the in-step kernel figures of `roofline` / `kernels_in_step`
    in_situ, n_inst = {}, 0
    if rank == 0 and world == 1 and not args.no_kernels:
        n_inst = max(2, min(5, args.steps))
        trainer.release_graph()  # the events are recorded launch by launch: eager steps
        amk_ops.KERNEL_EVENTS = {}
        timed_steps(n_inst)
        in_situ = amk_ops.kernel_event_summary(amk_ops.KERNEL_EVENTS)
        amk_ops.KERNEL_EVENTS = None
        note("instrumented pass done")

    def variant(setup, teardown, what):
        trainer.release_graph()
        setup()
        for _ in range(2):
            trainer.step(imgs)
        if use_graph:
            trainer.capture(imgs)
        dv, _ = timed_steps(args.steps)
        trainer.release_graph()
        teardown()
        return {"value": args.batch * args.steps / dv, "unit": "images/s", "ms_per_step": dv / args.steps * 1e3, "what": what}

    variants = {}
    if world == 1 and not args.no_variants:
        # informational variants, never `value`
        variants["variant_shared_generator_forward"] = variant(
            lambda: setattr(trainer, "share_forward", True), lambda: setattr(trainer, "share_forward", False),
            "generator forward run once per step and shared by the discriminator and generator phases (the "
            "reference runs it twice on unchanged weights); not the headline value")
        note("shared-forward variant done")
        variants["variant_reproducible_dq"] = variant(
            lambda: setattr(amk_ops, "DETERMINISTIC_ATTENTION_BACKWARD", True),
            lambda: setattr(amk_ops, "DETERMINISTIC_ATTENTION_BACKWARD", False),
            "the headline step with the attention backward storing per-key-block dq partials and summing them in order "
            "(bitwise reproducible gradients; AMK_DETERMINISTIC=1 or torch.use_deterministic_algorithms(True)) instead of "
            "adding dq by f32 atomics")
        note("reproducible-dq variant done")
        variants["variant_recomputed_scores"] = variant(
            lambda: setattr(amk_ops, "ATTENTION_KEEP_SCORES", False), lambda: setattr(amk_ops, "ATTENTION_KEEP_SCORES", True),
            "the headline step with the attention backward recomputing S = QK^T (five products) instead of reading "
            "the scores the forward kept in HBM (four products); identical results")
        note("recomputed-scores variant done")
        variants["variant_split_bf16_attention_forward"] = variant(
            lambda: setattr(amk_ops, "ATTENTION_FORWARD", "bf16x6"), lambda: setattr(amk_ops, "ATTENTION_FORWARD", "f32"),
            "attention forward with split-bf16 (bf16x6) products, f32-level error; the backward then recomputes S; "
            "not the headline value")
        note("split-bf16 forward variant done")
        variants["variant_split_bf16_linear_gemms"] = variant(
            lambda: setattr(amk_ops, "GEMM_MODE", "bf16x6"), lambda: setattr(amk_ops, "GEMM_MODE", "f32"),
            "every nn.Linear of the generator (projections, FFN, patch / quant layers): forward and input gradient on the "
            "split-bf16 GEMM csrc/gemm_x6.hip (f32 operands, six exact bf16 partial products per product, f32 "
            "accumulation: f32-level error, tests/test_gemm_x6_gpu.py) instead of the library's exact-f32 GEMM; weight "
            "gradients stay with the library; its bound is the bf16 MFMA peak / 6; not the headline value")
        note("split-bf16 GEMM variant done")

        def both_on():
            amk_ops.GEMM_MODE, amk_ops.ATTENTION_FORWARD = "bf16x6", "bf16x6"

        def both_off():
            amk_ops.GEMM_MODE, amk_ops.ATTENTION_FORWARD = "f32", "f32"

        variants["variant_split_bf16_gemms_and_attention_forward"] = variant(
            both_on, both_off, "the two split-bf16 variants above together; not the headline value")
        note("combined split-bf16 variant done")
        # the reference's shipped training precision (cfg/vitvqgan.yaml:73: accelerate bf16 autocast): its own line,
        # its own dtype, its own roofline -- never the headline (the north star's tolerance is an f32 one)
        v16 = variant(lambda: setattr(trainer, "autocast", torch.bfloat16), lambda: setattr(trainer, "autocast", None),
                      "both phases' forwards and losses under torch.autocast(bfloat16) as the reference's accelerator.autocast() "
                      "blocks run them (trainers/vitgqgan.py:149,170): Linear GEMMs in bf16 (weight / bias gradients, the w12 + SwiGLU forward and the dY W3 + "
                      "SwiGLU backward on csrc/gemm_bf16.hip, the rest on the vendor library; bf16 parameter copies refreshed by the optimizer "
                      "kernel), discriminator convolutions in bf16 (MIOpen), "
                      "attention on the bf16-MFMA kernels of csrc/attn_bf16.hip (bf16 operands, f32 scores / softmax / "
                      "accumulators), VQ lookup exact f32; parameters, gradients and optimizer state f32; not the headline value")
        v16["dtype"] = "bf16 (autocast)"
        v16.update(bf16_attention_block(args.batch, dev, args.kernel_iters))
        variants["variant_bf16_autocast"] = v16
        note("bf16-autocast variant done")
        # the reference's shipped schedule (cfg/vitvqgan.yaml:37,72-76): batch_size 8, gradient_accumulation_steps 2, bf16
        # autocast -- two micro-batches of 8 per optimizer step, eager (the accumulation micro-step is not the captured one)
        trainer.release_graph()
        trainer.autocast = torch.bfloat16
        img8a, img8b = imgs[:8].contiguous(), imgs[8:16].contiguous()

        def shipped_step():
            return trainer.step_accumulated([img8a, img8b])

        for _ in range(2):
            shipped_step()
        if use_graph:
            trainer.capture_accumulated([img8a, img8b])
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            shipped_step()
        sync()
        dts = time.perf_counter() - t0
        trainer.release_graph()
        trainer.autocast = None
        variants["variant_shipped_schedule"] = {
            "value": 16 * args.steps / dts, "unit": "images/s", "ms_per_optimizer_step": dts / args.steps * 1e3, "dtype": "bf16 (autocast)",
            "what": "the reference's shipped schedule, cfg/vitvqgan.yaml:37,72-76: batch_size 8 x gradient_accumulation_steps 2 under "
                    "bf16 autocast (two micro-batches per optimizer step, VQGANTrainStep.step_accumulated: the whole optimizer step as "
                    "one HIP-graph replay unless --no-graph); not the headline value"}
        note("shipped-schedule variant done")

    kernels = None
    if rank == 0 and not args.no_kernels:
        kernels = kernel_rooflines(args.batch, dev, args.kernel_iters)
        note("kernel rooflines done")
    vitmoe = None
    if rank == 0 and world == 1 and not args.no_kernels and not args.no_variants:
        vitmoe = vitmoe_block(dev)
        note("ViTMoE block done")
    agent = None
    if rank == 0 and world == 1 and not args.no_kernels and not args.no_variants:
        agent = agent_block(dev, args.kernel_iters)
        note("AgentAttention block done")
    dp_one = None
    if rank == 0 and world == 1 and not alone_rccl and not args.no_variants and not bf16:
        trainer.release_graph()
        dp_one = rccl_world_of_one_block(args, dev, init_state, imgs)
        note("RCCL world-of-one block done")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import train_step_cpu

        ips, cores, sample = train_step_cpu.time_train_step(init_state, VIT, NLayerDiscriminator(3, 64, 3))
        cpu = dict(value=ips, unit="images/s", cores=cores, kind="port", sample=sample)
        note("cpu baseline done")

    if rank == 0:
        line = headline(dt, host_enqueue_ms, use_graph, graph_why, overlap, loss)
        if dp_modes:
            line["dp_step_modes"] = dp_modes
        T_ = (VIT["img_size"] // VIT["patch_size"]) ** 2
        core = 4.0 * args.batch * VIT["n_heads"] * T_ * T_ * VIT["d_head"]
        flop_of = {"attn_fwd_kernel": core, "attn_fwd_keep_kernel": core, "attn_bwd_fused_kernel": 2 * core,
                   "attn_bwd_fused_kernel(kept scores)": 2 * core, "attn_bwd_dkdv+dq": 2 * core,
                   "vq_lookup_fwd": 2.0 * args.batch * T_ * CODEBOOK["codebook_size"] * CODEBOOK["codebook_dim"]}
        timed = []
        for name, (n, ms) in in_situ.items():
            fl = flop_of.get(name)
            if fl is None:
                continue
            timed.append(dict(kernel=name, launches_per_step=n / n_inst, avg_ms=ms, flop=fl,
                              tflops=fl / (ms * 1e-3) / 1e12, frac_of_f32_mfma_peak=fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                              ms_per_step=ms * n / n_inst))
        source = "HIP events on the launch stream around every launch of this kernel in %d instrumented train steps" % n_inst
        if not timed and kernels:
            timed = [r for r in kernels if r["launches_per_step"]]
            source = "HIP events on the launch stream, kernel launched back to back at the layer shape"
        if timed:
            dom = max(timed, key=lambda r: r["ms_per_step"])
            # HBM bytes per launch: not measurable from inside the process -- taken from the newest committed
            # rocprofv3 PMC digest (same kernel, same batch), and labelled so
            traffic, traffic_source = None, None
            pmc, pmc_path = latest_pmc_digest()
            if pmc and pmc.get("batch") == args.batch and dom["kernel"] in pmc.get("kernels", {}):
                traffic = pmc["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
                traffic_source = f"{pmc_path} (committed rocprofv3 --pmc passes of this bench command; not measured in this run)"
            line["roofline"] = {
                "kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"],
                "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac_of_f32_mfma_peak"],
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": dom["avg_ms"], "flop_per_launch": dom["flop"], "measured": source,
                # BASELINE's metric is "attn+VQ kernel %roofline": every hand-written attention / VQ kernel of the step,
                # same measurement, so that the fractions survive in the parsed line and not only in extra keys
                "kernels": [dict(kernel=r["kernel"], frac=r["frac_of_f32_mfma_peak"], achieved=r["tflops"],
                                 avg_launch_ms=r["avg_ms"], launches_per_step=r["launches_per_step"]) for r in timed],
            }
            # the same fractions as flat scalars (a nested list does not survive every parser): BASELINE's metric names
            # "attn+VQ kernel %roofline", so each of them is a key of `roofline` itself
            by = {r["kernel"]: r for r in timed}
            flat = {"frac_attn_fwd": "attn_fwd_kernel", "frac_attn_fwd_keep": "attn_fwd_keep_kernel", "frac_vq": "vq_lookup_fwd"}
            for key, name in flat.items():
                if name in by:
                    line["roofline"][key] = by[name]["frac_of_f32_mfma_peak"]
            bw = [r for n, r in by.items() if n.startswith("attn_bwd")]
            if bw:
                line["roofline"]["frac_attn_bwd"] = max(bw, key=lambda r: r["ms_per_step"])["frac_of_f32_mfma_peak"]
            if traffic:
                algo_bytes = 16.0 * args.batch * VIT["n_heads"] * VIT["d_head"] * 2 * T_   # 16*B*h*d*(I+J)
                line["roofline"]["algorithmic_bytes"] = algo_bytes
                line["roofline"]["traffic_ratio_vs_algorithmic"] = traffic / algo_bytes
            line["kernels_in_step"] = timed
        if kernels:
            for key, name in (("frac_attn_fwd_back_to_back", "attn_fwd_kernel"), ("frac_attn_fwd_keep_back_to_back", "attn_fwd_keep_kernel"),
                              ("frac_attn_bwd_back_to_back", "attn_bwd_fused_kernel(kept scores)" if amk_ops.ATTENTION_KEEP_SCORES else "attn_bwd_fused_kernel"),
                              ("frac_vq_back_to_back", "vq_lookup_fwd (prep+argmin+finalize)")):
                hit = [r for r in kernels if r["kernel"] == name]
                if hit and "roofline" in line:
                    line["roofline"][key] = hit[0]["frac_of_f32_mfma_peak"]
        v16 = variants.get("variant_bf16_autocast")
        if v16:
            # the reference's shipped precision as top-level scalars (never `value`: the headline is f32)
            line["bf16_images_per_s"] = v16["value"]
            line["bf16_ms_per_step"] = v16["ms_per_step"]
            ks = v16.get("roofline_bf16", {}).get("kernels", [])
            if len(ks) >= 2:
                line["bf16_frac_attn_fwd"], line["bf16_frac_attn_bwd"] = ks[0]["frac"], ks[1]["frac"]
        line.update(variants)
        if kernels:
            line["kernels_microbench"] = kernels
        if vitmoe:
            line["kernels_vitmoe"] = vitmoe
        if agent:
            line["kernels_agent"] = agent
        if dp_one:
            line["dp_step_rccl_world_of_one"] = dp_one
            if "graph" in dp_one and "ms_per_step" in dp_one.get("graph", {}):
                line["dp_captured_ms_per_step"] = dp_one["graph"]["ms_per_step"]
                line["dp_captured_host_enqueue_ms"] = dp_one["graph"]["host_enqueue_ms"]
        if cpu:
            line["cpu_baseline"] = cpu
        emit(line)
    if world > 1 or dist.is_initialized():
        # teardown in a defined order: rank 0's post-headline work (kernel micro-benchmarks, the JSON line) is done
        # before any rank leaves the group -- the other ranks wait here instead of tearing RCCL down under it
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
