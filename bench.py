"""ViT-VQGAN 256px training-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
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

    H, T, D = VIT["n_heads"], (VIT["img_size"] // VIT["patch_size"]) ** 2, VIT["d_head"]
    g = torch.Generator().manual_seed(99)
    mk = lambda: torch.randn(B, T, H, D, generator=g).to(dev).permute(0, 2, 1, 3)  # (B,T,h*d) storage
    q, k, v, d_o = mk(), mk(), mk(), mk()
    scale = D ** -0.5
    q, k, v, o, stats = ops._attn_forward(q, k, v, None, None, scale)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    delta = torch.empty(B, H, T, device=dev)
    bwd = lambda st: ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=st, delta=delta)
    bwd(1)
    core = 4.0 * B * H * T * T * D  # algorithmic FLOP of the forward
    layers_f, layers_b = 4 * VIT["depth"], 2 * VIT["depth"]  # per step: 2 model fwd + 1 bwd, enc+dec
    out = []
    t = time_launches(lambda: ops._attn_forward(q, k, v, None, None, scale), iters)
    out.append(dict(kernel="attn_fwd_kernel", launches_per_step=layers_f, avg_ms=t * 1e3, flop=core,
                    note="4*B*h*I*J*d"))
    old_mode, ops.ATTENTION_FORWARD = ops.ATTENTION_FORWARD, "bf16x6"
    t = time_launches(lambda: ops._attn_forward(q, k, v, None, None, scale), iters)
    ops.ATTENTION_FORWARD = old_mode
    out.append(dict(kernel="attn_fwd_x6 (split pre-pass + kernel)", launches_per_step=0, avg_ms=t * 1e3, flop=core,
                    note="the forward with split-bf16 products (amk_attn_fwd_x6); its bound is the bf16 MFMA peak / 6, "
                         "not the f32 MFMA peak the fraction below is taken against"))
    t = time_launches(lambda: bwd(8), iters)
    out.append(dict(kernel="attn_bwd_fused_kernel", launches_per_step=layers_b, avg_ms=t * 1e3, flop=2 * core,
                    note="8*B*h*I*J*d (dV,dP,dQ,dK; recomputed S not credited); includes the dq memset"))
    t = time_launches(lambda: bwd(2), iters)
    out.append(dict(kernel="attn_bwd_dkdv_kernel", launches_per_step=0, avg_ms=t * 1e3, flop=core,
                    note="credited dV,dK products: 4*B*h*I*J*d (recomputed S, dP not credited)"))
    t = time_launches(lambda: bwd(4), iters)
    out.append(dict(kernel="attn_bwd_dq_kernel", launches_per_step=0, avg_ms=t * 1e3, flop=core,
                    note="credited dP,dQ products: 4*B*h*I*J*d (recomputed S not credited)"))
    N, K, C = B * T, CODEBOOK["codebook_size"], CODEBOOK["codebook_dim"]
    z = torch.randn(N, C, generator=g).to(dev)
    E = torch.randn(K, C, generator=g).to(dev)
    t = time_launches(lambda: ops.vq_lookup(z, E, 0.25), iters)
    out.append(dict(kernel="vq_lookup_fwd (prep+argmin+finalize)", launches_per_step=2, avg_ms=t * 1e3,
                    flop=2.0 * N * K * C, note="2*N*K*C"))
    for r in out:
        r["tflops"] = r["flop"] / (r["avg_ms"] * 1e-3) / 1e12
        r["frac_of_f32_mfma_peak"] = r["tflops"] / F32_MFMA_PEAK_TFLOPS
        r["ms_per_step"] = r["avg_ms"] * r["launches_per_step"]
    return out


def main():
    args = parse()
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
    trainer = VQGANTrainStep(model, discr)
    g = torch.Generator().manual_seed(1234 + rank)
    imgs = torch.rand(args.batch, 3, VIT["img_size"], VIT["img_size"], generator=g).to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"rank {rank}/{world}: model built, batch {args.batch}/GPU")
    for _ in range(args.warmup):
        trainer.step(imgs)
    sync()
    note("warm-up done")
    from amk import ops as amk_ops
    if rank == 0:
        amk_ops.KERNEL_EVENTS = {}  # HIP events around every hot-path launch of the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = trainer.step(imgs)
    sync()
    dt = time.perf_counter() - t0
    in_situ = amk_ops.kernel_event_summary(amk_ops.KERNEL_EVENTS) if rank == 0 else {}
    amk_ops.KERNEL_EVENTS = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = float(logs["loss"])
    note(f"timed {args.steps} steps in {dt:.3f}s")

    # informational variant (never `value`): the same step with the generator forward shared by the
    # two phases (VQGANTrainStep(share_forward=True); identical numbers while dropout is 0)
    shared = None
    if world == 1 and not args.no_variants:
        trainer.share_forward = True
        for _ in range(2):
            trainer.step(imgs)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(imgs)
        sync()
        ds = time.perf_counter() - t1
        trainer.share_forward = False
        shared = {"value": args.batch * args.steps / ds, "unit": "images/s", "ms_per_step": ds / args.steps * 1e3,
                  "what": "generator forward run once per step and shared by the discriminator and generator "
                          "phases (the reference runs it twice on unchanged weights); not the headline value"}
        note("shared-forward variant done")
    # informational variant (never `value`): the reference-shaped step with the attention FORWARD on split-bf16
    # products (amk_attn_fwd_x6: f32 operands, six exact bf16 partial products per product, f32 accumulation;
    # measured error below the f32 MFMA path's).  The headline keeps the exact-f32 MFMA forward.
    x6 = None
    if world == 1 and not args.no_variants:
        prev_mode, amk_ops.ATTENTION_FORWARD = amk_ops.ATTENTION_FORWARD, "bf16x6"
        for _ in range(2):
            trainer.step(imgs)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(imgs)
        sync()
        dx = time.perf_counter() - t1
        amk_ops.ATTENTION_FORWARD = prev_mode
        x6 = {"value": args.batch * args.steps / dx, "unit": "images/s", "ms_per_step": dx / args.steps * 1e3,
              "what": "attention forward with split-bf16 (bf16x6) products, f32-level error; backward and everything "
                      "else as in the headline; not the headline value"}
        note("split-bf16 forward variant done")

    kernels = None
    if rank == 0 and not args.no_kernels:
        kernels = kernel_rooflines(args.batch, dev, args.kernel_iters)
        note("kernel rooflines done")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import train_step_cpu

        ips, cores, sample = train_step_cpu.time_train_step(init_state, VIT, NLayerDiscriminator(3, 64, 3))
        cpu = dict(value=ips, unit="images/s", cores=cores, kind="port", sample=sample)
        note("cpu baseline done")

    if rank == 0:
        global_batch = args.batch * world
        line = {
            "metric": "ViTVQGAN 256px train-step images/sec",
            "value": global_batch * args.steps / dt,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic U[0,1) images, random-init weights",
            "config": {
                "workload": "BASELINE.json configs[2]: ViTVQGAN dim=256 patch=8 img=256 depth=6+6 h=8 d=64 "
                            "mlp=2048 (SwiGLU), codebook 8192x32; GAN train step (D phase + G phase, "
                            "gradient penalty, per_loss_weight=0), Adam, clip 1.0",
                "global_batch": global_batch,
                "batch_per_gpu": args.batch,
                "tokens_per_image": (VIT["img_size"] // VIT["patch_size"]) ** 2,
                "parallelism": f"dp{world}",
                "final_loss": loss,
            },
        }
        # in-situ figures: HIP events on the launch stream, over the timed steps themselves
        T_ = (VIT["img_size"] // VIT["patch_size"]) ** 2
        core = 4.0 * args.batch * VIT["n_heads"] * T_ * T_ * VIT["d_head"]
        flop_of = {"attn_fwd_kernel": core, "attn_bwd_fused_kernel": 2 * core, "attn_bwd_dkdv+dq": 2 * core,
                   "vq_lookup_fwd": 2.0 * args.batch * T_ * CODEBOOK["codebook_size"] * CODEBOOK["codebook_dim"]}
        timed = []
        for name, (n, ms) in in_situ.items():
            fl = flop_of.get(name)
            timed.append(dict(kernel=name, launches_per_step=n / args.steps, avg_ms=ms, flop=fl,
                              tflops=fl / (ms * 1e-3) / 1e12, frac_of_f32_mfma_peak=fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                              ms_per_step=ms * n / args.steps))
        if timed:
            dom = max(timed, key=lambda r: r["ms_per_step"])
            traffic = None  # HBM bytes per launch from the committed PMC passes (same kernel, same batch)
            pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_kernels_b32.json")
            if os.path.exists(pmc_path):
                pmc = json.load(open(pmc_path))
                if pmc.get("batch") == args.batch and dom["kernel"] in pmc["kernels"]:
                    traffic = pmc["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
            line["roofline"] = {
                "kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"],
                "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac_of_f32_mfma_peak"],
                "traffic": traffic, "avg_launch_ms": dom["avg_ms"], "flop_per_launch": dom["flop"],
                "measured": "HIP events on the launch stream around every launch of this kernel in the timed steps",
            }
            line["kernels_in_step"] = timed
        if kernels and not timed:
            dom = max(kernels[:4], key=lambda r: r["ms_per_step"])
            traffic = None  # HBM bytes per launch from the committed PMC passes (same kernel, same batch)
            pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_kernels_b32.json")
            if os.path.exists(pmc_path):
                pmc = json.load(open(pmc_path))
                if pmc.get("batch") == args.batch and dom["kernel"] in pmc["kernels"]:
                    traffic = pmc["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
            line["roofline"] = {
                "kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"],
                "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac_of_f32_mfma_peak"],
                "traffic": traffic, "avg_launch_ms": dom["avg_ms"], "flop_per_launch": dom["flop"],
            }
        if shared:
            line["variant_shared_generator_forward"] = shared
        if x6:
            line["variant_split_bf16_attention_forward"] = x6
        if kernels:
            line["kernels_microbench"] = kernels
        if cpu:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
