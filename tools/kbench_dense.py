"""The nn.Linear GEMM shapes of the ViT-VQGAN step (batch 32: M = 32768 rows): vendor exact-f32 GEMM (TunableOp
selection) against amk_gemm_f32 (csrc/gemm_f32.hip) -- forward (NT), input gradient (NN), weight gradient (TN) --
and the fused forms against the launches they replace.
    python tools/kbench_dense.py [--batch 32] [--iters 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from bench import time_launches  # noqa: E402

PEAK = 157.3


CHECK = True


def err(x, ref):
    if not CHECK:
        return 0.0
    ref = ref() if callable(ref) else ref
    return float((x.double() - ref).abs().max() / ref.abs().max())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="comma-separated shape names")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the float64 error columns (PMC runs)")
    a = ap.parse_args()
    CHECK = not a.no_check
    from amk import dense, ops, tuning

    tuning.enable_gemm_tuning()
    dev = torch.device("cuda:0")
    M = a.batch * 1024
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = [("q", 256, 512), ("kv", 256, 1024), ("W_o", 512, 256), ("ffn w12", 256, 2736), ("ffn w3", 1368, 256),
              ("patch embed", 192, 256), ("pre_quant", 256, 32), ("post_quant", 32, 256), ("fc", 256, 192)]
    tot = dict(lib_nt=0.0, amk_nt=0.0, lib_nn=0.0, amk_nn=0.0, lib_tn=0.0, amk_tn=0.0)
    if a.only:
        shapes = [s for s in shapes if s[0].replace(" ", "_") in a.only.split(",")]
    for name, K, N in shapes:
        x, w, b, dy = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N), rnd(M, N)
        fl = 2.0 * M * N * K
        # forward
        t_lib = time_launches(lambda: F.linear(x, w, b), a.iters)
        t_amk = time_launches(lambda: dense.gemm_nt(x, w, b), a.iters)
        ref = (x.double() @ w.double().t() + b.double()) if CHECK else None
        e_lib, e_amk = err(F.linear(x, w, b), ref), err(dense.gemm_nt(x, w, b), ref)
        tot["lib_nt"] += t_lib; tot["amk_nt"] += t_amk
        print(f"NT {name:12s} K{K:5d} N{N:5d}  lib {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.1f} TF err {e_lib:.1e} | "
              f"amk {t_amk*1e6:7.1f} us {fl/t_amk/1e12:6.1f} TF ({fl/t_amk/1e12/PEAK:.3f}) err {e_amk:.1e}  x{t_lib/t_amk:.2f}", flush=True)
        # input gradient
        if N % 4 == 0:
            t_lib = time_launches(lambda: dy.mm(w), a.iters)
            t_amk = time_launches(lambda: dense.gemm_nn(dy, w), a.iters)
            ref = (dy.double() @ w.double()) if CHECK else None
            e_lib, e_amk = err(dy.mm(w), ref), err(dense.gemm_nn(dy, w), ref)
            tot["lib_nn"] += t_lib; tot["amk_nn"] += t_amk
            print(f"NN {name:12s} K{N:5d} N{K:5d}  lib {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.1f} TF err {e_lib:.1e} | "
                  f"amk {t_amk*1e6:7.1f} us {fl/t_amk/1e12:6.1f} TF ({fl/t_amk/1e12/PEAK:.3f}) err {e_amk:.1e}  x{t_lib/t_amk:.2f}", flush=True)
            # weight gradient + bias gradient
            t_lib = time_launches(lambda: (dy.t().mm(x), ops.colsum(dy)), a.iters)
            t_amk = time_launches(lambda: dense.gemm_tn(dy, x, want_bias=True), a.iters)
            ref = (dy.double().t() @ x.double()) if CHECK else None
            dw, _, db = dense.gemm_tn(dy, x, want_bias=True)
            e_lib, e_amk = err(dy.t().mm(x), ref), err(dw, ref)
            e_db = err(db, lambda: dy.double().sum(0))
            tot["lib_tn"] += t_lib; tot["amk_tn"] += t_amk
            print(f"TN {name:12s} M{M} -> {N}x{K}  lib+colsum {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.1f} TF err {e_lib:.1e} | "
                  f"amk {t_amk*1e6:7.1f} us {fl/t_amk/1e12:6.1f} TF ({fl/t_amk/1e12/PEAK:.3f}) err {e_amk:.1e} db {e_db:.1e}  x{t_lib/t_amk:.2f}", flush=True)
    print("sums (ms): " + "  ".join(f"{k} {v*1e3:.3f}" for k, v in tot.items()), flush=True)

    if a.no_fused:
        sys.exit(0)
    # ---- fused forms against the launches they replace (one encoder layer's Linear + element-wise work)
    D, Hh, inner = 256, 1368, 512
    h = rnd(M, D)
    gam, bet = 1 + 0.1 * rnd(D), 0.1 * rnd(D)
    wq, wkv = rnd(inner, D) * D ** -0.5, rnd(2 * inner, D) * D ** -0.5
    w12, b12 = rnd(2 * Hh, D) * D ** -0.5, rnd(2 * Hh)
    w3, b3 = rnd(D, Hh) * Hh ** -0.5, rnd(D)
    wo, bo = rnd(D, inner) * inner ** -0.5, rnd(D)
    o = rnd(M, inner)

    def old_qkv():
        y = ops.layer_norm(h, gam, bet)
        return F.linear(y, wq), F.linear(y, wkv)

    def new_qkv():
        mean, rstd = dense.row_stats(h)
        return dense.gemm_nt(h, wq, w2=wkv, ln=(mean, rstd, gam, bet))

    def old_ffn1():
        return ops.swiglu(F.linear(ops.layer_norm(h, gam, bet), w12, b12))

    def new_ffn1(keep=True):
        mean, rstd = dense.row_stats(h)
        return dense.gemm_nt_swiglu(h, w12, b12, ln=(mean, rstd, gam, bet), keep_ab=keep)

    def old_res(x, w, b):
        return F.linear(x, w, b) + h

    def new_res(x, w, b):
        return dense.gemm_nt(x, w, b, resid=h)

    q_ref, kv_ref = old_qkv()
    q_new, kv_new = new_qkv()
    g_ref = old_ffn1()
    g_new, ab_new = new_ffn1()
    gate = g_ref
    print(f"fused LN+q|kv  err {err(q_new, q_ref.double()):.1e} {err(kv_new, kv_ref.double()):.1e}; "
          f"LN+w12+SwiGLU err {err(g_new, g_ref.double()):.1e}; "
          f"W_o+resid err {err(new_res(o, wo, bo), old_res(o, wo, bo).double()):.1e}; "
          f"w3+resid err {err(new_res(gate, w3, b3), old_res(gate, w3, b3).double()):.1e}")
    rows = [("LN + q + kv", old_qkv, new_qkv), ("LN + w12 + SwiGLU (keeps a|b)", old_ffn1, new_ffn1),
            ("LN + w12 + SwiGLU (no a|b)", old_ffn1, lambda: new_ffn1(False)),
            ("W_o + residual", lambda: old_res(o, wo, bo), lambda: new_res(o, wo, bo)),
            ("w3 + residual", lambda: old_res(gate, w3, b3), lambda: new_res(gate, w3, b3))]
    for name, fo, fn in rows:
        to, tn = time_launches(fo, a.iters), time_launches(fn, a.iters)
        print(f"{name:32s} separate launches {to*1e6:7.1f} us | fused {tn*1e6:7.1f} us  x{to/tn:.2f}", flush=True)
    # backward pieces
    dg = rnd(M, D)
    ab = ab_new

    def old_ffn_bwd():
        dgate = dg.mm(w3)
        d_ab = torch.empty_like(ab)
        L = __import__("amk.lib", fromlist=["x"]).load()
        import ctypes
        L.amk_swiglu_bwd(ctypes.c_void_p(ab.data_ptr()), ctypes.c_void_p(dgate.data_ptr()), M, Hh, ctypes.c_void_p(d_ab.data_ptr()),
                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return d_ab

    def new_ffn_bwd():
        return dense.gemm_nn(dg, w3, swiglu_ab=ab)

    print(f"dGate + SwiGLU' err {err(new_ffn_bwd(), old_ffn_bwd().double()):.1e}")
    to, tn = time_launches(old_ffn_bwd, a.iters), time_launches(new_ffn_bwd, a.iters)
    print(f"{'dY W3 + SwiGLU backward':32s} separate launches {to*1e6:7.1f} us | fused {tn*1e6:7.1f} us  x{to/tn:.2f}", flush=True)
    dq, dkv = rnd(M, inner), rnd(M, 2 * inner)
    old2 = lambda: dq.mm(wq) + dkv.mm(wkv)
    new2 = lambda: dense.gemm_nn(dq, wq, a2=dkv, w2=wkv)
    print(f"dq Wq + dkv Wkv err {err(new2(), (dq.double() @ wq.double() + dkv.double() @ wkv.double())):.1e}")
    to, tn = time_launches(old2, a.iters), time_launches(new2, a.iters)
    print(f"{'dq Wq + dkv Wkv':32s} separate launches {to*1e6:7.1f} us | fused {tn*1e6:7.1f} us  x{to/tn:.2f}", flush=True)
    mean, rstd = dense.row_stats(h)
    y = ops.layer_norm(h, gam, bet)
    old3 = lambda: (dq.t().mm(y), dkv.t().mm(y))
    new3 = lambda: dense.gemm_tn(dq, h, y2=dkv, ln=(mean, rstd, gam, bet))
    r = new3()
    print(f"dWq | dWkv (LN recomputed) err {err(r[0], dq.double().t() @ y.double()):.1e} {err(r[1], dkv.double().t() @ y.double()):.1e}")
    to, tn = time_launches(old3, a.iters), time_launches(new3, a.iters)
    print(f"{'dWq | dWkv':32s} separate launches {to*1e6:7.1f} us | fused {tn*1e6:7.1f} us  x{to/tn:.2f}", flush=True)
