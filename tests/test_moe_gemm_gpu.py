"""The three grouped expert GEMMs of libamk.so called through the C ABI on edge shapes -- output widths that are
not multiples of the 128-wide tiles, more than 64 experts (the unit decode walks the experts 64 at a time),
experts without pairs, row divisors that are not powers of two, with and without scale / bias -- against a float64
loop over the pairs (what the reference's per-expert Python loop computes: models/moe.py:27-36,
models/switchhead_attention.py:63-88)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

# (R units, k, E, N, Kd, a_div, x_div, empty experts)
CASES = [
    (130, 2, 4, 256, 128, 2, 2, 0),     # the wide kernels' plain case
    (260, 2, 32, 192, 160, 2, 2, 0),    # widths that end inside a 128-wide tile
    (97, 2, 70, 128, 256, 2, 2, 0),     # more than 64 experts
    (300, 2, 6, 128, 128, 2, 2, 3),     # experts without pairs
    (64, 3, 5, 256, 256, 3, 3, 0),      # divisor that is not a power of two
    (520, 2, 8, 64, 1024, 16, 2, 0),    # SwitchHead V experts: 64 outputs from the model width
    (520, 2, 8, 1024, 64, 1, 1, 0),     # SwitchHead output experts: K = 64
    (9, 1, 3, 128, 96, 1, 1, 0),        # depth that is a multiple of 32 only
    (2000, 2, 2, 128, 128, 2, 2, 0),    # many rows per expert (tiles of four blocks, several rounds)
    (1500, 2, 4, 64, 256, 2, 2, 1),     # few output tiles, long experts: the weight gradient's two-workgroup form, one expert empty
    (5000, 1, 16, 256, 64, 1, 1, -1),   # the same with skewed routing: experts of a handful of pairs next to long ones
]


def _route(R, k, E, empty, dev, seed):
    from amk import ops

    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(R, E, generator=g)
    if empty > 0:
        logits[:, :empty] = -1e4  # never chosen
    elif empty < 0:
        logits += torch.linspace(-7.0, 3.0, E)  # skewed: the first experts get a few pairs, the last ones most
    return ops.moe_route(logits.to(dev), k)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "R{}k{}E{}N{}K{}a{}x{}z{}".format(*c))
def test_grouped_gemms_vs_pair_loop(device, case):
    from amk import lib as L_

    R, k, E, N, Kd, a_div, x_div, empty = case
    L = L_.load()
    dev = device
    r = _route(R, k, E, empty, dev, 5)
    P = R * k
    g = torch.Generator().manual_seed(6)
    rows_a = (P - 1) // a_div + 1
    rows_x = (P - 1) // x_div + 1
    A = torch.randn(rows_a, Kd, generator=g).to(dev)          # nt input (rows addressed p / a_div)
    W = torch.randn(E, N, Kd, generator=g).to(dev)
    bias = torch.randn(E, N, generator=g).to(dev)
    Gm = torch.randn(rows_a, N, generator=g).to(dev)          # nn / wgrad input
    X = torch.randn(rows_x, Kd, generator=g).to(dev)
    scale = r["gate"].reshape(-1).contiguous()
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ids = r["ids"].reshape(-1).cpu()
    sc = scale.cpu().double()
    Ac, Wc, bc, Gc, Xc = (t.cpu().double() for t in (A, W, bias, Gm, X))

    def close(got, want, what):
        err = float((got.cpu().double() - want).abs().max())
        ref = max(float(want.abs().max()), 1e-6)
        assert err <= 2e-5 * ref, f"{what}: abs err {err:.3e} at scale {ref:.3e}"

    # forward: Y[p] = A[p / a_div] W_e^T + b_e
    Y = torch.full((P, N), float("nan"), device=dev)
    L_.check(L.amk_grouped_gemm_nt(ptr(A), Kd, a_div, ptr(W), ptr(bias), ptr(r["offsets"]), ptr(r["perm"]), P, E, N, Kd, ptr(Y), st),
             "amk_grouped_gemm_nt")
    want = torch.stack([Ac[p // a_div] @ Wc[int(ids[p])].t() + bc[int(ids[p])] for p in range(P)])
    close(Y, want, "grouped_nt")

    # input gradient: Y[p] = s[p] * G[p / a_div] W_e
    Y2 = torch.full((P, Kd), float("nan"), device=dev)
    L_.check(L.amk_grouped_gemm_nn(ptr(Gm), N, a_div, ptr(W), ptr(scale), ptr(r["offsets"]), ptr(r["perm"]), P, E, N, Kd, ptr(Y2), st),
             "amk_grouped_gemm_nn")
    want = torch.stack([sc[p] * (Gc[p // a_div] @ Wc[int(ids[p])]) for p in range(P)])
    close(Y2, want, "grouped_nn")

    # the accumulating forms: pair p adds into row p / y_div of a zeroed output (wide shapes only)
    for y_div in (k, 4 * k):
        if N >= 128 and Kd % 32 == 0:
            Ya = torch.zeros(((P - 1) // y_div + 1, N), device=dev)
            L_.check(L.amk_grouped_gemm_nt_acc(ptr(A), Kd, a_div, ptr(W), ptr(bias), ptr(r["offsets"]), ptr(r["perm"]), P, E, N, Kd,
                                               ptr(Ya), y_div, st), "amk_grouped_gemm_nt_acc")
            want_a = torch.zeros(Ya.shape, dtype=torch.float64)
            want_a.index_add_(0, torch.arange(P) // y_div, torch.stack([Ac[p // a_div] @ Wc[int(ids[p])].t() + bc[int(ids[p])] for p in range(P)]))
            close(Ya, want_a, f"grouped_nt_acc y_div={y_div}")
        if Kd >= 128 and N % 32 == 0:
            Yb = torch.zeros(((P - 1) // y_div + 1, Kd), device=dev)
            L_.check(L.amk_grouped_gemm_nn_acc(ptr(Gm), N, a_div, ptr(W), ptr(scale), ptr(r["offsets"]), ptr(r["perm"]), P, E, N, Kd,
                                               ptr(Yb), y_div, st), "amk_grouped_gemm_nn_acc")
            want_b2 = torch.zeros(Yb.shape, dtype=torch.float64)
            want_b2.index_add_(0, torch.arange(P) // y_div, want)
            close(Yb, want_b2, f"grouped_nn_acc y_div={y_div}")

    # weight gradient (with and without the scale)
    for use_scale in (True, False):
        dW = torch.full((E, N, Kd), float("nan"), device=dev)
        db = torch.full((E, N), float("nan"), device=dev)
        L_.check(L.amk_grouped_gemm_wgrad(ptr(Gm), N, a_div, ptr(X), Kd, x_div, ptr(scale) if use_scale else None, ptr(r["offsets"]),
                                          ptr(r["perm"]), P, E, N, Kd, ptr(dW), ptr(db), st), "amk_grouped_gemm_wgrad")
        want_w = torch.zeros(E, N, Kd, dtype=torch.float64)
        want_b = torch.zeros(E, N, dtype=torch.float64)
        for p in range(P):
            s = sc[p] if use_scale else 1.0
            want_w[int(ids[p])] += s * torch.outer(Gc[p // a_div], Xc[p // x_div])
            want_b[int(ids[p])] += s * Gc[p // a_div]
        close(dW, want_w, f"grouped_wgrad dW (scale={use_scale})")
        close(db, want_b, f"grouped_wgrad dbias (scale={use_scale})")
