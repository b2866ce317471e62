"""Diagnostic: per-workgroup time stamps of the NT GEMM (build with -DAMK_DENSE_STAMPS, AMK_LIB=...libamk_stamps.so)."""
import ctypes, sys, os
sys.path.insert(0, "attention-models_amd")
import numpy as np
import torch
from amk import lib as amk_lib, dense
from amk.lib import GemmDesc

dev = torch.device("cuda:0")
M, K, N = 32768, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
L = amk_lib.load()
ntile = ((M + 127) // 128) * ((N + 127) // 128)
ws = torch.zeros(ntile * 8, dtype=torch.int64, device=dev)
d = GemmDesc(op=0, epilogue=0, m=M, n=N, k=K)
P = lambda t: ctypes.c_void_p(t.data_ptr())
d.a, d.lda, d.w, d.ldw, d.c, d.ldc = P(x), K, P(w), K, P(c), N
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    amk_lib.check(L.amk_gemm_f32(ctypes.byref(d), P(ws), ws.numel() * 8, st), "gemm")
torch.cuda.synchronize()
s = ws.cpu().numpy().reshape(ntile, 8)
t = s[:, :4].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0  # us (100 MHz)
print(f"tiles {ntile}; kernel span {t[:, 3].max():.1f} us")
print("prologue us: mean %.2f p50 %.2f p95 %.2f" % ((t[:, 1] - t[:, 0]).mean(), np.median(t[:, 1] - t[:, 0]), np.percentile(t[:, 1] - t[:, 0], 95)))
print("main loop us: mean %.2f p50 %.2f p95 %.2f" % ((t[:, 2] - t[:, 1]).mean(), np.median(t[:, 2] - t[:, 1]), np.percentile(t[:, 2] - t[:, 1], 95)))
print("epilogue us: mean %.2f p50 %.2f p95 %.2f" % ((t[:, 3] - t[:, 2]).mean(), np.median(t[:, 3] - t[:, 2]), np.percentile(t[:, 3] - t[:, 2], 95)))
hw, xcc = s[:, 6], s[:, 7]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (xcc & 0xF) * 1000 + se * 100 + sh * 20 + cu
print("distinct CU keys", len(set(key.tolist())))
# concurrency: for one CU, list the workgroups in start order
k0 = key[0]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(t[idx, 0])]
for i in idx[:12]:
    print("  wg %5d start %7.2f loop %7.2f..%7.2f end %7.2f" % (i, t[i, 0], t[i, 1], t[i, 2], t[i, 3]))
# average number of resident workgroups per CU over time
ev = []
for kk in set(key.tolist()):
    ii = np.where(key == kk)[0]
    for i in ii:
        ev.append((t[i, 0], 1, kk)); ev.append((t[i, 3], -1, kk))
tot_busy = sum(t[:, 3] - t[:, 0])
print("mean resident workgroups per CU while the kernel runs: %.2f" % (tot_busy / (t[:, 3].max() * len(set(key.tolist())))))
# start-time histogram of first-round workgroups
first = np.sort(t[:, 0])[: 512]
print("first 512 starts: min %.2f max %.2f" % (first.min(), first.max()))
