"""Probe for the 'dense Z' form of SwitchHead's output experts at the ViTMoE layer: out = Z (G, E*d) @ Wcat (E*d, N)
against the routed GEMM + combine it would replace (0.147 + 0.074 ms).  Also the projection GEMMs of the layer
(M 4160, N 1536, K 1024) on own kernels against the library."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch
from bench import time_launches
from amk import dense, tuning
tuning.enable_gemm_tuning()
dev = torch.device("cuda:0")
G, E, d, N = 4160, 32, 64, 1024
Z = torch.randn(G, E * d, device=dev)
Wout = torch.randn(E, N, d, device=dev)
Wv = torch.randn(E, d, N, device=dev)
def fl(t, f): return f"{t*1e6:7.1f} us  {f/t/1e12:6.1f} TFLOP/s"
F = 2.0 * G * N * E * d
t = time_launches(lambda: Wout.permute(0, 2, 1).reshape(E * d, N).contiguous(), 20); print("Wout -> (E*d, N) copy", f"{t*1e6:.1f} us")
t = time_launches(lambda: Wout.permute(1, 0, 2).reshape(N, E * d).contiguous(), 20); print("Wout -> (N, E*d) copy", f"{t*1e6:.1f} us")
Wt = Wout.permute(0, 2, 1).reshape(E * d, N).contiguous()
Wn = Wout.permute(1, 0, 2).reshape(N, E * d).contiguous()
t = time_launches(lambda: torch.mm(Z, Wt), 20); print("lib  NN  Z @ Wt      ", fl(t, F))
t = time_launches(lambda: torch.nn.functional.linear(Z, Wn), 20); print("lib  NT  Z @ Wn^T    ", fl(t, F))
t = time_launches(lambda: dense.gemm_nn(Z, Wt), 20); print("own  NN  Z @ Wt      ", fl(t, F))
t = time_launches(lambda: dense.gemm_nt(Z, Wn), 20); print("own  NT  Z @ Wn^T    ", fl(t, F))
Wv2 = Wv.view(E * d, N)
t = time_launches(lambda: torch.mm(Z, Wv2), 20); print("lib  NN  Z' @ Wv      ", fl(t, F))
t = time_launches(lambda: dense.gemm_nn(Z, Wv2), 20); print("own  NN  Z' @ Wv      ", fl(t, F))
# projection of the layer
M, Np, K = 4160, 1536, 1024
x = torch.randn(M, K, device=dev); W = torch.randn(Np, K, device=dev); dy = torch.randn(M, Np, device=dev)
Fp = 2.0 * M * Np * K
t = time_launches(lambda: torch.nn.functional.linear(x, W), 20); print("proj lib NT", fl(t, Fp))
t = time_launches(lambda: dense.gemm_nt(x, W), 20); print("proj own NT", fl(t, Fp))
t = time_launches(lambda: torch.mm(dy, W), 20); print("proj lib NN", fl(t, Fp))
t = time_launches(lambda: dense.gemm_nn(dy, W), 20); print("proj own NN", fl(t, Fp))
t = time_launches(lambda: torch.mm(dy.t(), x), 20); print("proj lib TN", fl(t, Fp))
t = time_launches(lambda: dense.gemm_tn(dy, x), 20); print("proj own TN", fl(t, Fp))
