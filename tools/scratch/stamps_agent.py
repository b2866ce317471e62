"""Diagnostic: per-workgroup phase times of agent_s2_bwd_kernel (build agent.hip with -DAMK_AGENT_STAMPS into AMK_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "attention-models_amd"))
import numpy as np
import torch
from amk import lib, ops
from amk.models import AgentAttention

dev = torch.device("cuda:0")
B, h, d, T = 64, 6, 64, 1024
ag = AgentAttention(h * d, h, d).to(dev)
qkv = torch.randn(B, T, 3 * h * d, device=dev, requires_grad=True)
co = torch.randn(B, T, h * d, device=dev)
nwg = B * h * 8
ws = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
C = ctypes.CDLL(lib.LIB_PATH)
for it in range(3):
    o = ops.agent_attention(qkv, ag.dwc[1].weight, ag.dwc[1].bias, h, d, ag.pool_size, ag.scale)
    if it == 2:
        assert C.amk_debug_agent_set_stamps(ctypes.c_void_p(ws.data_ptr())) == 0
    torch.autograd.grad(o, [qkv], co)
torch.cuda.synchronize()
s = ws.cpu().numpy().reshape(nwg, 8)[:, :6].astype(np.float64)
s = (s - s[:, 0].min()) / 100.0
print("kernel span %.1f us" % s[:, 5].max())
names = ["q tile staged", "dO tile staged (+ q dots)", "phase A rest (softmax, dq rows)", "dq stores + phase B", "reductions + partial stores"]
for k, n in enumerate(names):
    dt = s[:, k + 1] - s[:, k]
    print("%-34s mean %6.2f p50 %6.2f p95 %6.2f us" % (n, dt.mean(), np.median(dt), np.percentile(dt, 95)))
life = s[:, 5] - s[:, 0]
print("workgroup life mean %.2f us; starts: first 512 within %.2f us" % (life.mean(), np.sort(s[:, 0])[511]))
