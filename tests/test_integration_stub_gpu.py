"""The ctypes stub printed in INTEGRATION.md section 2, executed as written: binding the C ABI from
outside amk.lib / amk.ops gives the same attention output as the packaged wrapper."""
import ctypes
import os

import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded
from util import assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub(device):
    _amk = ctypes.CDLL(os.path.join(ROOT, "attention-models_amd", "amk", "libamk.so"))
    _P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    _amk.amk_attn_fwd.restype = _I
    _amk.amk_attn_fwd.argtypes = [_P] * 7 + [_I] * 5 + [_L] * 12 + [_F, _P]
    _amk.amk_last_error.restype = ctypes.c_char_p

    def attn_core(q, k, v, scale, key_mask_u8=None, causal_u8=None):      # (B,h,T,d) fp32, last axis contiguous
        B, H, I, D = q.shape
        J = k.shape[2]
        o = torch.empty(B, I, H, D, device=q.device).permute(0, 2, 1, 3)   # (B,I,h*d) storage for W_o
        stats = torch.empty(B, H, I, 2, device=q.device)
        s = lambda t: (t.stride(0), t.stride(2), t.stride(1))              # (sb, st, sh) in elements
        p = lambda t: _P(t.data_ptr()) if t is not None else _P(0)
        rc = _amk.amk_attn_fwd(p(q), p(k), p(v), p(o), p(stats), p(key_mask_u8), p(causal_u8),
                               B, H, I, J, D, *s(q), *s(k), *s(v), *s(o), scale,
                               _P(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(_amk.amk_last_error())
        return o, stats

    B, H, I, J, D = 2, 3, 70, 77, 64
    q, k, v = seeded((B, H, I, D), 1), seeded((B, H, J, D), 2), seeded((B, H, J, D), 3)
    km = torch.ones(B, J, dtype=torch.bool)
    km[1, -9:] = False
    o, stats = attn_core(q.to(device), k.to(device), v.to(device), D ** -0.5, km.to(torch.uint8).to(device))
    assert_close(o, ref_cpu.attention_core(q, k, v, D ** -0.5, km), 2e-5, "stub output vs oracle")
    assert tuple(o.permute(0, 2, 1, 3).reshape(B, I, H * D).shape) == (B, I, H * D)   # no copy needed for W_o
    # error path: unsupported head dim comes back as a code + message, not an exception across the ABI
    q48 = torch.zeros(1, 1, 4, 48, device=device)   # head dims 32, 64 and 128 are built
    with pytest.raises(RuntimeError):
        attn_core(q48, q48, q48, 1.0)
