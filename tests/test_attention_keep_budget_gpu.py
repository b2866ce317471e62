"""The account of kept attention scores (amk/ops.py: ATTENTION_KEEP_SCORES_BUDGET_BYTES): the forward keeps the raw
scores for the fused backward only while the total alive stays inside a budget, and recomputes beyond it -- same bits.
Reference: the autograd graph of models/softmax_attention.py:52-78 keeps the same tensor alive."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_kept_scores_respect_a_budget_across_layers(device, monkeypatch):
    """Kept scores are accounted across calls: once the live total would pass the budget the forward recomputes
    (same results), and the account is released when the autograd graph is freed."""
    from amk import ops

    B, H, T, D = 2, 2, 256, 64
    per_call = 4 * B * H * T * T
    monkeypatch.setattr(ops, "ATTENTION_KEEP_SCORES_BUDGET_BYTES", 2 * per_call + 1)
    monkeypatch.setattr(ops, "DETERMINISTIC_ATTENTION_BACKWARD", True)   # (dq by ordered sums: comparable bit for bit)
    monkeypatch.setattr(ops, "ATTENTION_KEEP_SCORES", True)
    monkeypatch.setattr(ops, "ATTENTION_BACKWARD_TWO_KERNEL", False)
    monkeypatch.setattr(ops, "ATTENTION_FORWARD", "f32")
    base = ops._kept_scores_bytes[0]
    q = torch.randn(B, T, H * D, device=device, requires_grad=True)
    kv = torch.randn(B, T, 2 * H * D, device=device, requires_grad=True)
    outs = [ops.attention_fused_kv(q, kv, H, D, D ** -0.5) for _ in range(4)]
    assert ops._kept_scores_bytes[0] - base == 2 * per_call        # two calls kept their scores, two recompute
    grads = [torch.autograd.grad(o.sum(), [q, kv], retain_graph=True) for o in outs]
    for i, g in enumerate(grads[1:], 1):
        if i < 2:
            assert torch.equal(g[0], grads[0][0]) and torch.equal(g[1], grads[0][1])   # kept twice: the same bits
        else:
            # recomputed: the same numbers to rounding (the forward that keeps its scores leaves them raw and subtracts the
            # row reference afterwards; the one that does not subtracts it inside the MFMA chain)
            for a, b in zip(g, grads[0]):
                assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    del outs, grads, g
    import gc
    gc.collect()
    assert ops._kept_scores_bytes[0] == base
