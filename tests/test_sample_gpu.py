"""The fused sampling step of the masked-token decode (csrc/sample.hip) and the MaskGit model around it,
against the oracle's restatement of the reference's op chain (models/muse.py:211-236,
models/maskgit.py:80-91,255-272).  Predicted ids must be identical (the Gumbel noise is passed to both
sides); probabilities within 1e-5."""
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded
from util import assert_close

pytestmark = pytest.mark.gpu


def _gumbel(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return -torch.empty(shape).exponential_(generator=g).log()   # F.gumbel_softmax's noise


@pytest.mark.parametrize("B,T,V", [(2, 16, 8192), (1, 7, 1000), (3, 5, 64), (1, 3, 36864)])
@pytest.mark.parametrize("cfg", [True, False])
@pytest.mark.parametrize("tau", [1.0, 1.0 / 18, 0.0])
def test_sample_step_matches_reference_chain(device, B, T, V, cfg, tau):
    from amk import ops

    logits = seeded((B, T, V), 1, 2.0)
    null = seeded((B, T, V), 2, 2.0) if cfg else None
    g = _gumbel((B, T, V), 3)
    ids = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(4))
    mask = torch.rand(B, T, generator=torch.Generator().manual_seed(5)) < 0.6
    unmasked = None if cfg else 1.0     # Muse keeps every probability, MaskGit writes 1.0 outside the mask
    want_ids, want_sc = ref_cpu.sampling_step(logits, ids, mask, g, tau, null_logits=null, unmasked_score=unmasked)
    got_ids = ids.clone().to(device)
    got_sc = ops.sample_step(logits.to(device), got_ids, mask=mask.to(device), null_logits=None if null is None else null.to(device),
                             cfg_scale=3.0, tau=tau, p=0.9, gumbel=g.to(device), unmasked_score=unmasked)
    assert torch.equal(got_ids.cpu(), want_ids)
    assert_close(got_sc, want_sc, 1e-5, "scores")


def test_sample_step_in_kernel_noise(device):
    """Without explicit noise the kernel draws Gumbel noise itself (Philox): predictions stay inside the kept
    top-k set, differ between calls, repeat under the same seed, and follow the softmax of the logits
    (Gumbel-max: P(pred = j) = softmax(s)_j over the kept set) -- a chi-square-free sanity bound on frequencies."""
    from amk import ops

    V, R = 64, 20000
    logits = seeded((1, 1, V), 7, 1.5).expand(1, R, V).contiguous().to(device)
    ids = torch.zeros(1, R, dtype=torch.long, device=device)
    ops.sample_step(logits, ids, tau=1.0, p=0.5, seed=123)
    keep = logits[0, 0].topk(32).indices
    assert bool(torch.isin(ids, keep).all())
    ids2 = torch.zeros_like(ids)
    ops.sample_step(logits, ids2, tau=1.0, p=0.5, seed=123)
    assert not torch.equal(ids, ids2)          # the call counter advances the stream
    # torch's generator drives the stream when no seed is given: manual_seed repeats the samples, and a call advances it
    a, b, c = (torch.zeros_like(ids) for _ in range(3))
    torch.manual_seed(77)
    ops.sample_step(logits, a, tau=1.0, p=0.5)
    ops.sample_step(logits, b, tau=1.0, p=0.5)
    torch.manual_seed(77)
    ops.sample_step(logits, c, tau=1.0, p=0.5)
    assert torch.equal(a, c) and not torch.equal(a, b)
    s = logits[0, 0, keep].double()
    want = torch.softmax(s, 0)
    freq = torch.stack([(ids == k).double().mean() for k in keep])
    assert float((freq - want).abs().max()) < 0.02


def test_maskgit_model(device):
    """MaskGitTransformer: reference state-dict keys, transformer logits against the oracle, training loss,
    eval decode and the 18-step generate (fused and eager sampling agree in distribution: same shapes, finite)."""
    from amk.models import ViTVQGAN
    from amk.models.maskgit import MaskGitTransformer

    torch.manual_seed(0)
    vq = ViTVQGAN(dict(dim=64, img_size=32, patch_size=4, n_heads=2, d_head=64, depth=1, mlp_dim=96, dropout=0.0),
                  dict(codebook_size=256, codebook_dim=32))
    m = MaskGitTransformer(dim=128, vq=vq, vocab_size=256, n_heads=2, d_head=64, dec_depth=2, mult=4, dropout=0.0)
    keys = set(m.bidirectional_transformer.state_dict())
    for k in ("input_proj.weight", "pos_enc", "init_norm.gamma", "init_norm.beta", "final_norm.gamma", "linear.weight",
              "decoder.layers.0.self_attn.q.0.weight", "decoder.layers.1.feed_forward.ff.0.weight"):
        assert k in keys, k
    w = {k: v.detach().clone() for k, v in m.bidirectional_transformer.state_dict().items()}
    ids = torch.randint(0, 257, (2, vq.num_patches), generator=torch.Generator().manual_seed(1))
    want = ref_cpu.maskgit_transformer(ids, w, 2, 64, 2)
    m = m.to(device)
    got = m.bidirectional_transformer(ids.to(device))
    assert_close(got, want, 2e-5, "transformer logits")
    imgs = torch.rand(2, 3, 32, 32, device=device)
    m.train()
    loss = m(imgs)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is None for p in m.vq.parameters())
    assert sum(float(p.grad.abs().sum()) for p in m.bidirectional_transformer.parameters()) > 0
    m.eval()
    assert tuple(m(imgs).shape) == (2, 3, 32, 32)
    for fused in (True, False):
        m.fused_sampling = fused
        out = m.generate(batch=2, timesteps=6)
        assert tuple(out.shape) == (2, 3, 32, 32) and torch.isfinite(out).all()


@pytest.mark.parametrize("fused", [True, False])
def test_parallel_decode_loop_matches_the_reference_run(device, fused):
    """The decode loop on the GPU (amk.models.muse.parallel_decode: HIP attention in the decoder, amk_sample_step for
    the sampling chain) against the fixture the reference's own MUSE.generate produced (tests/golden/
    muse_generate_small.npz, oracle/gen_golden.py:gen_muse_generate), with the Gumbel noise that run drew passed in:
    the ids entering the decoder at every step and the final ids must be identical."""
    import json
    import os

    from amk.models.muse import BidirectionalDecoder, parallel_decode
    from util import GOLDEN, load_golden, weights_of

    fx = load_golden("muse_generate_small")
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))["muse_generate_small"]
    cfg = meta["cfg"]
    dec = BidirectionalDecoder(**cfg).to(device).eval()
    dec.load_state_dict({k: v.to(device) for k, v in weights_of(fx).items()}, strict=True)
    trace = []
    ids = parallel_decode(dec, torch.from_numpy(fx["context"]).to(device), cfg["codebook_size"], cfg["num_patches"],
                          meta["timesteps"], fused_sampling=fused, gumbel=torch.from_numpy(fx["gumbel"]).to(device), trace=trace)
    for t, got in enumerate(trace):
        assert torch.equal(got.cpu(), torch.from_numpy(fx["ids_in"][t])), f"decoder input at step {t}"
    assert torch.equal(ids.cpu(), torch.from_numpy(fx["final_ids"]))
