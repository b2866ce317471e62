"""Seeded tensor recipes shared by oracle/gen_golden.py and the tests, so fixtures that
store only seeds (the config-size cases) can be regenerated bit-for-bit on any box with the
same torch build.  Test infrastructure only."""
import torch


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g) * scale


def seeded_param(name, shape, seed, index):
    """Value for parameter `name` (position `index` in the sorted parameter list)."""
    shape = tuple(shape)
    fan = shape[-1] if len(shape) > 1 else shape[0]
    is_norm_weight = name.endswith("weight") and (
        ("norm" in name and len(shape) == 1)
        or name.endswith("to_patch_embedding.1.weight")
        or name.endswith("to_patch_embedding.3.weight")
    )
    scale = max(fan, 1) ** -0.5
    if name.endswith("bias") or is_norm_weight:
        scale = 0.1
    v = seeded(shape, seed * 1000 + index, scale)
    if is_norm_weight:
        v = v + 1.0
    return v


def seeded_params(shapes, seed):
    """{name: tensor} for a {name: shape} dict, in sorted-name order."""
    return {n: seeded_param(n, shapes[n], seed, i) for i, n in enumerate(sorted(shapes))}


def randomize_(module, seed):
    shapes = {n: tuple(p.shape) for n, p in module.named_parameters()}
    vals = seeded_params(shapes, seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(vals[n])
