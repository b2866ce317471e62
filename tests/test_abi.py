"""The C-ABI library loads on a GPU-less box and exports exactly what include/amk.h declares.
No kernel is launched here: the calls below fail argument validation before any HIP call."""
import ctypes

import pytest

from amk import lib as amk_lib


def test_header_and_binding_agree():
    declared = set(amk_lib.declared_symbols())
    assert declared, "no amk_* declarations found in include/amk.h"
    assert declared == set(amk_lib.SIGNATURES), (
        f"header-only: {declared - set(amk_lib.SIGNATURES)}, binding-only: {set(amk_lib.SIGNATURES) - declared}"
    )


def test_library_exports_every_declared_symbol():
    L = amk_lib.load()
    for name in amk_lib.declared_symbols():
        assert hasattr(L, name), f"libamk.so does not export {name}"
    assert L.amk_version() == 140
    assert L.amk_arch() == b"gfx950"


def test_argument_errors_are_reported_not_crashed():
    L = amk_lib.load()
    null = ctypes.c_void_p(0)
    rc = L.amk_attn_fwd(null, null, null, null, null, null, null, 1, 1, 1, 1, 64, *([0] * 12), 1.0, null)
    assert rc == -1 and b"null" in L.amk_last_error()
    rc = L.amk_vq_gather(null, null, 1, 1, 32, null, null, null)
    assert rc == -1
    with pytest.raises(RuntimeError, match="amk_vq_gather"):
        amk_lib.check(rc, "amk_vq_gather")


def test_product_has_no_cpu_path():
    """CPU tensors must be refused loudly, never silently computed on the host."""
    import torch

    from amk import ops
    from amk.models import Codebook, SoftmaxAttention

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SoftmaxAttention(64, 1, 64)(torch.randn(1, 4, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Codebook(64, 32)(torch.randn(1, 4, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.attention(*(torch.randn(1, 1, 4, 64) for _ in range(3)), 0.125)
