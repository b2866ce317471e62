"""ctypes binding of libamk.so (the C ABI declared in include/amk.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950) and lives next to
this file.  There is no fallback: if the shared object is missing, or a kernel reports an
error, this module raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMK_LIB") or os.path.join(_HERE, "libamk.so")  # AMK_LIB: A/B builds (tools/)
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "amk.h"))

AMK_OK = 0

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float



class GemmDesc(_c.Structure):
    """amk_gemm_desc of include/amk.h (field for field)."""
    _fields_ = [("op", _c.c_int32), ("epilogue", _c.c_int32), ("m", _L),
                ("n", _c.c_int32), ("k", _c.c_int32), ("split", _c.c_int32), ("reserved", _c.c_int32),
                ("a", _P), ("a2", _P), ("w", _P), ("w2", _P), ("c", _P), ("c2", _P),
                ("lda", _L), ("lda2", _L), ("ldw", _L), ("ldw2", _L), ("ldc", _L), ("ldc2", _L),
                ("bias", _P), ("bias2", _P), ("resid", _P), ("ldr", _L),
                ("ln_mean", _P), ("ln_rstd", _P), ("ln_gamma", _P), ("ln_beta", _P),
                ("ab", _P), ("ldab", _L), ("gate", _P), ("ldg", _L), ("dbias", _P)]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_RESID, EPI_SWIGLU, EPI_SWIGLU_BWD = 0, 1, 2, 3

# name -> (restype, argtypes); mirrors include/amk.h one to one (tests/test_abi.py checks it).
SIGNATURES = {
    "amk_version": (_I, []),
    "amk_arch": (_c.c_char_p, []),
    "amk_last_error": (_c.c_char_p, []),
    "amk_attn_fwd": (_I, [_P] * 7 + [_I] * 5 + [_L] * 12 + [_F, _P]),
    "amk_attn_scores_bytes": (_L, [_I, _I, _I, _I]),
    "amk_attn_fwd_keep": (_I, [_P] * 8 + [_I] * 5 + [_L] * 12 + [_F, _P]),
    "amk_attn_fwd_x6_ws_bytes": (_L, [_I, _I, _I]),
    "amk_attn_fwd_x6": (_I, [_P] * 8 + [_I] * 5 + [_L] * 12 + [_F, _P]),
    "amk_attn_bwd_ws_floats": (_L, [_I, _I, _I, _I, _I]),
    "amk_attn_bwd": (_I, [_P] * 12 + [_I] * 5 + [_L] * 24 + [_F, _I, _P]),
    "amk_attn_bwd_kept": (_I, [_P] * 13 + [_I] * 5 + [_L] * 24 + [_F, _I, _P]),
    "amk_vq_num_partials": (_L, [_L]),
    "amk_vq_lookup_fwd": (_I, [_P, _P, _L, _I, _I, _I] + [_P] * 9 + [_P]),
    "amk_vq_lookup_bwd": (_I, [_P] * 7 + [_F, _L, _I, _I, _P, _P, _P]),
    "amk_vq_lookup_bwd_rows": (_I, [_P] * 7 + [_F, _L, _I, _I, _P, _P, _P]),
    "amk_vq_padded_codes": (_I, [_I, _I]),
    "amk_vq_gather": (_I, [_P, _P, _L, _I, _I, _P, _P, _P]),
    "amk_agent_num_chunks": (_I, [_I]),
    "amk_agent_ws_floats": (_L, [_I, _I, _I, _I, _I]),
    "amk_agent_attn_fwd": (_I, [_P] * 10 + [_I] * 5 + [_L] * 12 + [_F, _P]),
    "amk_agent_attn_bwd": (_I, [_P] * 14 + [_I] * 5 + [_L] * 21 + [_F, _P]),
    "amk_agent_conv_grad_reduce": (_I, [_P, _P, _L, _I, _P, _P, _P]),
    "amk_rowsum_num_partials": (_I, [_L]),
    "amk_add_layernorm_fwd": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "amk_add_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "amk_colsum": (_I, [_P, _L, _I, _P, _P]),
    "amk_swiglu_fwd": (_I, [_P, _L, _I, _P, _P]),
    "amk_swiglu_bwd": (_I, [_P, _P, _L, _I, _P, _P]),
    "amk_geglu_fwd": (_I, [_P, _L, _I, _P, _P]),
    "amk_geglu_bwd": (_I, [_P, _P, _L, _I, _P, _P]),
    "amk_opt_num_partials": (_I, []),
    "amk_sumsq_partials": (_I, [_P, _L, _P, _P]),
    "amk_adam_flat_step": (_I, [_P] * 4 + [_L, _P, _P, _P, _I] + [_F] * 6 + [_I, _P, _P]),
    "amk_adam_flat_step_shadow": (_I, [_P] * 4 + [_L, _P, _P, _P, _I] + [_F] * 6 + [_I, _P, _P, _P]),
    "amk_gemm_x6_planes_bytes": (_L, [_I, _I]),
    "amk_gemm_x6_split": (_I, [_P, _L, _I, _I, _P, _P]),
    "amk_gemm_x6_nt": (_I, [_P, _L, _P, _P, _P, _L, _I, _I, _I, _P]),
    "amk_gemm_f32_ws_bytes": (_L, [_c.POINTER(GemmDesc)]),
    "amk_gemm_f32": (_I, [_c.POINTER(GemmDesc), _P, _L, _P]),
    "amk_row_stats": (_I, [_P, _L, _I, _F, _P, _P, _P]),
    "amk_attn_bf16_fwd": (_I, [_P] * 7 + [_I] * 5 + [_L] * 12 + [_F, _P]),
    "amk_attn_bf16_bwd_ws_floats": (_L, [_I, _I, _I, _I]),
    "amk_attn_bf16_bwd": (_I, [_P] * 12 + [_I] * 5 + [_L] * 24 + [_F, _P]),
    "amk_swiglu_bf16_fwd": (_I, [_P, _L, _I, _P, _P]),
    "amk_swiglu_bf16_bwd": (_I, [_P, _P, _L, _I, _P, _P]),
    "amk_add_layernorm_mixed_fwd": (_I, [_P, _I, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "amk_add_layernorm_mixed_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "amk_gemm_bf16": (_I, [_I, _I, _P, _L, _P, _L, _P, _P, _L, _P, _L, _L, _I, _I, _P]),
    "amk_gemm_bf16_swiglu_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _L, _I, _I, _P]),
    "amk_gemm_tn_bf16_ws_bytes": (_L, [_L, _I, _I]),
    "amk_gemm_tn_bf16": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _P, _L, _P]),
    "amk_sample_step": (_I, [_P, _P, _F, _P, _c.c_uint64, _c.c_uint64, _F, _L, _I, _I, _P, _F, _P, _P, _P]),
    "amk_moe_route_ws_ints": (_L, [_L, _I, _I]),
    "amk_moe_route": (_I, [_P, _L, _I, _I] + [_P] * 8),
    "amk_grouped_gemm_nt": (_I, [_P, _L, _I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "amk_grouped_gemm_nn": (_I, [_P, _L, _I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "amk_grouped_gemm_nt_acc": (_I, [_P, _L, _I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _I, _P]),
    "amk_grouped_gemm_nn_acc": (_I, [_P, _L, _I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _I, _P]),
    "amk_grouped_gemm_wgrad": (_I, [_P, _L, _I, _P, _L, _I, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P]),
    "amk_moe_combine": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "amk_moe_expert_sums": (_I, [_P, _L, _I, _P, _P, _L, _I, _I, _I, _P, _P]),
    "amk_moe_topk": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "amk_moe_route_distinct": (_I, [_P, _L, _I, _I, _P, _P, _P, _P]),
    "amk_moe_combine_rows": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "amk_moe_gate_grad_rows": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "amk_moe_gate_grad": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
}

_lib = None


def declared_symbols(header_path=HEADER_PATH):
    """Function names declared in include/amk.h (used by the ABI test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amk_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load libamk.so once and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). amk has no CPU or eager fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    arch = lib.amk_arch().decode()
    if arch != "gfx950":
        raise RuntimeError(f"libamk.so was built for {arch}, expected gfx950")
    _lib = lib
    return lib


def check(rc, what):
    if rc != AMK_OK:
        msg = load().amk_last_error().decode()
        raise RuntimeError(f"{what} failed with code {rc}: {msg}")
