"""ctypes binding of ``libvla_native.so`` (C ABI declared in ``include/vla_native.h``).

The product path has NO fallback: if the HIP library is missing this module raises, and every op raises on a
non-zero return code with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = (os.environ.get("VLA_NATIVE_LIB") or None) or os.path.join(_HERE, "libvla_native.so")   # override: same-box A/B of two builds

ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_TANH, ACT_SWIGLU, ACT_SWIGLU_BWD = 0, 1, 2, 3, 4, 5
ABI_VERSION = 6          # include/vla_native.h: VLA_ABI_VERSION


class NativeLibraryMissing(ImportError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
                ("C2", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("ldr", C.c_int), ("ldc2", C.c_int), ("res_mod", C.c_int), ("act", C.c_int), ("batch", C.c_int),
                ("sA", C.c_longlong), ("sB", C.c_longlong), ("sC", C.c_longlong), ("sR", C.c_longlong),
                ("sC2", C.c_longlong), ("sBias", C.c_longlong), ("alpha", C.c_float),
                ("a_group", C.c_int), ("c_group", C.c_int), ("a_group_stride", C.c_longlong), ("c_group_stride", C.c_longlong),
                ("rope_mode", C.c_int), ("rope_T", C.c_int), ("rope_dh", C.c_int), ("rope_cols", C.c_int),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("r_group", C.c_int), ("r_group_stride", C.c_longlong), ("c_live_mod", C.c_int), ("c_live_from", C.c_int),
                ("split_k", C.c_int), ("ws", C.c_void_p), ("bias_post_round", C.c_int),
                ("fp8", C.c_int), ("a_scale", C.c_void_p), ("b_scale", C.c_void_p),
                ("A2", C.c_void_p), ("B2", C.c_void_p), ("K2", C.c_int), ("lda2", C.c_int), ("ldb2", C.c_int)]


class GemmTnDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("R", C.c_void_p),
                ("M", C.c_int), ("N1", C.c_int), ("N2", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("ldr", C.c_int), ("batch", C.c_int),
                ("sA", C.c_longlong), ("sB", C.c_longlong), ("sC", C.c_longlong), ("sR", C.c_longlong), ("alpha", C.c_float),
                ("a_group", C.c_int), ("b_group", C.c_int), ("a_group_stride", C.c_longlong), ("b_group_stride", C.c_longlong),
                ("a_col_group", C.c_int), ("a_col_group_stride", C.c_int), ("split", C.c_int), ("ws", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("kmask", C.c_void_p),
                ("q_sb", C.c_longlong), ("k_sb", C.c_longlong), ("v_sb", C.c_longlong), ("o_sb", C.c_longlong),
                ("q_ss", C.c_int), ("k_ss", C.c_int), ("v_ss", C.c_int), ("o_ss", C.c_int),
                ("B", C.c_int), ("Sq", C.c_int), ("Sk", C.c_int), ("Hq", C.c_int), ("Hkv", C.c_int), ("dh", C.c_int),
                ("causal", C.c_int), ("scale", C.c_float),
                ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("delta", C.c_void_p),
                ("do_sb", C.c_longlong), ("dq_sb", C.c_longlong), ("dk_sb", C.c_longlong), ("dv_sb", C.c_longlong),
                ("do_ss", C.c_int), ("dq_ss", C.c_int), ("dk_ss", C.c_int), ("dv_ss", C.c_int),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("q_off", C.c_int), ("dkv_k0", C.c_int), ("lse_hs", C.c_int)]


class HeadAttnDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k_self", C.c_void_p), ("v_self", C.c_void_p), ("k_adp", C.c_void_p),
                ("v_adp", C.c_void_p), ("k_task", C.c_void_p), ("v_task", C.c_void_p), ("gate", C.c_void_p),
                ("out", C.c_void_p), ("probs", C.c_void_p),
                ("B", C.c_int), ("T", C.c_int), ("Ka", C.c_int), ("Kt", C.c_int), ("H", C.c_int), ("dh", C.c_int),
                ("ld_q", C.c_int), ("ld_self", C.c_int), ("ld_adp", C.c_int), ("ld_task", C.c_int), ("ld_out", C.c_int),
                ("gate_on_adapter", C.c_int),
                ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk_self", C.c_void_p), ("dv_self", C.c_void_p),
                ("dk_adp", C.c_void_p), ("dv_adp", C.c_void_p), ("dk_task", C.c_void_p), ("dv_task", C.c_void_p),
                ("dgate", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("ws", C.c_void_p), ("ws_floats", C.c_longlong), ("ref_softmax", C.c_int)]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double
_PROTOS = {
    "vla_version": ([], _I),
    "vla_desc_size": ([_I], _I),
    "vla_gemm256_extent_ok": ([C.POINTER(GemmDesc)], _I),
    "vla_gemm_uses_256": ([C.POINTER(GemmDesc)], _I),
    "vla_gemm_latency_hint": ([C.c_int], _I),
    "vla_gemm_bf16_tn": ([_P, C.POINTER(GemmTnDesc)], _I),
    "vla_gemm_bf16_tn_grouped": ([_P, C.POINTER(GemmTnDesc), _I], _I),
    "vla_copy_rows3d": ([_P, _P, _P, _I, _I, _I, _L, _L, _L, _L], _I),
    "vla_layerscale_fwd": ([_P, _P, _P, _P, _P, _L, _I], _I),
    "vla_layerscale_bwd": ([_P, _P, _P, _P, _P, _P, _I, _I], _I),
    "vla_gemm_bf16_nt": ([_P, C.POINTER(GemmDesc)], _I),
    "vla_transpose_bf16": ([_P, _P, _P, _I, _I, _I, _I, _I, _L, _L], _I),
    "vla_layernorm_fwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F], _I),
    "vla_layernorm_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I], _I),
    "vla_rmsnorm_fwd": ([_P, _P, _P, _P, _P, _I, _I, _F], _I),
    "vla_rmsnorm_bwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I], _I),
    "vla_attn_fwd": ([_P, C.POINTER(AttnDesc)], _I),
    "vla_attn_bwd": ([_P, C.POINTER(AttnDesc)], _I),
    "vla_rope_half": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I], _I),
    "vla_rope_interleaved": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I], _I),
    "vla_im2col_patch": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I], _I),
    "vla_image_normalize_u8": ([_P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _I], _I),
    "vla_action_tokenize": ([_P, _P, _P, _P, _L, _I, _F, _F, _L], _I),
    "vla_action_mask": ([_P, _P, _P, _P, _P, _I, _I, _I], _I),
    "vla_embed_splice": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I], _I),
    "vla_action_query_grad": ([_P, _P, _P, _P, _I, _I, _I, _I, _I], _I),
    "vla_gather_rows": ([_P, _P, _P, _P, _I, _I, _I, _I], _I),
    "vla_scatter_add_rows": ([_P, _P, _P, _P, _I, _I, _I, _I], _I),
    "vla_add_bf16": ([_P, _P, _P, _P, _L], _I),
    "vla_gelu_fwd": ([_P, _P, _P, _L], _I),
    "vla_gelu_bwd": ([_P, _P, _P, _P, _L], _I),
    "vla_relu_bwd": ([_P, _P, _P, _P, _L], _I),
    "vla_swiglu_bwd": ([_P, _P, _P, _P, _I, _I], _I),
    "vla_swiglu_fwd": ([_P, _P, _P, _I, _I], _I),
    "vla_colsum_bf16": ([_P, _P, _P, _I, _I, _I, _I, _L, _L], _I),
    "vla_cast_f32_bf16": ([_P, _P, _P, _L], _I),
    "vla_cast_bf16_f32": ([_P, _P, _P, _L], _I),
    "vla_rmsnorm_dw": ([_P, _P, _P, _P, _P, _I, _I], _I),
    "vla_embed_grad": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I], _I),
    "vla_resample_u8": ([_P, _P, _P, _L, _I, _I, _I, _P, _P, _I], _I),
    "vla_token_ce": ([_P, _P, _L, _P, _I, _I, _P], _I),
    "vla_token_ce_bwd": ([_P, _P, _L, _P, _I, _I, _P, _F, _P, _L], _I),
    "vla_copy2d": ([_P, _P, _P, _L, _I, _L, _L, _I, _I, _I, _I, _L], _I),
    "vla_fill_zero": ([_P, _P, _L], _I),
    "vla_dropout_bf16": ([_P, _P, _P, _L, C.c_int, _L, _L, C.c_float, C.c_ulonglong, _P], _I),
    "vla_dropout_bwd_add_bf16": ([_P, _P, _P, _L, C.c_int, _L, _L, C.c_float, C.c_ulonglong, _P], _I),
    "vla_inc_i32": ([_P, _P], _I),
    "vla_quant_fp8_rows": ([_P, _P, _P, _P, _I, _I, _I, _I], _I),
    "vla_rmsnorm_fwd_q8": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F], _I),
    "vla_layernorm_fwd_q8": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F], _I),
    "vla_head_index_prep": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I], _I),
    "vla_add_scalar_f32": ([_P, _P, _P, _I], _I),
    "vla_head_attn_fwd": ([_P, C.POINTER(HeadAttnDesc)], _I),
    "vla_head_attn_bwd": ([_P, C.POINTER(HeadAttnDesc)], _I),
    "vla_l1_loss": ([_P, _P, _P, _P, _P, _I, _I, _I, _F], _I),
    "vla_adamw_bf16": ([_P, _P, _P, _P, _P, _L, _D, _D, _D, _D, _D, _I, _I, _F], _I),
}
# symbols include/vla_native.h declares (checked by tests/test_abi.py without touching a GPU)
ABI_SYMBOLS = sorted(list(_PROTOS) + ["vla_last_error"])

_lib = None


def load():
    """Load the shared library (once).  Raises NativeLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {os.path.join(_HERE, 'csrc')}`).  There is no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    lib.vla_last_error.argtypes, lib.vla_last_error.restype = [], C.c_char_p
    # the struct ABI: a library built from another header revision would read past (or short of) the descriptors passed to it
    if lib.vla_version() != ABI_VERSION:
        raise NativeLibraryMissing(f"{LIB_PATH} speaks ABI {lib.vla_version()}, this binding ABI {ABI_VERSION}: rebuild (make -C csrc)")
    for which, st in enumerate((GemmDesc, AttnDesc, HeadAttnDesc, GemmTnDesc)):
        if lib.vla_desc_size(which) != C.sizeof(st):
            raise NativeLibraryMissing(f"{LIB_PATH}: sizeof({st.__name__}) is {lib.vla_desc_size(which)} in the library, {C.sizeof(st)} in the binding")
    _lib = lib
    return lib


class NativeError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vla_last_error().decode(errors="replace")
        raise NativeError(f"libvla_native: {what} failed with code {rc}: {msg}")
