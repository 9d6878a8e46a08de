"""ctypes binding of the kernel-level C-ABI (include/fastllama_hip.h) of libfastllama_hip.so.

This module is plumbing only: it loads the shared object that ``build.sh`` / ``__graft_entry__.build()``
produce in-tree and declares the prototypes.  There is NO fallback: if the library is missing, or no
gfx950 device is visible when a compute entry point is called, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# FASTLLAMA_HIP_LIB: development override (kernel ablation builds); the default is the in-tree library
LIB_PATH = os.environ.get("FASTLLAMA_HIP_LIB") or os.path.join(PKG_DIR, "libfastllama_hip.so")

FL_OK, FL_EINVAL, FL_EHIP, FL_ENOMEM, FL_ENODEV = 0, -1, -2, -3, -4
Q4_0, Q4_1 = 2, 3
BLOCK_BYTES = {Q4_0: 20, Q4_1: 24}
Q8_BLOCK_BYTES = 40
QK = 32


class FastLlamaHipError(RuntimeError):
    pass


_ROW_DEQ = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)
_ROW_Q = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)
_DOT = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class QuantizeFns(C.Structure):
    """fl_quantize_fns_t -- field-for-field the reference's quantize_fns_t (include/ggml.h:854-860)."""
    _fields_ = [("dequantize_row_q", _ROW_DEQ), ("quantize_row_q", _ROW_Q),
                ("quantize_row_q_reference", _ROW_Q), ("quantize_row_q_dot", _ROW_Q),
                ("vec_dot_q", _DOT)]


class ModelParams(C.Structure):
    """fl_model_params"""
    _fields_ = [(n, C.c_int) for n in ("n_vocab", "n_embd", "n_head", "n_layer", "n_ff", "n_ctx", "qtype",
                                       "max_batch", "tp_rank", "tp_size")]


_PROTOS = {
    # name: (restype, [argtypes])
    "fl_device_count": (C.c_int, []),
    "fl_init": (C.c_int, [C.c_int]),
    "fl_last_error": (C.c_char_p, []),
    "fl_set_warn_handler": (None, [C.c_void_p]),
    "fl_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "fl_version": (C.c_char_p, []),
    "fl_malloc": (C.c_void_p, [C.c_size_t]),
    "fl_free": (C.c_int, [C.c_void_p]),
    "fl_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fl_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fl_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fl_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "fl_stream_synchronize": (C.c_int, [C.c_void_p]),
    "fl_stream_create": (C.c_void_p, []),
    "fl_stream_destroy": (C.c_int, [C.c_void_p]),
    "fl_event_create": (C.c_void_p, []),
    "fl_event_destroy": (C.c_int, [C.c_void_p]),
    "fl_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "fl_qtensor_upload": (C.c_void_p, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_qtensor_from_device": (C.c_void_p, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_qtensor_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_qtensor_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fl_qtensor_device_bytes": (C.c_size_t, [C.c_void_p]),
    "fl_qtensor_build_h16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_qtensor_drop_h16": (None, [C.c_void_p]),
    "fl_qtensor_build_qwd": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_qtensor_drop_qwd": (None, [C.c_void_p]),
    "fl_qtensor_free": (None, [C.c_void_p]),
    "fl_quantize_row_q8_0": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_dequantize_row_q4_0": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_dequantize_row_q4_1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_vec_dot_q4_0_q8_0": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_vec_dot_q4_1_q8_0": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_get_quantize_fn": (QuantizeFns, [C.c_size_t]),
    "fl_qact_create": (C.c_void_p, [C.c_int, C.c_int]),
    "fl_qact_free": (None, [C.c_void_p]),
    "fl_quantize_q8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "fl_qact_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_mul_mat_q": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_mul_mat_q_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_comm_unique_id": (C.c_int, [C.c_void_p]),
    "fl_comm_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "fl_comm_create_local": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "fl_comm_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fl_comm_allgather_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "fl_comm_is_local": (C.c_int, [C.c_void_p]),
    "fl_comm_create_p2p": (C.c_void_p, [C.c_int, C.c_int]),
    "fl_comm_p2p_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_comm_p2p_import": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_comm_has_p2p": (C.c_int, [C.c_void_p]),
    "fl_comm_p2p_selftest": (C.c_int, [C.c_void_p]),
    "fl_comm_p2p_timeouts": (C.c_int, [C.c_void_p]),
    "fl_comm_p2p_check": (C.c_int, [C.c_void_p]),
    "fl_comm_debug_graph_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "fl_comm_rank": (C.c_int, [C.c_void_p]),
    "fl_comm_size": (C.c_int, [C.c_void_p]),
    "fl_comm_rccl_ranks": (C.c_int, [C.c_void_p]),
    "fl_comm_destroy": (None, [C.c_void_p]),
    "fl_model_create": (C.c_void_p, [C.POINTER(ModelParams)]),
    "fl_model_lora_shape": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fl_model_lora_apply": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int]),
    "fl_model_lora_restore": (C.c_int, [C.c_void_p]),
    "fl_model_tensor_download": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "fl_model_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "fl_model_finalize": (C.c_int, [C.c_void_p]),
    "fl_model_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_model_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_model_ingest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_model_profile": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "fl_model_set_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "fl_model_debug_layers": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_model_debug_export_q8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_model_logits_dev": (C.c_void_p, [C.c_void_p]),
    "fl_model_logits_ld": (C.c_int, [C.c_void_p]),
    "fl_model_logits_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_model_logits_nll": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fl_model_stream": (C.c_void_p, [C.c_void_p]),
    "fl_model_device_bytes": (C.c_size_t, [C.c_void_p]),
    "fl_model_kv_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_model_kv_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_model_free": (None, [C.c_void_p]),
    "fl_debug_tables": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_debug_rope_table": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "fl_debug_rmsnorm_quant": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.c_void_p]),
    "fl_debug_gemv_norm": (C.c_int, [C.c_void_p] * 6),
    "fl_debug_gemv_silu": (C.c_int, [C.c_void_p] * 6),
    "fl_debug_gemv_norm_silu": (C.c_int, [C.c_void_p] * 6),
    "fl_debug_gemv_norm_silu_q8": (C.c_int, [C.c_void_p] * 6),
    "fl_debug_gemv_quant": (C.c_int, [C.c_void_p] * 5),
    "fl_debug_gemv_q8": (C.c_int, [C.c_void_p] * 5),
    "fl_debug_prefill_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "fl_debug_prefill_attention_scratch": (C.c_int, [C.c_void_p, C.c_int, C.c_long]),
    "fl_debug_decode_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "fl_debug_decode_attention_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]),
    "fl_debug_silu_mul_quant": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_debug_silu_mul_quant_woven": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_debug_rope_kv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "fl_debug_gemm_f32_abt": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                        C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "fl_debug_gemm_f32_abt_exact": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                              C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "fl_model_set_exact": (C.c_int, [C.c_void_p, C.c_int]),
    "fl_model_get_exact": (C.c_int, [C.c_void_p]),
    "fl_default_exact": (C.c_int, []),
    "fl_model_prepare": (C.c_int, [C.c_void_p, C.c_int]),
    "fl_model_prepared": (C.c_int, [C.c_void_p]),
    "fl_model_memory": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fl_model_graph_nodes": (C.c_int, [C.c_void_p]),
    "fl_model_tp_folded": (C.c_int, [C.c_void_p]),
    "fl_set_op_mode": (C.c_int, [C.c_int]),
    "fl_debug_attn_exact": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_debug_softmax_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "fl_debug_attn_pv_exact_q8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p]),
    "fl_debug_mul_mat_q": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fl_debug_mul_mat_q_resid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_debug_gemm_qkv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p]),
    "fl_debug_gemm_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fl_quantize_row_q4_0": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_quantize_row_q4_1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_quantize_row_q4_0_reference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_quantize_row_q4_1_reference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fl_debug_qact_layout": (C.c_int, [C.c_void_p]),
    "fl_debug_gemm_mixed_split": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fl_debug_set": (C.c_int, [C.c_int, C.c_int]),
    "fl_quantize_q8_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


def hooks_path(lib_path: str) -> str:
    """libX.so -> libX_hooks.so next to it (build.sh)"""
    return lib_path[:-3] + "_hooks.so" if lib_path.endswith(".so") else lib_path + "_hooks"


class _Libs:
    """The product library plus, on first use of an fl_debug_* name, the test-hook library that is linked against it
    (include/fastllama_hip_test.h).  Attribute access goes to whichever exports the name."""

    def __init__(self, main: C.CDLL, path: str):
        self._main, self._path, self._hooks = main, path, None

    def _load_hooks(self) -> C.CDLL:
        if self._hooks is None:
            hp = hooks_path(self._path)
            if not os.path.exists(hp):
                raise FastLlamaHipError(f"{hp} is missing: run ./build.sh (the fl_debug_* test hooks live there)")
            lib = C.CDLL(hp)
            for name, (res, args) in _PROTOS.items():
                if name.startswith("fl_debug_"):
                    fn = getattr(lib, name)      # AttributeError if the symbol is not exported
                    fn.restype = res
                    fn.argtypes = args
            self._hooks = lib
        return self._hooks

    def __getattr__(self, name):
        if name.startswith("fl_debug_"):
            return getattr(self._load_hooks(), name)
        return getattr(self._main, name)


def load(path: str | None = None) -> "_Libs":
    """Load libfastllama_hip.so (once) and declare every prototype of include/fastllama_hip.h; fl_debug_* names resolve in
    libfastllama_hip_hooks.so (include/fastllama_hip_test.h), loaded on first use."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FastLlamaHipError(
            f"{p} is missing: run ./build.sh (or __graft_entry__.build()). "
            "fastllama_amd has no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)      # (the hook library resolves its references against this one)
    for name, (res, args) in _PROTOS.items():
        if name.startswith("fl_debug_"):
            continue
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    libs = _Libs(lib, p)
    if path is None:
        _lib = libs
    return libs


def check(rc: int, what: str = "") -> None:
    if rc != FL_OK:
        msg = load().fl_last_error().decode(errors="replace")
        raise FastLlamaHipError(f"{what or 'libfastllama_hip'} failed (rc={rc}): {msg}")


def require_device(device: int = 0) -> None:
    """Raise unless a gfx950 device is usable -- the product path never falls back to the CPU."""
    check(load().fl_init(device), "fl_init")
