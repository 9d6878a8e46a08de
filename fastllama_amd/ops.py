"""Host-side mirror of the reference's operator interface for the Q4 hot path, over the C-ABI.

Names and argument meaning follow the reference (``quantize_fns_t`` include/ggml.h:850-862 and
``ggml_mul_mat`` lib/ggml.c:4925-4948):

* ``QTensor``              a Q4_0/Q4_1 weight ``ne = [K, M]`` resident in HBM (QW16 layout)
* ``quantize_row_q8_0``    quantize_fns[type].quantize_row_q_dot
* ``dequantize_row_q``     quantize_fns[type].dequantize_row_q
* ``vec_dot_q``            quantize_fns[type].vec_dot_q
* ``mul_mat``              ggml_mul_mat(W, x) for quantized W  ->  ggml_compute_forward_mul_mat_q_f32

torch is used only to own device memory and to pick the stream; every computation happens inside
libfastllama_hip.so.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import hip
from .hip import Q4_0, Q4_1, BLOCK_BYTES, Q8_BLOCK_BYTES, QK, check


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor) -> C.c_void_p:
    if not t.is_cuda:
        raise hip.FastLlamaHipError("expected a device tensor: fastllama_amd has no CPU path")
    return C.c_void_p(t.data_ptr())


class QTensor:
    """Quantized weight matrix W[M rows][K], uploaded from the reference's AoS block bytes."""

    def __init__(self, qtype: int, blocks, M: int, K: int):
        lib = hip.load()
        hip.require_device(torch.cuda.current_device())
        self.qtype, self.M, self.K = qtype, M, K
        if qtype not in BLOCK_BYTES:
            raise hip.FastLlamaHipError(f"type {qtype} is not Q4_0 (2) / Q4_1 (3)")
        nbytes = M * (K // QK) * BLOCK_BYTES[qtype]
        if isinstance(blocks, torch.Tensor) and blocks.is_cuda:
            assert blocks.numel() * blocks.element_size() == nbytes
            h = lib.fl_qtensor_from_device(qtype, _dev(blocks), M, K, _stream())
        else:
            arr = np.ascontiguousarray(blocks, dtype=np.uint8)
            assert arr.size == nbytes, (arr.size, nbytes)
            h = lib.fl_qtensor_upload(qtype, arr.ctypes.data_as(C.c_void_p), M, K, _stream())
        if not h:
            raise hip.FastLlamaHipError("fl_qtensor_upload: " + lib.fl_last_error().decode())
        self.handle = C.c_void_p(h)

    def download(self) -> np.ndarray:
        out = np.empty((self.M, self.K // QK * BLOCK_BYTES[self.qtype]), dtype=np.uint8)
        check(hip.load().fl_qtensor_download(self.handle, out.ctypes.data_as(C.c_void_p), _stream()),
              "fl_qtensor_download")
        return out

    @property
    def device_bytes(self) -> int:
        return hip.load().fl_qtensor_device_bytes(self.handle)

    def free(self):
        if getattr(self, "handle", None):
            hip.load().fl_qtensor_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class QAct:
    """The Q8_0 activation scratch of mul_mat_q_f32 (params->wdata, lib/ggml.c:8105-8119)."""

    def __init__(self, max_N: int, K: int):
        lib = hip.load()
        hip.require_device(torch.cuda.current_device())
        h = lib.fl_qact_create(max_N, K)
        if not h:
            raise hip.FastLlamaHipError("fl_qact_create: " + lib.fl_last_error().decode())
        self.handle = C.c_void_p(h)
        self.max_N, self.K, self.N = max_N, K, 0

    def quantize(self, x: torch.Tensor, layout: int | None = None) -> "QAct":
        assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        N, K = x.shape
        lib = hip.load()
        if layout is None:
            check(lib.fl_quantize_q8(self.handle, _dev(x), x.stride(0), N, K, _stream()), "fl_quantize_q8")
        else:
            check(lib.fl_quantize_q8_layout(self.handle, _dev(x), x.stride(0), N, K, layout, _stream()),
                  "fl_quantize_q8_layout")
        self.N, self.K = N, K
        return self

    def export(self) -> torch.Tensor:
        """-> uint8 [N, K/32*40] device tensor in the reference's block_q8_0 layout."""
        out = torch.empty((self.N, self.K // QK * Q8_BLOCK_BYTES), dtype=torch.uint8, device="cuda")
        check(hip.load().fl_qact_export(self.handle, _dev(out), _stream()), "fl_qact_export")
        return out

    def free(self):
        if getattr(self, "handle", None):
            hip.load().fl_qact_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def quantize_row_q8_0(x: torch.Tensor) -> torch.Tensor:
    """quantize_row_q_dot(x, y, k): f32 [k] -> block_q8_0 bytes (uint8 [k/32*40]) on the device."""
    assert x.dtype == torch.float32 and x.dim() == 1 and x.is_contiguous()
    k = x.numel()
    y = torch.empty(k // QK * Q8_BLOCK_BYTES, dtype=torch.uint8, device=x.device)
    check(hip.load().fl_quantize_row_q8_0(_dev(x), _dev(y), k, _stream()), "fl_quantize_row_q8_0")
    return y


def quantize_row_q(qtype: int, x: torch.Tensor, reference: bool = False) -> torch.Tensor:
    """quantize_row_q(x, y, k) / quantize_row_q_reference: f32 [k] -> block_q4_0 / block_q4_1 bytes on the device."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    k = x.numel()
    y = torch.empty(k // QK * BLOCK_BYTES[qtype], dtype=torch.uint8, device=x.device)
    name = "fl_quantize_row_q4_%d%s" % (0 if qtype == Q4_0 else 1, "_reference" if reference else "")
    check(getattr(hip.load(), name)(_dev(x), _dev(y), k, _stream()), name)
    return y


def dequantize_row_q(qtype: int, blocks: torch.Tensor, k: int) -> torch.Tensor:
    """dequantize_row_q(x, y, k): AoS Q4 block bytes -> f32 [k]."""
    y = torch.empty(k, dtype=torch.float32, device=blocks.device)
    fn = hip.load().fl_dequantize_row_q4_0 if qtype == Q4_0 else hip.load().fl_dequantize_row_q4_1
    check(fn(_dev(blocks), _dev(y), k, _stream()), "fl_dequantize_row_q4")
    return y


def vec_dot_q(qtype: int, n: int, wrow: torch.Tensor, xq: torch.Tensor) -> torch.Tensor:
    """vec_dot_q(n, s, x, y): one Q4 row times one Q8_0 row (both AoS bytes) -> f32 [1]."""
    s = torch.empty(1, dtype=torch.float32, device=wrow.device)
    fn = hip.load().fl_vec_dot_q4_0_q8_0 if qtype == Q4_0 else hip.load().fl_vec_dot_q4_1_q8_0
    check(fn(n, _dev(s), _dev(wrow), _dev(xq), _stream()), "fl_vec_dot_q4")
    return s


def mul_mat(W: QTensor, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """ggml_mul_mat(W[K,M], x[K,N]) -> y[M,N]; tensors are row-major so x is [N, K], y is [N, M]."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == W.K
    N = x.shape[0]
    if out is None:
        ld = (W.M + 3) // 4 * 4
        buf = torch.empty((N, ld), dtype=torch.float32, device=x.device)
        out = buf[:, :W.M]
    check(hip.load().fl_mul_mat_q_f32(W.handle, _dev(x), x.stride(0), _dev(out), out.stride(0), N, _stream()),
          "fl_mul_mat_q_f32")
    return out


def mul_mat_q(W: QTensor, a: QAct, which: int | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """COMPUTE phase only, on an already quantized activation workspace."""
    if out is None:
        ld = (W.M + 3) // 4 * 4
        out = torch.empty((a.N, ld), dtype=torch.float32, device="cuda")[:, :W.M]
    lib = hip.load()
    if which is None:
        check(lib.fl_mul_mat_q(W.handle, a.handle, _dev(out), out.stride(0), _stream()), "fl_mul_mat_q")
    else:
        check(lib.fl_debug_mul_mat_q(W.handle, a.handle, _dev(out), out.stride(0), which, _stream()),
              "fl_debug_mul_mat_q")
    return out
