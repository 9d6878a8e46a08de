"""oracle -- CPU checkers for the Q4_0/Q4_1 x Q8_0 hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``fastllama_amd``) never does and fails loudly without its HIP library.

Two checkers live here:

* ``port``  -- ``liboracle.so`` built from ``oracle/q4_oracle.c``: our plain-C restatement of the
  reference's AVX2 arithmetic (each function cites the /root/reference file:line it follows).
* ``ref``   -- ``oracle/_ref/*.so``: the reference ITSELF, compiled out-of-tree from the sources
  where they lie under /root/reference by ``oracle/Makefile`` (never copied into this repo).
  ``_ref`` is git-ignored but ships to the GPU box with the gpurun snapshot.

Parity status: the port is PINNED -- bit-identical to ``ref`` on every function
(tests/test_oracle_pinning.py) and on the committed golden vectors (tests/golden/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

Q4_0, Q4_1 = 2, 3                      # include/ggml.h:200-212
BLOCK_BYTES = {Q4_0: 20, Q4_1: 24}     # lib/ggml.c:590-603
Q8_BLOCK_BYTES = 40                    # lib/ggml.c:620-626
QK = 32


def build(verbose: bool = False) -> None:
    """(Re)build liboracle.so and, when /root/reference is present, oracle/_ref/."""
    r = subprocess.run(["make", "-s", "-C", HERE], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + (r.stdout or "") + (r.stderr or ""))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Port:
    """ctypes view of liboracle.so (oracle/q4_oracle.c)."""

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        L = self.lib
        for name in ("orc_quantize_row_q4_0", "orc_quantize_row_q4_1", "orc_quantize_row_q8_0", "orc_quantize_row_q4_0_simd",
                     "orc_quantize_row_q4_1_simd"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            getattr(L, name).restype = None
        for name in ("orc_dequantize_row_q4_0", "orc_dequantize_row_q4_1"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            getattr(L, name).restype = None
        for name in ("orc_vec_dot_q4_0_q8_0", "orc_vec_dot_q4_1_q8_0"):
            getattr(L, name).argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(L, name).restype = None
        L.orc_mul_mat_q_f32.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_mul_mat_q_f32.restype = C.c_int
        L.orc_mul_mat_q_f32_ex.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_mul_mat_q_f32_ex.restype = C.c_int
        L.orc_quantize_q4.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_quantize_q4.restype = None
        L.orc_lora_add.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_void_p]
        L.orc_lora_add.restype = C.c_int
        L.orc_mul_mat_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_mul_mat_f32.restype = None
        L.orc_vec_dot_f32_mm.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_vec_dot_f32_mm.restype = C.c_float

    # ---- weights ----
    def quantize_q4(self, qtype: int, w: np.ndarray) -> np.ndarray:
        """f32 [M, K] -> AoS blocks, uint8 [M, K/32 * block_bytes] (reference file layout)."""
        w = np.ascontiguousarray(w, dtype=np.float32)
        M, K = w.shape
        out = np.empty((M, K // QK * BLOCK_BYTES[qtype]), dtype=np.uint8)
        self.lib.orc_quantize_q4(qtype, _ptr(w), _ptr(out), w.size, K)
        return out

    def quantize_row_q4(self, qtype: int, x: np.ndarray, reference: bool) -> np.ndarray:
        """quantize_fns[type].quantize_row_q_reference (reference=True) / .quantize_row_q as built for AVX2."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.size // QK * BLOCK_BYTES[qtype], dtype=np.uint8)
        name = "orc_quantize_row_q4_%d%s" % (0 if qtype == Q4_0 else 1, "" if reference else "_simd")
        getattr(self.lib, name)(_ptr(x), _ptr(out), x.size)
        return out

    def dequantize_row(self, qtype: int, row: np.ndarray, K: int) -> np.ndarray:
        row = np.ascontiguousarray(row, dtype=np.uint8)
        out = np.empty(K, dtype=np.float32)
        fn = self.lib.orc_dequantize_row_q4_0 if qtype == Q4_0 else self.lib.orc_dequantize_row_q4_1
        fn(_ptr(row), _ptr(out), K)
        return out

    def dequantize(self, qtype: int, wq: np.ndarray, K: int) -> np.ndarray:
        return np.stack([self.dequantize_row(qtype, r, K) for r in wq])


    def mul_mat_f32(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        """ggml_mul_mat on two f32 matrices (ggml_compute_forward_mul_mat_f32 as the reference's build compiled it):
        a [M, K], b [N, K] (rows may be strided views with contiguous elements) -> [N, M]."""
        assert a.dtype == np.float32 and b.dtype == np.float32 and a.shape[1] == b.shape[1]
        assert a.strides[1] == 4 and b.strides[1] == 4 and a.strides[0] % 4 == 0 and b.strides[0] % 4 == 0
        M, K = a.shape
        N = b.shape[0]
        out = np.empty((N, M), dtype=np.float32)
        self.lib.orc_mul_mat_f32(_ptr(a), a.strides[0] // 4, _ptr(b), b.strides[0] // 4, _ptr(out), M, M, N, K)
        return out

    def lora_add(self, qtype: int, wq: np.ndarray, K: int, a=None, b=None, ba=None, sign: float = 1.0):
        """LoRA merge on AoS Q4 rows (oracle/q4_oracle.c:orc_lora_add / oracle/ref_driver.c:ref_lora_add).
        a: [K, r] f32, b: [M, r] f32 (uncached adapter) or ba: [M, K] f32 (cached).  Returns (new rows, BA)."""
        out = np.ascontiguousarray(wq, dtype=np.uint8).copy()
        M = out.shape[0]
        if ba is not None:
            ba = np.ascontiguousarray(ba, dtype=np.float32)
            r, pa, pb, pba = 0, None, None, _ptr(ba)
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            r, pa, pb, pba = a.shape[1], _ptr(a), _ptr(b), None
        got = np.empty((M, K), dtype=np.float32)
        rc = self.lib.orc_lora_add(qtype, _ptr(out), pa, pb, pba, r, M, K, float(sign), _ptr(got))
        if rc != 0:
            raise RuntimeError(f"lora_add failed rc={rc}")
        return out, got

    # ---- activations ----
    def quantize_row_q8_0(self, x: np.ndarray) -> np.ndarray:
        """f32 [K] -> AoS block_q8_0 bytes, uint8 [K/32*40]."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.size // QK * Q8_BLOCK_BYTES, dtype=np.uint8)
        self.lib.orc_quantize_row_q8_0(_ptr(x), _ptr(out), x.size)
        return out

    def vec_dot(self, qtype: int, K: int, wrow: np.ndarray, xq: np.ndarray) -> np.float32:
        s = np.zeros(1, dtype=np.float32)
        fn = self.lib.orc_vec_dot_q4_0_q8_0 if qtype == Q4_0 else self.lib.orc_vec_dot_q4_1_q8_0
        fn(K, _ptr(s), _ptr(np.ascontiguousarray(wrow)), _ptr(np.ascontiguousarray(xq)))
        return s[0]

    def mul_mat_q(self, qtype: int, wq: np.ndarray, x: np.ndarray, n_threads: int = 0, strict: bool = True) -> np.ndarray:
        """y[N, M] = mul_mat_q_f32(W[M, K] (AoS blocks), x[N, K]).  strict=False waives the reference's
        even-block-count assert (tensor-parallel K shards)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        wq = np.ascontiguousarray(wq, dtype=np.uint8)
        N, K = x.shape
        M = wq.shape[0]
        assert wq.shape[1] == K // QK * BLOCK_BYTES[qtype]
        y = np.empty((N, M), dtype=np.float32)
        rc = self.lib.orc_mul_mat_q_f32_ex(qtype, _ptr(wq), _ptr(x), _ptr(y), M, K, N,
                                           n_threads or (os.cpu_count() or 1), 1 if strict else 0)
        if rc != 0:
            raise ValueError("orc_mul_mat_q_f32 rejected the arguments (K must be a multiple of 64)")
        return y


# quantize_fns_t, include/ggml.h:850-862 (five function pointers, in this order)
_FN_ROW = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)
_FN_DOT = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class _QuantizeFns(C.Structure):
    _fields_ = [("dequantize_row_q", _FN_ROW), ("quantize_row_q", _FN_ROW),
                ("quantize_row_q_reference", _FN_ROW), ("quantize_row_q_dot", _FN_ROW),
                ("vec_dot_q", _FN_DOT)]


def have_ref() -> bool:
    return (os.path.exists(os.path.join(REF_DIR, "libggml_ref.so"))
            and os.path.exists(os.path.join(REF_DIR, "pyfastllama.so")))


class Ref:
    """ctypes view of the compiled reference (oracle/_ref/libggml_ref.so)."""

    def __init__(self):
        if not have_ref():
            build()
        if not have_ref():
            raise FileNotFoundError("oracle/_ref is not built and /root/reference is absent")
        self.lib = C.CDLL(os.path.join(REF_DIR, "libggml_ref.so"))
        L = self.lib
        L.ggml_internal_get_quantize_fn.restype = _QuantizeFns
        L.ggml_internal_get_quantize_fn.argtypes = [C.c_size_t]
        for name in ("ggml_quantize_q4_0", "ggml_quantize_q4_1"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ref_mul_mat_q.restype = C.c_int
        L.ref_mul_mat_q.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_mm_open.restype = C.c_void_p
        L.ref_mm_open.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_mm_run.argtypes = [C.c_void_p]
        L.ref_mm_run.restype = None
        L.ref_mm_read.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_mm_read.restype = None
        L.ref_mm_close.argtypes = [C.c_void_p]
        L.ref_mm_close.restype = None
        L.ref_lora_add.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_void_p, C.c_int]
        L.ref_lora_add.restype = C.c_int
        self.fns = {t: L.ggml_internal_get_quantize_fn(t) for t in (Q4_0, Q4_1)}

    def quantize_q4(self, qtype: int, w: np.ndarray) -> np.ndarray:
        w = np.ascontiguousarray(w, dtype=np.float32)
        M, K = w.shape
        out = np.empty((M, K // QK * BLOCK_BYTES[qtype]), dtype=np.uint8)
        hist = np.zeros(16, dtype=np.int64)
        fn = self.lib.ggml_quantize_q4_0 if qtype == Q4_0 else self.lib.ggml_quantize_q4_1
        # ggml_quantize_q4_x takes an int element count: go row-chunk by row-chunk for big tensors
        rows_per = max(1, (1 << 30) // K)
        for r0 in range(0, M, rows_per):
            r1 = min(M, r0 + rows_per)
            fn(_ptr(w[r0:r1]), _ptr(out[r0:r1]), (r1 - r0) * K, K, _ptr(hist))
        return out

    def quantize_row_q4(self, qtype: int, x: np.ndarray, reference: bool) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.size // QK * BLOCK_BYTES[qtype], dtype=np.uint8)
        f = self.fns[qtype]
        (f.quantize_row_q_reference if reference else f.quantize_row_q)(_ptr(x), _ptr(out), x.size)
        return out

    def dequantize_row(self, qtype: int, row: np.ndarray, K: int) -> np.ndarray:
        row = np.ascontiguousarray(row, dtype=np.uint8)
        out = np.empty(K, dtype=np.float32)
        self.fns[qtype].dequantize_row_q(_ptr(row), _ptr(out), K)
        return out

    def lora_add(self, qtype: int, wq: np.ndarray, K: int, a=None, b=None, ba=None, sign: float = 1.0):
        """LoRA merge on AoS Q4 rows (oracle/q4_oracle.c:orc_lora_add / oracle/ref_driver.c:ref_lora_add).
        a: [K, r] f32, b: [M, r] f32 (uncached adapter) or ba: [M, K] f32 (cached).  Returns (new rows, BA)."""
        out = np.ascontiguousarray(wq, dtype=np.uint8).copy()
        M = out.shape[0]
        if ba is not None:
            ba = np.ascontiguousarray(ba, dtype=np.float32)
            r, pa, pb, pba = 0, None, None, _ptr(ba)
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            r, pa, pb, pba = a.shape[1], _ptr(a), _ptr(b), None
        got = np.empty((M, K), dtype=np.float32)
        rc = self.lib.ref_lora_add(qtype, _ptr(out), pa, pb, pba, r, M, K, float(sign), _ptr(got), 4)
        if rc != 0:
            raise RuntimeError(f"lora_add failed rc={rc}")
        return out, got

    def quantize_row_q8_0(self, x: np.ndarray, qtype: int = Q4_0) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.size // QK * Q8_BLOCK_BYTES, dtype=np.uint8)
        self.fns[qtype].quantize_row_q_dot(_ptr(x), _ptr(out), x.size)
        return out

    def vec_dot(self, qtype: int, K: int, wrow: np.ndarray, xq: np.ndarray) -> np.float32:
        s = np.zeros(1, dtype=np.float32)
        self.fns[qtype].vec_dot_q(K, _ptr(s), _ptr(np.ascontiguousarray(wrow)),
                                  _ptr(np.ascontiguousarray(xq)))
        return s[0]

    def mul_mat_q(self, qtype: int, wq: np.ndarray, x: np.ndarray, n_threads: int = 0,
                  reps: int = 1) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        wq = np.ascontiguousarray(wq, dtype=np.uint8)
        N, K = x.shape
        M = wq.shape[0]
        y = np.empty((N, M), dtype=np.float32)
        rc = self.lib.ref_mul_mat_q(qtype, _ptr(wq), _ptr(x), _ptr(y), M, K, N,
                                    n_threads or (os.cpu_count() or 1), reps)
        if rc != 0:
            raise RuntimeError(f"ref_mul_mat_q failed rc={rc}")
        return y

    def timed_mul_mat(self, qtype: int, wq: np.ndarray, x: np.ndarray, n_threads: int, reps: int):
        """Returns (list of per-rep seconds, y) timing only ggml_graph_compute."""
        import time
        x = np.ascontiguousarray(x, dtype=np.float32)
        wq = np.ascontiguousarray(wq, dtype=np.uint8)
        N, K = x.shape
        M = wq.shape[0]
        h = self.lib.ref_mm_open(qtype, _ptr(wq), _ptr(x), M, K, N, n_threads)
        if not h:
            raise RuntimeError("ref_mm_open failed")
        ts = []
        try:
            for _ in range(reps):
                t0 = time.perf_counter()
                self.lib.ref_mm_run(h)
                ts.append(time.perf_counter() - t0)
            y = np.empty((N, M), dtype=np.float32)
            self.lib.ref_mm_read(h, _ptr(y))
        finally:
            self.lib.ref_mm_close(h)
        return ts, y


def split_q8(xq: np.ndarray):
    """AoS block_q8_0 bytes -> (d f32[nb], s f32[nb], q int8[nb,32])."""
    b = np.ascontiguousarray(xq, dtype=np.uint8).reshape(-1, Q8_BLOCK_BYTES)
    d = b[:, 0:4].copy().view(np.float32).reshape(-1)
    s = b[:, 4:8].copy().view(np.float32).reshape(-1)
    q = b[:, 8:40].copy().view(np.int8)
    return d, s, q


def split_q4(qtype: int, wq: np.ndarray):
    """AoS Q4 bytes [M, nb*bs] -> (d f32[M,nb], m f32[M,nb] or None, qs uint8[M,nb,16])."""
    wq = np.ascontiguousarray(wq, dtype=np.uint8)
    M = wq.shape[0]
    bs = BLOCK_BYTES[qtype]
    b = wq.reshape(M, -1, bs)
    d = b[:, :, 0:4].copy().view(np.float32).reshape(M, -1)
    if qtype == Q4_0:
        return d, None, b[:, :, 4:20].copy()
    m = b[:, :, 4:8].copy().view(np.float32).reshape(M, -1)
    return d, m, b[:, :, 8:24].copy()
