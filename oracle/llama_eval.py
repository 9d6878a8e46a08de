"""oracle/llama_eval.py -- numpy restatement of the reference's Model::eval (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/lib/llama.cpp:272-499 op by op, with the numerics of the ggml ops it calls
(citations inline).  The quantized matmuls go through the C restatement (oracle.Port.mul_mat_q), which is
bit-exact against the reference; everything else is numpy in f32/f64 exactly where ggml uses them.
Pinned against the compiled reference itself in tests/test_llama_eval_oracle.py (logits through the
reference's own C-ABI on a synthetic GGJT model).

Besides checking the GPU engine, this is the CPU stand-in used by the world_size-2 gloo tests to prove the
tensor-parallel splits reproduce the unsharded eval: row / column shards + all-reduce of the partial sums (to round-off), and the
all-rows split + all-gathers of the reference-order mode (bit for bit).
"""
from __future__ import annotations

import numpy as np

from . import Port, Q4_0, Q4_1, QK, BLOCK_BYTES

import ctypes as _C
import ctypes.util as _cu

f32 = np.float32

_libm = _C.CDLL(_cu.find_library("m") or "libm.so.6")
for _n in ("expf", "sinf", "cosf"):
    getattr(_libm, _n).restype = _C.c_float
    getattr(_libm, _n).argtypes = [_C.c_float]
_libm.powf.restype = _C.c_float
_libm.powf.argtypes = [_C.c_float, _C.c_float]


def _expf(v: np.ndarray) -> np.ndarray:
    """glibc expf element-wise: the reference's tables are built with the host libm (lib/ggml.c:3681-3688)."""
    out = np.empty(v.shape, f32)
    fl, ol = v.reshape(-1), out.reshape(-1)
    for i in range(fl.size):
        ol[i] = _libm.expf(float(fl[i]))
    return out


def _fp16_tables():
    """table_exp_f16 / table_silu_f16 as f32 values indexed by fp16 bits -- lib/ggml.c:3681-3688."""
    bits = np.arange(1 << 16, dtype=np.uint16)
    f = bits.view(np.float16).astype(f32)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        e = _expf(f)                                                # expf
        exp_tab = e.astype(np.float16).astype(f32)                  # GGML_FP32_TO_FP16 (round to nearest even)
        silu = (f / (f32(1.0) + _expf(-f))).astype(f32)             # ggml_silu_f32, lib/ggml.c:3196
        silu_tab = silu.astype(np.float16).astype(f32)
    return exp_tab, silu_tab


_EXP_TAB, _SILU_TAB = None, None


def tables():
    global _EXP_TAB, _SILU_TAB
    if _EXP_TAB is None:
        _EXP_TAB, _SILU_TAB = _fp16_tables()
    return _EXP_TAB, _SILU_TAB


def rms_norm_mul(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """ggml_rms_norm (lib/ggml.c:7406-7430: f64 sum of f32 squares, eps 1e-6) then * weight (llama.cpp:312-318)."""
    sq = (x * x).astype(f32)
    mean = (sq.astype(np.float64).sum(axis=1) / x.shape[1]).astype(f32)
    scale = (f32(1.0) / np.sqrt(mean + f32(1e-6), dtype=f32)).astype(f32)
    return (w[None, :] * (x * scale[:, None]).astype(f32)).astype(f32)


def rope(x: np.ndarray, n_past: int, n_head: int) -> np.ndarray:
    """ggml_rope mode 0 (lib/ggml.c:8655-8682): theta = p, theta *= theta_scale per pair; gcc build's fma pattern
    d0 = fma(x0, cos, -(x1*sin)), d1 = fma(x0, sin, x1*cos) (oracle/_ref disassembly).  numpy has no fma: the
    products are formed in f64 (exact for f32 inputs) and rounded once, which is what an fma does."""
    N, E = x.shape
    D = E // n_head
    theta_scale = f32(_libm.powf(10000.0, float(f32(-2.0) / f32(D))))
    out = np.empty_like(x)
    xv = x.reshape(N, n_head, D // 2, 2)
    ov = out.reshape(N, n_head, D // 2, 2)
    for n in range(N):
        theta = f32(n_past + n)
        cs = np.empty(D // 2, f32)
        sn = np.empty(D // 2, f32)
        for i in range(D // 2):
            cs[i], sn[i] = _libm.cosf(float(theta)), _libm.sinf(float(theta))
            theta = f32(theta * theta_scale)
        x0, x1 = xv[n, :, :, 0], xv[n, :, :, 1]
        p1 = (x1 * sn[None, :]).astype(f32)
        p2 = (x1 * cs[None, :]).astype(f32)
        ov[n, :, :, 0] = (x0.astype(np.float64) * cs[None, :] - p1).astype(f32)
        ov[n, :, :, 1] = (x0.astype(np.float64) * sn[None, :] + p2).astype(f32)
    return out


def soft_max_rows(s: np.ndarray) -> np.ndarray:
    """ggml_soft_max (lib/ggml.c:8558-8580) on rows already masked with -inf: fp16 exp table, f64 sum."""
    exp_tab, _ = tables()
    mx = s.max(axis=-1, keepdims=True)
    with np.errstate(invalid="ignore", over="ignore"):
        h = (s - mx).astype(f32).astype(np.float16).view(np.uint16)
    val = np.where(np.isneginf(s), f32(0), exp_tab[h]).astype(f32)
    inv = (1.0 / val.astype(np.float64).sum(axis=-1, keepdims=True)).astype(f32)
    return (val * inv).astype(f32)


def silu(x: np.ndarray) -> np.ndarray:
    """ggml_vec_silu_f32 with GGML_SILU_FP16 (lib/ggml.c:3207-3215)."""
    _, silu_tab = tables()
    with np.errstate(over="ignore"):
        return silu_tab[x.astype(np.float16).view(np.uint16)]


class Weights:
    """Tensors as in a GGJT file: name -> (gtype, shape, data); 2-D data are AoS Q4 blocks uint8 [M, bytes]."""

    def __init__(self, cfg: dict, qtype: int, tensors: dict):
        self.cfg, self.qtype, self.t = cfg, qtype, tensors
        self.E, self.H, self.L, self.V = cfg["n_embd"], cfg["n_head"], cfg["n_layer"], cfg["n_vocab"]

    def q(self, name):
        return self.t[name][2]

    def f(self, name):
        return np.asarray(self.t[name][2], dtype=f32)


def slice_rows(wq: np.ndarray, r0: int, r1: int) -> np.ndarray:
    return np.ascontiguousarray(wq[r0:r1])


def slice_kblocks(wq: np.ndarray, qtype: int, kb0: int, kb1: int) -> np.ndarray:
    bs = BLOCK_BYTES[qtype]
    M = wq.shape[0]
    return np.ascontiguousarray(wq.reshape(M, -1, bs)[:, kb0:kb1, :]).reshape(M, -1)


class KV:
    def __init__(self, n_layer, n_ctx, width):
        self.k = np.zeros((n_layer, n_ctx, width), f32)
        self.v = np.zeros((n_layer, n_ctx, width), f32)   # (the reference stores V transposed; values are the same)


def layer_forward(w: Weights, kv: KV, il: int, x: np.ndarray, n_past: int, port: Port, tp_rank: int = 0, tp_size: int = 1,
                  allreduce=None, allgather=None):
    """One transformer layer of Model::eval (lib/llama.cpp:308-443) on the layer input x [N, E].  Returns the layer output and
    the f32 tensors whose Q8_0 quantization feeds the wo / w1|w3 / w2 matmuls ("att", "ffn_in", "act") -- the discrete
    intermediates in which a rounding flip of an implementation that sums in another order becomes visible."""
    E, H = w.E, w.H
    D = E // H
    G, r = tp_size, tp_rank
    El, Hl = E // G, H // G
    qt = w.qtype
    N = x.shape[0]
    if True:
        p = f"layers.{il}."
        cur = rms_norm_mul(x, w.f(p + "attention_norm.weight"))
        rows = slice(r * El, (r + 1) * El)
        q = port.mul_mat_q(qt, slice_rows(w.q(p + "attention.wq.weight"), rows.start, rows.stop), cur)
        k = port.mul_mat_q(qt, slice_rows(w.q(p + "attention.wk.weight"), rows.start, rows.stop), cur)
        v = port.mul_mat_q(qt, slice_rows(w.q(p + "attention.wv.weight"), rows.start, rows.stop), cur)
        q, k = rope(q, n_past, Hl), rope(k, n_past, Hl)
        kv.k[il, n_past:n_past + N] = k
        kv.v[il, n_past:n_past + N] = v
        P = n_past + N
        K = kv.k[il, :P].reshape(P, Hl, D)
        Vv = kv.v[il, :P].reshape(P, Hl, D)
        Q = q.reshape(N, Hl, D)
        scale = f32(1.0) / np.sqrt(f32(E) / f32(H), dtype=f32)
        att = np.empty((N, El), f32)
        for h in range(Hl):
            s = port.mul_mat_f32(K[:, h, :], Q[:, h, :])                     # ggml_mul_mat(K, Q) f32: [N, P], AVX2 lane order
            s = (s * scale).astype(f32)                                      # ggml_scale
            mask = np.arange(P)[None, :] > (n_past + np.arange(N))[:, None]  # diag_mask_inf
            s[mask] = -np.inf
            pr = soft_max_rows(s)
            vt = np.ascontiguousarray(Vv[:, h, :].T)                         # the reference's V view: D rows of P keys
            att[:, h * D:(h + 1) * D] = port.mul_mat_f32(vt, pr)             # ggml_mul_mat(V_trans, KQ_soft_max): [N, D]
        kb = (El // QK)
        if G > 1 and allgather is not None:
            # ROW split of wo (the reference's own split across threads, lib/ggml.c:8127-8135): the operand of every rank -- each rank
            # holds whole 32-element blocks of its features, so quantizing the gathered f32 rows equals gathering the Q8_0 blocks the
            # device exchanges -- then this rank's rows over the full K, then the rows of every rank.  Nothing is summed across ranks.
            part = allgather(port.mul_mat_q(qt, slice_rows(w.q(p + "attention.wo.weight"), r * El, (r + 1) * El), allgather(att)))
        else:
            part = port.mul_mat_q(qt, slice_kblocks(w.q(p + "attention.wo.weight"), qt, r * kb, (r + 1) * kb), att, strict=G == 1)
            if G > 1:
                part = allreduce(part)
        x2 = (part + x).astype(f32)                                          # inpFF = cur + inpSA
        cur = rms_norm_mul(x2, w.f(p + "ffn_norm.weight"))
        F = w.q(p + "feed_forward.w1.weight").shape[0]
        Fl = F // G
        h3 = port.mul_mat_q(qt, slice_rows(w.q(p + "feed_forward.w3.weight"), r * Fl, (r + 1) * Fl), cur)
        h1 = port.mul_mat_q(qt, slice_rows(w.q(p + "feed_forward.w1.weight"), r * Fl, (r + 1) * Fl), cur)
        hh = (silu(h1) * h3).astype(f32)
        kbf = Fl // QK
        if G > 1 and allgather is not None:
            part = allgather(port.mul_mat_q(qt, slice_rows(w.q(p + "feed_forward.w2.weight"), r * El, (r + 1) * El), allgather(hh)))
        else:
            part = port.mul_mat_q(qt, slice_kblocks(w.q(p + "feed_forward.w2.weight"), qt, r * kbf, (r + 1) * kbf), hh, strict=G == 1)
            if G > 1:
                part = allreduce(part)
        x = (part + x2).astype(f32)
    return x, dict(att=att, ffn_in=cur, act=hh)


def eval_tokens(w: Weights, kv: KV, tokens, n_past: int, port: Port | None = None, tp_rank: int = 0, tp_size: int = 1,
                allreduce=None, allgather=None) -> tuple[np.ndarray, np.ndarray]:
    """Model::eval for N tokens at n_past.  Returns (logits [N, V], embeddings [N, E] = final normed activations).
    With tp_size > 1 this computes rank `tp_rank`'s shard (heads / ffn rows split, wo / w2 column blocks) and calls
    allreduce(partial [N, E]) -> summed array after wo and w2 -- the exchange SURVEY.md 8(e) specifies.  With `allgather`
    (allgather(a [N, w]) -> [N, G w], rank order) wo / w2 are split by ROWS instead and nothing is summed: the reference-order
    mode's tensor parallelism, bit-identical to the unsharded eval."""
    port = port or Port()
    E, H, L = w.E, w.H, w.L
    D = E // H
    G, r = tp_size, tp_rank
    El, Hl = E // G, H // G
    qt = w.qtype
    N = len(tokens)
    x = np.stack([port.dequantize_row(qt, w.q("tok_embeddings.weight")[t], E) for t in tokens])   # get_rows_q
    for il in range(L):
        x, _ = layer_forward(w, kv, il, x, n_past, port, tp_rank, tp_size, allreduce, allgather)
    xn = rms_norm_mul(x, w.f("norm.weight"))
    logits = port.mul_mat_q(qt, w.q("output.weight"), xn)
    return logits, xn
