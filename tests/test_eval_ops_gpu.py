"""GPU: each non-matmul eval kernel (fastllama_amd/csrc/eval_kernels.hip) against the numpy restatement of the
reference op (oracle/llama_eval.py, itself pinned to the reference), at toy width and at LLaMA-7B width.
Integer outputs (Q8_0 quants) and table-driven values must be bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle
from oracle import llama_eval as le

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fast_mode")]   # (conftest.py: the fast kernels, explicitly)


@pytest.fixture(scope="module")
def env():
    import torch
    from fastllama_amd import hip, ops
    hip.require_device(0)
    return torch, hip, ops, hip.load(), oracle.Port()


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_fp16_tables_and_rope_table_bitexact(env):
    torch, hip, ops, L, port = env
    e = np.empty(1 << 16, np.uint16)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
    exp_tab, silu_tab = le.tables()
    ok = ~np.isnan(exp_tab)
    assert np.array_equal(e.view(np.float16).astype(np.float32)[ok], exp_tab[ok])
    ok = ~np.isnan(silu_tab)
    assert np.array_equal(s.view(np.float16).astype(np.float32)[ok], silu_tab[ok])
    for D in (32, 128):
        rt = np.empty((64, D // 2, 2), np.float32)
        L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), 64, D)
        x = np.zeros((64, D), np.float32)
        x[:, 0::2] = 1.0                                  # rope of (1, 0) pairs = (cos, sin)
        r = np.stack([le.rope(x[p:p + 1], p, 1)[0] for p in range(64)])
        assert np.array_equal(rt[:, :, 0], r[:, 0::2]) and np.array_equal(rt[:, :, 1], r[:, 1::2])


@pytest.mark.parametrize("N,E", [(1, 256), (5, 256), (40, 256), (3, 4096), (96, 4096), (20, 8192)])
def test_rmsnorm_quant_bitexact(env, N, E):
    torch, hip, ops, L, port = env
    rng = np.random.default_rng(E + N)
    x = (rng.standard_normal((N, E)) * rng.uniform(0.01, 3.0, (N, 1))).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(E)).astype(np.float32)
    want = le.rms_norm_mul(x, w)
    xd, wd = dev(torch, x), dev(torch, w)
    y = torch.empty((N, E), device="cuda")
    for layout in (1, 16):
        if layout == 1 and N > 8:
            continue
        a = ops.QAct(N, E)
        hip.check(L.fl_debug_rmsnorm_quant(xd.data_ptr(), E, wd.data_ptr(), N, E, y.data_ptr(), E, a.handle, layout, None))
        hip.check(L.fl_quantize_q8_layout(a.handle, y.data_ptr(), E, N, E, layout, None))   # sets N/layout bookkeeping
        a.N, a.K = N, E
        hip.check(L.fl_debug_rmsnorm_quant(xd.data_ptr(), E, wd.data_ptr(), N, E, y.data_ptr(), E, a.handle, layout, None))
        assert np.array_equal(y.cpu().numpy().view(np.uint32), want.view(np.uint32)), (N, E, layout)
        q8 = a.export().cpu().numpy()
        assert np.array_equal(q8, np.stack([port.quantize_row_q8_0(r) for r in want]))


@pytest.mark.parametrize("N,F", [(2, 704), (33, 704), (96, 11008)])
def test_silu_mul_quant_bitexact(env, N, F):
    torch, hip, ops, L, port = env
    rng = np.random.default_rng(F)
    h = (rng.standard_normal((N, 2 * F)) * 2).astype(np.float32)
    want = (le.silu(h[:, :F]) * h[:, F:]).astype(np.float32)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
    sd, hd = dev(torch, s.view(np.int16)), dev(torch, h)
    layout = 1 if N <= 8 else 16
    a = ops.QAct(N, F)
    hip.check(L.fl_quantize_q8_layout(a.handle, hd.data_ptr(), 2 * F, N, F, layout, None))
    a.N, a.K = N, F
    hip.check(L.fl_debug_silu_mul_quant(hd.data_ptr(), 2 * F, N, F, sd.data_ptr(), a.handle, layout, None))
    assert np.array_equal(a.export().cpu().numpy(), np.stack([port.quantize_row_q8_0(r) for r in want]))


@pytest.mark.parametrize("N,E,D,n_past", [(1, 128, 32, 7), (20, 256, 32, 5), (64, 4096, 128, 30)])
def test_rope_kv_bitexact(env, N, E, D, n_past):
    torch, hip, ops, L, port = env
    n_ctx = 128
    rng = np.random.default_rng(N * E)
    qkv = rng.standard_normal((N, 3 * E)).astype(np.float32)
    rt = np.empty((n_ctx, D // 2, 2), np.float32)
    L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), n_ctx, D)
    qd, rd = dev(torch, qkv), dev(torch, rt)
    kc = torch.zeros((n_ctx, E), device="cuda")
    vc = torch.zeros((E, n_ctx), device="cuda")
    hip.check(L.fl_debug_rope_kv(qd.data_ptr(), 3 * E, N, E, D, n_past, n_ctx, rd.data_ptr(), kc.data_ptr(), vc.data_ptr(), None))
    H = E // D
    want_q = le.rope(qkv[:, :E], n_past, H)
    want_k = le.rope(qkv[:, E:2 * E], n_past, H)
    out = qd.cpu().numpy()
    assert np.array_equal(out[:, :E].view(np.uint32), want_q.view(np.uint32))
    assert np.array_equal(kc.cpu().numpy()[n_past:n_past + N].view(np.uint32), want_k.view(np.uint32))
    assert np.array_equal(vc.cpu().numpy()[:, n_past:n_past + N], qkv[:, 2 * E:].T)
    assert not kc.cpu().numpy()[:n_past].any() and not vc.cpu().numpy()[:, n_past + N:].any()


@pytest.mark.parametrize("N,P0,D,H", [(1, 17, 32, 4), (20, 0, 32, 8), (96, 0, 128, 32), (40, 23, 128, 4)])
def test_attention_scores_softmax_kqv(env, N, P0, D, H):
    """KQ -> scale -> mask -> soft_max (fp16 table, f64 sum) -> KQV against numpy; softmax given identical
    scores must be bit-exact, the two f32 GEMMs agree to f32 round-off (MFMA k-order chain vs BLAS)."""
    torch, hip, ops, L, port = env
    n_ctx, E, P = 128, H * D, P0 + N
    rng = np.random.default_rng(N + D + H)
    q = rng.standard_normal((N, E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32)
    vc = np.zeros((E, n_ctx), np.float32)
    kc[:P] = rng.standard_normal((P, E))
    vc[:, :P] = rng.standard_normal((E, P))
    e = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    ed = dev(torch, e.view(np.int16))
    qd, kd, vd = dev(torch, q), dev(torch, kc), dev(torch, vc)
    att = torch.full((H, N, n_ctx), 7.0, device="cuda")       # stale garbage must never leak into results
    scale = np.float32(1.0) / np.sqrt(np.float32(D))
    hip.check(L.fl_debug_gemm_f32_abt(qd.data_ptr(), E, D, kd.data_ptr(), E, D, att.data_ptr(), n_ctx, N * n_ctx, N, P, D, H,
                                      float(scale), 1, P0, None))
    s_gpu = att.cpu().numpy()[:, :, :P].copy()
    mask = np.arange(P)[None, :] > (P0 + np.arange(N))[:, None]
    for h in range(H):
        want = ((q[:, h * D:(h + 1) * D] @ kc[:P, h * D:(h + 1) * D].T).astype(np.float32) * scale).astype(np.float32)
        got = s_gpu[h]
        assert np.max(np.abs(got - want)[~mask]) <= 1e-5 * np.max(np.abs(want))
    hip.check(L.fl_debug_softmax_rows(att.data_ptr(), n_ctx, N * n_ctx, N, P, P0, H, ed.data_ptr(), None))
    p_gpu = att.cpu().numpy()[:, :, :P]
    for h in range(H):
        s = s_gpu[h].copy()
        s[mask] = -np.inf
        assert np.array_equal(p_gpu[h].view(np.uint32), le.soft_max_rows(s).view(np.uint32))   # bit-exact softmax
    ao = torch.zeros((N, E), device="cuda")
    hip.check(L.fl_debug_gemm_f32_abt(att.data_ptr(), n_ctx, N * n_ctx, vd.data_ptr(), n_ctx, D * n_ctx, ao.data_ptr(), E, D,
                                      N, D, P, H, 1.0, 2, P0, None))
    got = ao.cpu().numpy()
    for h in range(H):
        want = (p_gpu[h] @ vc[h * D:(h + 1) * D, :P].T).astype(np.float32)
        assert np.max(np.abs(got[:, h * D:(h + 1) * D] - want)) <= 1e-5 * np.max(np.abs(want))


@pytest.mark.parametrize("qtype", [2, 3])
@pytest.mark.parametrize("M,K", [(48, 64), (300, 256), (12288, 4096), (32000, 4096), (4096, 4096), (1024, 8192)])
def test_gemv_norm_fused_equals_unfused(env, qtype, M, K):
    """decode: rms_norm*w -> Q8_0 -> mul_mat in ONE launch must equal the three-kernel sequence bit for bit, and the
    Q8_0 it builds in LDS is the oracle's quantize_row_q8_0 of the oracle's norm (checked through the unfused path)."""
    torch, hip, ops, L, port = env
    rng = np.random.default_rng(M + K + qtype)
    wq = port.quantize_q4(qtype, (rng.standard_normal((M, K)) * 0.05).astype(np.float32))
    W = ops.QTensor(qtype, wq, M, K)
    x = (rng.standard_normal((1, K)) * 1.7).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    xd, nd = dev(torch, x), dev(torch, nw)
    a = ops.QAct(1, K)
    yn0 = torch.empty((1, K), device="cuda")
    hip.check(L.fl_quantize_q8_layout(a.handle, xd.data_ptr(), K, 1, K, 1, None))
    a.N, a.K = 1, K
    hip.check(L.fl_debug_rmsnorm_quant(xd.data_ptr(), K, nd.data_ptr(), 1, K, yn0.data_ptr(), K, a.handle, 1, None))
    y0 = torch.empty(M, device="cuda")
    hip.check(L.fl_debug_mul_mat_q(W.handle, a.handle, y0.data_ptr(), M, 2, None))
    y1 = torch.full((M,), 3.0, device="cuda")
    yn1 = torch.zeros((1, K), device="cuda")
    hip.check(L.fl_debug_gemv_norm(W.handle, xd.data_ptr(), nd.data_ptr(), yn1.data_ptr(), y1.data_ptr(), None))
    assert np.array_equal(y1.cpu().numpy().view(np.uint32), y0.cpu().numpy().view(np.uint32))
    assert np.array_equal(yn1.cpu().numpy().view(np.uint32), yn0.cpu().numpy().view(np.uint32))
    want = port.mul_mat_q(qtype, wq, le.rms_norm_mul(x, nw))[0]
    assert np.max(np.abs(y1.cpu().numpy() - want)) <= 2e-5 * np.max(np.abs(want))


@pytest.mark.parametrize("D,H,n_past", [(32, 4, 0), (32, 4, 9), (128, 32, 0), (128, 32, 1), (128, 32, 130), (128, 8, 511),
                                        (64, 5, 37)])
def test_decode_attention_fused(env, D, H, n_past):
    """single-token attention in one launch: rope + KV store bit-exact, probabilities/outputs against numpy
    (oracle/llama_eval.py ops), Q8_0 output equal to quantize_row_q8_0 of the kernel's own f32 result semantics."""
    torch, hip, ops, L, port = env
    n_ctx, E, P = 512, H * D, n_past + 1
    rng = np.random.default_rng(D + H + n_past)
    qkv = rng.standard_normal((1, 3 * E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32)
    vc = np.zeros((E, n_ctx), np.float32)
    kc[:n_past] = rng.standard_normal((n_past, E))
    vc[:, :n_past] = rng.standard_normal((E, n_past))
    e = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    rt = np.empty((n_ctx, D // 2, 2), np.float32)
    L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), n_ctx, D)
    ed, rd, qd, kd, vd = dev(torch, e.view(np.int16)), dev(torch, rt), dev(torch, qkv), dev(torch, kc), dev(torch, vc)
    a = ops.QAct(1, E)
    hip.check(L.fl_quantize_q8_layout(a.handle, qd.data_ptr(), 3 * E, 1, E, 1, None))
    a.N, a.K = 1, E
    scale = np.float32(1.0) / np.sqrt(np.float32(D))
    hip.check(L.fl_debug_decode_attention(qd.data_ptr(), E, D, H, n_past, n_ctx, rd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                          ed.data_ptr(), float(scale), a.handle, None))
    q_r = le.rope(qkv[:, :E], n_past, H)
    k_r = le.rope(qkv[:, E:2 * E], n_past, H)
    kc2, vc2 = kd.cpu().numpy(), vd.cpu().numpy()
    assert np.array_equal(kc2[n_past].view(np.uint32), k_r[0].view(np.uint32))
    assert np.array_equal(vc2[:, n_past], qkv[0, 2 * E:])
    assert np.array_equal(kc2[:n_past], kc[:n_past]) and np.array_equal(vc2[:, :n_past], vc[:, :n_past])
    assert not kc2[P:].any() and not vc2[:, P:].any()
    assert np.array_equal(qd.cpu().numpy(), qkv)                       # inputs untouched
    want = np.empty((1, E), np.float32)
    for h in range(H):
        sl = slice(h * D, (h + 1) * D)
        s = ((q_r[:, sl] @ kc2[:P, sl].T).astype(np.float32) * scale).astype(np.float32)
        p = le.soft_max_rows(s)
        want[:, sl] = (p @ vc2[sl, :P].T).astype(np.float32)
    q8 = a.export().cpu().numpy()[0].reshape(-1, 40)                   # block_q8_0: d, s, 32 x int8
    d = q8[:, :4].copy().view(np.float32)[:, 0]
    qs = q8[:, 8:].view(np.int8).astype(np.float32)
    got = (qs * d[:, None]).reshape(-1)
    wq8 = port.quantize_row_q8_0(want[0]).reshape(-1, 40)
    wd = wq8[:, :4].copy().view(np.float32)[:, 0]
    assert np.max(np.abs(d - wd)) <= 1e-5 * np.max(np.abs(wd))
    assert np.max(np.abs(got - want[0])) <= np.max(wd) * 1.01          # within one quantization step everywhere
    assert np.mean(q8[:, 8:].view(np.int8) == wq8[:, 8:].view(np.int8)) > 0.98   # only round-off flips of f32 dots
    s_field = q8[:, 4:8].copy().view(np.float32)[:, 0]
    assert np.array_equal(s_field, (d * qs.reshape(-1, 32).sum(1).astype(np.float32)).astype(np.float32))


@pytest.mark.parametrize("dyn", [False, True])
@pytest.mark.parametrize("D,H,n_past,n_ctx", [(32, 4, 0, 512), (64, 5, 37, 512), (128, 32, 255, 1024), (128, 32, 256, 1024),
                                              (128, 8, 700, 1024), (128, 4, 2046, 2048), (128, 3, 1023, 1024), (96, 2, 515, 1024),
                                              (64, 3, 97, 100), (128, 2, 290, 292), (32, 2, 99, 100)])
def test_decode_attention_split_equals_fused(env, D, H, n_past, n_ctx, dyn):
    """long-context decode attention (scores over (head, 128-position slice) workgroups, then soft_max + KQV + Q8_0 over
    (head, 32 features)) == the one-launch kernel, bit for bit: caches, and the Q8_0 operand of the wo matmul; with the
    position as an argument and read from device memory (the hipGraph form, grid sized for n_ctx)."""
    torch, hip, ops, L, port = env
    E = H * D
    rng = np.random.default_rng(D + H + n_past)
    qkv = rng.standard_normal((1, 3 * E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32)
    vc = np.zeros((E, n_ctx), np.float32)
    kc[:n_past] = rng.standard_normal((n_past, E))
    vc[:, :n_past] = rng.standard_normal((E, n_past))
    vc[:, n_past:] = 7.0                                     # stale values beyond the position must not leak in
    e = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    rt = np.empty((n_ctx, D // 2, 2), np.float32)
    L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), n_ctx, D)
    ed, rd, qd = dev(torch, e.view(np.int16)), dev(torch, rt), dev(torch, qkv)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    res = []
    for split in (False, True):
        kd, vd = dev(torch, kc), dev(torch, vc)
        a = ops.QAct(1, E)
        hip.check(L.fl_quantize_q8_layout(a.handle, qd.data_ptr(), 3 * E, 1, E, 1, None))
        a.N, a.K = 1, E
        if split:
            sc = torch.full((H, n_ctx), float("nan"), device="cuda")
            pd = torch.tensor([n_past], dtype=torch.int32, device="cuda")
            hip.check(L.fl_debug_decode_attention_split(qd.data_ptr(), E, D, H, 0 if dyn else n_past, n_ctx, rd.data_ptr(),
                                                        kd.data_ptr(), vd.data_ptr(), ed.data_ptr(), scale, sc.data_ptr(),
                                                        a.handle, pd.data_ptr() if dyn else None, None))
        else:
            hip.check(L.fl_debug_decode_attention(qd.data_ptr(), E, D, H, n_past, n_ctx, rd.data_ptr(), kd.data_ptr(),
                                                  vd.data_ptr(), ed.data_ptr(), scale, a.handle, None))
        torch.cuda.synchronize()
        res.append((kd.cpu().numpy(), vd.cpu().numpy(), a.export().cpu().numpy().copy()))
    (k0, v0, q0), (k1, v1, q1) = res
    assert np.array_equal(k0.view(np.uint32), k1.view(np.uint32))
    assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    assert np.array_equal(q0, q1)
    assert np.any(q0.reshape(-1, 40)[:, 8:] != 0)


@pytest.mark.parametrize("qtype", [2, 3])
@pytest.mark.parametrize("M,F", [(48, 64), (256, 704), (4096, 11008), (5120, 13824)])
def test_gemv_silu_fused_equals_unfused(env, qtype, M, F):
    """decode: silu(w1 x)*(w3 x) -> Q8_0 -> mul_mat (+ residual) in ONE launch == silu_mul_quant + gemv, bit for bit."""
    torch, hip, ops, L, port = env
    rng = np.random.default_rng(M + F + qtype)
    wq = port.quantize_q4(qtype, (rng.standard_normal((M, F)) * 0.05).astype(np.float32))
    W = ops.QTensor(qtype, wq, M, F)
    h = (rng.standard_normal((1, 2 * F)) * 2).astype(np.float32)
    res = rng.standard_normal(M).astype(np.float32)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
    sd, hd, rd = dev(torch, s.view(np.int16)), dev(torch, h), dev(torch, res)
    a = ops.QAct(1, F)
    hip.check(L.fl_quantize_q8_layout(a.handle, hd.data_ptr(), 2 * F, 1, F, 1, None))
    a.N, a.K = 1, F
    hip.check(L.fl_debug_silu_mul_quant(hd.data_ptr(), 2 * F, 1, F, sd.data_ptr(), a.handle, 1, None))
    y0 = torch.empty(M, device="cuda")
    hip.check(L.fl_debug_mul_mat_q(W.handle, a.handle, y0.data_ptr(), M, 2, None))
    y1 = torch.full((M,), 3.0, device="cuda")
    hip.check(L.fl_debug_gemv_silu(W.handle, hd.data_ptr(), sd.data_ptr(), y1.data_ptr(), None, None))
    assert np.array_equal(y1.cpu().numpy().view(np.uint32), y0.cpu().numpy().view(np.uint32))
    hip.check(L.fl_debug_gemv_silu(W.handle, hd.data_ptr(), sd.data_ptr(), y1.data_ptr(), rd.data_ptr(), None))
    assert np.array_equal(y1.cpu().numpy(), y0.cpu().numpy() + res)
    want = port.mul_mat_q(qtype, wq, (le.silu(h[:, :F]) * h[:, F:]).astype(np.float32))[0]
    assert np.max(np.abs(y0.cpu().numpy() - want)) <= 2e-5 * np.max(np.abs(want))


@pytest.mark.parametrize("N,P0,D,H,n_ctx", [(20, 0, 32, 8, 128), (96, 0, 128, 32, 128), (40, 23, 128, 4, 128), (33, 7, 64, 3, 64),
                                            (512, 0, 128, 32, 1024), (100, 690, 128, 2, 1024), (9, 0, 96, 2, 64), (480, 0, 128, 3, 512), (300, 200, 128, 2, 512)])
def test_prefill_attention_fused_equals_three_kernels(env, N, P0, D, H, n_ctx):
    """KQ*scale + mask + soft_max + KQV in ONE launch (score rows in LDS) == gemm_f32_abt -> softmax_rows -> gemm_f32_abt,
    bit for bit (same MFMA sequence per output), on every valid output row."""
    torch, hip, ops, L, port = env
    E, P = H * D, P0 + N
    rng = np.random.default_rng(N + D + H + P0)
    qkv = rng.standard_normal((N, 3 * E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32)
    vc = np.zeros((E, n_ctx), np.float32)
    kc[:P] = rng.standard_normal((P, E))
    vc[:, :P] = rng.standard_normal((E, P))
    e = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    ed, qd, kd, vd = dev(torch, e.view(np.int16)), dev(torch, qkv), dev(torch, kc), dev(torch, vc)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    att = torch.full((H, N, n_ctx), 7.0, device="cuda")
    ao0 = torch.zeros((N, E), device="cuda")
    hip.check(L.fl_debug_gemm_f32_abt(qd.data_ptr(), 3 * E, D, kd.data_ptr(), E, D, att.data_ptr(), n_ctx, N * n_ctx, N, P, D, H,
                                      scale, 1, P0, None))
    hip.check(L.fl_debug_softmax_rows(att.data_ptr(), n_ctx, N * n_ctx, N, P, P0, H, ed.data_ptr(), None))
    hip.check(L.fl_debug_gemm_f32_abt(att.data_ptr(), n_ctx, N * n_ctx, vd.data_ptr(), n_ctx, D * n_ctx, ao0.data_ptr(), E, D,
                                      N, D, P, H, 1.0, 2, P0, None))
    ao1 = torch.full((N, E), -3.0, device="cuda")
    hip.check(L.fl_debug_prefill_attention(qd.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kd.data_ptr(), vd.data_ptr(), ed.data_ptr(),
                                           scale, ao1.data_ptr(), E, None, None))
    torch.cuda.synchronize()
    assert torch.equal(ao1, ao0)
    # the same launch writing the Q8_0 operand of the wo matmul instead: == quantize_row_q8_0 of the f32 rows, and zero
    # blocks for the padding columns N..N16-1
    if N >= 9:
        a = ops.QAct(N, E)
        hip.check(L.fl_quantize_q8_layout(a.handle, ao0.data_ptr(), E, N, E, 16, None))      # sizes / layout bookkeeping
        a.N, a.K = N, E
        junk = torch.full((N, E), 5.0, device="cuda")
        hip.check(L.fl_quantize_q8_layout(a.handle, junk.data_ptr(), E, N, E, 16, None))     # poison the workspace
        hip.check(L.fl_debug_prefill_attention(qd.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kd.data_ptr(), vd.data_ptr(),
                                               ed.data_ptr(), scale, ao1.data_ptr(), E, a.handle, None))
        got_q8 = a.export().cpu().numpy()
        want_q8 = np.stack([port.quantize_row_q8_0(r_) for r_ in ao0.cpu().numpy()])
        assert np.array_equal(got_q8, want_q8)
    # and against numpy on a few rows (the three-kernel path is itself checked against numpy above)
    for n in (0, N - 1):
        for h in (0, H - 1):
            sl = slice(h * D, (h + 1) * D)
            Lr = P0 + n + 1
            s = ((qkv[n:n + 1, sl] @ kc[:Lr, sl].T).astype(np.float32) * np.float32(scale)).astype(np.float32)
            want = (le.soft_max_rows(s) @ vc[sl, :Lr].T).astype(np.float32)[0]
            got = ao1[n, sl].cpu().numpy()
            assert np.max(np.abs(got - want)) <= 2e-5 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize("N,P0,D,H,n_ctx", [(512, 512, 128, 8, 1024), (512, 1536, 128, 4, 2048), (100, 1900, 128, 2, 2048),
                                            (33, 7, 64, 3, 64), (300, 203, 128, 2, 512), (9, 0, 96, 2, 64), (77, 530, 32, 5, 640),
                                            (512, 0, 128, 4, 512), (256, 3000, 128, 2, 4096)])
def test_prefill_attention_deep_equals_three_kernels(env, N, P0, D, H, n_ctx):
    """The key-tiled form of the prefill attention launch (deep contexts: a block's score rows do not fit LDS, they take one
    trip through a scratch buffer) == gemm_f32_abt -> softmax_rows -> gemm_f32_abt bit for bit, f32 rows and the Q8_0 operand,
    for chunk-boundary cases (n_past not a multiple of 8 / 32 / 256, N not a multiple of 32, head_dim 32..128, 4096 keys)."""
    torch, hip, ops, L, port = env
    E, P = H * D, P0 + N
    rng = np.random.default_rng(N + D + H + P0)
    qkv = rng.standard_normal((N, 3 * E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32)
    vc = np.zeros((E, n_ctx), np.float32)
    kc[:P] = rng.standard_normal((P, E))
    vc[:, :P] = rng.standard_normal((E, P))
    e = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    ed, qd, kd, vd = dev(torch, e.view(np.int16)), dev(torch, qkv), dev(torch, kc), dev(torch, vc)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    att = torch.full((H, N, n_ctx), 7.0, device="cuda")
    ao0 = torch.zeros((N, E), device="cuda")
    hip.check(L.fl_debug_gemm_f32_abt(qd.data_ptr(), 3 * E, D, kd.data_ptr(), E, D, att.data_ptr(), n_ctx, N * n_ctx, N, P, D, H,
                                      scale, 1, P0, None))
    hip.check(L.fl_debug_softmax_rows(att.data_ptr(), n_ctx, N * n_ctx, N, P, P0, H, ed.data_ptr(), None))
    hip.check(L.fl_debug_gemm_f32_abt(att.data_ptr(), n_ctx, N * n_ctx, vd.data_ptr(), n_ctx, D * n_ctx, ao0.data_ptr(), E, D,
                                      N, D, P, H, 1.0, 2, P0, None))
    scratch = torch.full((H, N, n_ctx), float("nan"), device="cuda")
    guard = scratch.clone()
    ao1 = torch.full((N, E), -3.0, device="cuda")
    hip.check(L.fl_debug_prefill_attention_scratch(scratch.data_ptr(), n_ctx, N * n_ctx))
    try:
        hip.check(L.fl_debug_prefill_attention(qd.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kd.data_ptr(), vd.data_ptr(), ed.data_ptr(),
                                               scale, ao1.data_ptr(), E, None, None))
        torch.cuda.synchronize()
        assert torch.equal(ao1, ao0)
        del guard
        if N >= 9:
            a = ops.QAct(N, E)
            junk = torch.full((N, E), 5.0, device="cuda")
            hip.check(L.fl_quantize_q8_layout(a.handle, junk.data_ptr(), E, N, E, 16, None))     # sizes, layout; poison
            a.N, a.K = N, E
            hip.check(L.fl_debug_prefill_attention(qd.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kd.data_ptr(), vd.data_ptr(),
                                                   ed.data_ptr(), scale, ao1.data_ptr(), E, a.handle, None))
            got_q8 = a.export().cpu().numpy()
            want_q8 = np.stack([port.quantize_row_q8_0(r_) for r_ in ao0.cpu().numpy()])
            assert np.array_equal(got_q8, want_q8)
    finally:
        hip.check(L.fl_debug_prefill_attention_scratch(None, 0, 0))


@pytest.mark.parametrize("qtype", [2, 3])
@pytest.mark.parametrize("F,K", [(64, 64), (704, 256), (11008, 4096)])
def test_gemv_pair_silu_epilogue_and_quant_prologue(env, qtype, F, K):
    """decode feed-forward on the woven w1|w3 matrix: (a) norm prologue + two dots + silu*mul epilogue in ONE launch ==
    the plain norm-GEMV followed by silu*mul on the woven output, bit for bit; (b) the w2 GEMV that quantizes the f32
    activation in its prologue == quantize_q8 + GEMV."""
    torch, hip, ops, L, port = env
    rng = np.random.default_rng(F + K + qtype)
    w1 = port.quantize_q4(qtype, (rng.standard_normal((F, K)) * 0.05).astype(np.float32))
    w3 = port.quantize_q4(qtype, (rng.standard_normal((F, K)) * 0.05).astype(np.float32))
    rb = w1.shape[1]
    woven = np.empty((2 * F, rb), np.uint8)
    wv = woven.reshape(F // 16, 2, 16, rb)
    wv[:, 0] = w1.reshape(F // 16, 16, rb)
    wv[:, 1] = w3.reshape(F // 16, 16, rb)
    W = ops.QTensor(qtype, woven, 2 * F, K)
    x = (rng.standard_normal((1, K)) * 1.3).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
    xd, nd, sd = dev(torch, x), dev(torch, nw), dev(torch, s.view(np.int16))
    y = torch.empty(2 * F, device="cuda")
    act = torch.full((F,), 9.0, device="cuda")
    # the wave count per row group (chosen from the number of workgroups) fixes the order of the partial sums: pin it
    # so that both launches add in the same order
    L.fl_debug_set(1, 8)
    try:
        hip.check(L.fl_debug_gemv_norm(W.handle, xd.data_ptr(), nd.data_ptr(), None, y.data_ptr(), None))
        hip.check(L.fl_debug_gemv_norm_silu(W.handle, xd.data_ptr(), nd.data_ptr(), sd.data_ptr(), act.data_ptr(), None))
    finally:
        L.fl_debug_set(1, 0)
    yv = y.cpu().numpy().reshape(F // 16, 2, 16)
    want = (le.silu(yv[:, 0].reshape(1, F)) * yv[:, 1].reshape(1, F)).astype(np.float32)[0]
    assert np.array_equal(act.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # (b) w2: [E rows][F]
    E = 48 if F == 64 else 256
    w2 = port.quantize_q4(qtype, (rng.standard_normal((E, F)) * 0.05).astype(np.float32))
    W2 = ops.QTensor(qtype, w2, E, F)
    res = rng.standard_normal(E).astype(np.float32)
    rd = dev(torch, res)
    a = ops.QAct(1, F)
    hip.check(L.fl_quantize_q8_layout(a.handle, act.data_ptr(), F, 1, F, 1, None))
    a.N, a.K = 1, F
    y0 = torch.empty(E, device="cuda")
    hip.check(L.fl_debug_mul_mat_q(W2.handle, a.handle, y0.data_ptr(), E, 2, None))
    y1 = torch.empty(E, device="cuda")
    hip.check(L.fl_debug_gemv_quant(W2.handle, act.data_ptr(), y1.data_ptr(), rd.data_ptr(), None))
    assert np.array_equal(y1.cpu().numpy(), y0.cpu().numpy() + res)
