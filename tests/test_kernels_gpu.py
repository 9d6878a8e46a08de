"""GPU parity tests of the HIP path, called through the C-ABI (libfastllama_hip.so) and checked
against the oracle (oracle/, CPU) and the committed golden vectors made by the reference itself.

Bars: bit-exact for everything integer/byte/scale (Q8_0 quantization, repack round trip, dequant);
float dots within 1e-5 of max|ref| (only the f32 summation ORDER differs from the reference; the
north-star budget on logits is 1e-3).
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from util import bits, golden, make_weights, make_x, rel_err_rows, rel_max_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fast_mode")]   # (conftest.py: the fast kernels, explicitly)

TOL = 1e-5
Q4 = [("q40", oracle.Q4_0), ("q41", oracle.Q4_1)]


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU; there is no CPU fallback to test"
    return torch


@pytest.fixture(scope="module")
def ops(torch):
    from fastllama_amd import hip, ops
    hip.require_device(0)
    return ops


@pytest.fixture(scope="module")
def port():
    return oracle.Port()


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_device_is_gfx950(torch, ops):
    from fastllama_amd import hip
    buf = C.create_string_buffer(256)
    hip.check(hip.load().fl_device_name(buf, 256))
    assert b"gfx950" in buf.value, buf.value


# ------------------------------------------------------------------ a4 quantize_row_q8_0 -----------
@pytest.mark.parametrize("tag", ["s", "m"])
def test_quantize_row_q8_0_matches_golden_bitexact(torch, ops, tag):
    g = golden()
    for n, row in enumerate(g[f"{tag}_x"]):
        got = ops.quantize_row_q8_0(dev(torch, row)).cpu().numpy()
        assert np.array_equal(got, g[f"{tag}_q8"][n]), (tag, n)


def test_quantize_row_q8_0_random_and_edges_bitexact(torch, ops, port):
    rng = np.random.default_rng(3)
    for K, scale in [(64, 1.0), (4096, 1.0), (11008, 30.0), (4096, 1e-30), (4096, 1e30)]:
        x = (rng.normal(0, 1, K) * scale).astype(np.float32)
        x[:32] = 0
        x[40] = -x[40]
        got = ops.quantize_row_q8_0(dev(torch, x)).cpu().numpy()
        assert np.array_equal(got, port.quantize_row_q8_0(x)), (K, scale)
    # exact ties: x*id = k + 0.5 must round half to EVEN (_mm256_round_ps, lib/ggml.c:1375-1378)
    x = np.zeros(64, dtype=np.float32)
    x[0] = 127.0
    x[1:9] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5, 126.5]
    got = ops.quantize_row_q8_0(dev(torch, x)).cpu().numpy()
    assert np.array_equal(got, port.quantize_row_q8_0(x))
    q = got[8:40].view(np.int8)
    assert list(q[:9]) == [127, 0, 2, 2, 0, -2, -2, 4, 126]


@pytest.mark.parametrize("N", [1, 2, 5, 8, 9, 16, 33])
@pytest.mark.parametrize("layout", [1, 16])
def test_internal_q8_workspaces_export_bitexact(torch, ops, port, N, layout):
    """The QA1 / QA16 device layouts that feed the matmul kernels hold exactly the reference's Q8_0."""
    K = 704
    x = make_x(N, K, 11 + N)
    x[0, 32:64] = 0
    a = ops.QAct(N, K).quantize(dev(torch, x), layout=layout)
    got = a.export().cpu().numpy()
    want = np.stack([port.quantize_row_q8_0(r) for r in x])
    assert np.array_equal(got, want)


def test_quantize_strided_rows(torch, ops, port):
    K, N = 256, 4
    big = dev(torch, make_x(N, 2 * K, 5))
    view = big[:, K:]                      # row stride 2K, 16-byte aligned
    a = ops.QAct(N, K).quantize(view)
    want = np.stack([port.quantize_row_q8_0(r) for r in view.cpu().numpy()])
    assert np.array_equal(a.export().cpu().numpy(), want)


# ------------------------------------------------------------------ a7 dequantize ------------------
@pytest.mark.parametrize("nm,qt", Q4)
def test_dequantize_row_matches_golden_bitexact(torch, ops, nm, qt):
    g = golden()
    K = g["s_w"].shape[1]
    for m, row in enumerate(g[f"s_{nm}"]):
        got = ops.dequantize_row_q(qt, dev(torch, row), K).cpu().numpy()
        assert np.array_equal(bits(got), bits(g[f"s_{nm}_deq"][m])), m


@pytest.mark.parametrize("nm,qt", Q4)
def test_dequantize_whole_matrix_bitexact(torch, ops, port, nm, qt):
    M, K = 33, 4096
    wq = make_weights(port, qt, M, K, 21)
    got = ops.dequantize_row_q(qt, dev(torch, wq), M * K).cpu().numpy().reshape(M, K)
    assert np.array_equal(bits(got), bits(port.dequantize(qt, wq, K)))


# ------------------------------------------------------------------ a1/a2 repack -------------------
@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K", [(16, 64), (48, 256), (37, 704), (4000, 4096), (1, 32)])
def test_qtensor_repack_round_trip_is_lossless(torch, ops, port, nm, qt, M, K):
    wq = make_weights(port, qt, M, K, M + K)
    W = ops.QTensor(qt, wq, M, K)
    assert np.array_equal(W.download(), wq)
    Wd = ops.QTensor(qt, dev(torch, wq), M, K)   # from a device-resident AoS payload
    assert np.array_equal(Wd.download(), wq)


def test_qtensor_rejects_bad_arguments(torch, ops, port):
    from fastllama_amd import hip
    with pytest.raises(hip.FastLlamaHipError):
        ops.QTensor(7, np.zeros(20, np.uint8), 1, 32)          # not a Q4 type
    L = hip.load()
    assert not L.fl_qtensor_upload(oracle.Q4_0, None, 4, 48, None)   # K % 32 != 0
    assert b"K" in L.fl_last_error()
    # a Q4_0 scale below 2^-122 cannot be divided by 16 exactly -> rejected, not silently rounded
    wq = make_weights(port, oracle.Q4_0, 16, 64, 1)
    wq[3, 0:4] = np.array([1e-38], dtype=np.float32).view(np.uint8)
    with pytest.raises(hip.FastLlamaHipError):
        ops.QTensor(oracle.Q4_0, wq, 16, 64)


# ------------------------------------------------------------------ a5/a6 vec_dot ------------------
@pytest.mark.parametrize("tag", ["s", "m"])
@pytest.mark.parametrize("nm,qt", Q4)
def test_vec_dot_matches_golden(torch, ops, tag, nm, qt):
    g = golden()
    wq, q8, ref = g[f"{tag}_{nm}"], g[f"{tag}_q8"], g[f"{tag}_{nm}_vd"]
    K = g[f"{tag}_x"].shape[1]
    rows = range(0, wq.shape[0], 5)
    finite = np.isfinite(ref)
    for n in range(q8.shape[0]):
        if not finite[n].all():
            continue                      # the 3e30 row overflows in the reference too; checked below
        got = np.array([ops.vec_dot_q(qt, K, dev(torch, wq[m]), dev(torch, q8[n])).item() for m in rows])
        assert rel_max_err(got, ref[n, list(rows)]) <= TOL, (tag, nm, n)


def test_quantize_fns_table_mirrors_reference_vtable(torch, ops, port):
    """fl_get_quantize_fn(type) has the reference's five slots and void signatures (ggml.h:850-862)."""
    from fastllama_amd import hip
    L = hip.load()
    K = 256
    x = make_x(1, K, 9)[0]
    for qt in (oracle.Q4_0, oracle.Q4_1):
        f = L.fl_get_quantize_fn(qt)
        xd = dev(torch, x)
        for slot, reference in ((f.quantize_row_q, False), (f.quantize_row_q_reference, True)):
            wb = torch.empty(K // 32 * oracle.BLOCK_BYTES[qt], dtype=torch.uint8, device="cuda")
            slot(xd.data_ptr(), wb.data_ptr(), K)
            assert np.array_equal(wb.cpu().numpy(), port.quantize_row_q4(qt, x, reference))
        yq = torch.empty(K // 32 * 40, dtype=torch.uint8, device="cuda")
        f.quantize_row_q_dot(xd.data_ptr(), yq.data_ptr(), K)
        assert np.array_equal(yq.cpu().numpy(), port.quantize_row_q8_0(x))
        wq = make_weights(port, qt, 1, K, 4)[0]
        wd = dev(torch, wq)
        out = torch.empty(K, dtype=torch.float32, device="cuda")
        f.dequantize_row_q(wd.data_ptr(), out.data_ptr(), K)
        assert np.array_equal(bits(out.cpu().numpy()), bits(port.dequantize_row(qt, wq, K)))
        s = torch.zeros(1, dtype=torch.float32, device="cuda")
        f.vec_dot_q(K, s.data_ptr(), wd.data_ptr(), yq.data_ptr())
        want = port.vec_dot(qt, K, wq, port.quantize_row_q8_0(x))
        assert abs(s.item() - want) <= TOL * max(1.0, abs(want))
    assert not L.fl_get_quantize_fn(0).vec_dot_q     # F32 has no entry, like quantize_fns[GGML_TYPE_F32]


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("reference", [False, True])
def test_quantize_row_q4_slots_match_reference(torch, ops, port, nm, qt, reference):
    """quantize_row_q (the SIMD flavour of the reference's x86 build) and quantize_row_q_reference, bit for bit against
    the oracle restatement (itself pinned to the compiled reference, tests/test_oracle_pinning.py) -- incl. exact .5
    ties, where the two flavours differ (rint vs roundf), all-zero and constant blocks, tiny and huge magnitudes."""
    rng = np.random.default_rng(77)
    K = 32 * 64
    x = rng.standard_normal(K).astype(np.float32)
    x[32:64] = 0.0                                           # all-zero block: d = 0, id = 0
    x[64:96] = 3.25                                          # constant block (Q4_1: max == min)
    x[96:128] = np.arange(32, dtype=np.float32) * 0.5 - 7.0  # amax 8.5... exact halves after scaling by 7/amax? no: next
    x[128:160] = np.concatenate([[7.0, -7.0], (np.arange(30) % 14 - 7) + 0.5]).astype(np.float32)   # id = 1: x.5 ties
    x[160:192] *= 1e-30
    x[192:224] *= 1e30
    x[224:256] = np.concatenate([[0.0, 15.0], np.arange(30) % 15 + 0.5]).astype(np.float32)         # Q4_1 ties: d = 1
    got = ops.quantize_row_q(qt, dev(torch, x), reference=reference).cpu().numpy()
    assert np.array_equal(got, port.quantize_row_q4(qt, x, reference))
    if oracle.have_ref():
        assert np.array_equal(got, oracle.Ref().quantize_row_q4(qt, x, reference))
    # what synthetic models are made of (harness/synth.py) is the reference's own file quantizer
    w = rng.standard_normal((5, 256)).astype(np.float32) * 0.02
    got = ops.quantize_row_q(qt, dev(torch, w).view(-1), reference=True).cpu().numpy().reshape(5, -1)
    assert np.array_equal(got, port.quantize_q4(qt, w))


# ------------------------------------------------------------------ a9 mul_mat_q_f32 ---------------
SHAPES = [(48, 256, 1), (48, 256, 2), (48, 256, 3), (48, 256, 5), (48, 256, 8), (48, 256, 9), (48, 256, 16),
          (48, 256, 17), (130, 704, 40), (37, 704, 7), (37, 704, 23), (256, 1024, 128), (300, 2048, 130)]


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K,N", SHAPES)
def test_mul_mat_matches_oracle(torch, ops, port, nm, qt, M, K, N):
    wq = make_weights(port, qt, M, K, 1000 + M + N)
    x = make_x(N, K, 2000 + K + N)
    if N > 2:
        x[1, :64] = 0.0
    want = port.mul_mat_q(qt, wq, x)
    W = ops.QTensor(qt, wq, M, K)
    got = ops.mul_mat(W, dev(torch, x)).cpu().numpy()
    assert got.shape == want.shape
    assert rel_err_rows(got, want) <= TOL, (M, K, N, rel_err_rows(got, want))


@pytest.mark.parametrize("tag", ["s", "m"])
@pytest.mark.parametrize("nm,qt", Q4)
def test_mul_mat_matches_golden(torch, ops, tag, nm, qt):
    g = golden()
    wq, x, ref = g[f"{tag}_{nm}"], g[f"{tag}_x"], g[f"{tag}_{nm}_y"]
    M, K = wq.shape[0], x.shape[1]
    W = ops.QTensor(qt, wq, M, K)
    got = ops.mul_mat(W, dev(torch, x)).cpu().numpy()
    ok = np.isfinite(ref).all(axis=1)
    assert ok.sum() >= x.shape[0] - 1
    assert rel_err_rows(got[ok], ref[ok]) <= TOL
    # the row with a 3e30 outlier: same infinities / NaNs pattern is not required, finiteness pattern is
    assert np.array_equal(np.isfinite(got[~ok]), np.isfinite(ref[~ok]))


@pytest.mark.parametrize("nm,qt", Q4)
def test_three_device_kernels_agree(torch, ops, port, nm, qt):
    """naive (1 thread/output), MFMA and wave-dot GEMV kernels on the same quantized inputs."""
    M, K, N = 200, 1408, 8
    wq = make_weights(port, qt, M, K, 77)
    x = dev(torch, make_x(N, K, 78))
    W = ops.QTensor(qt, wq, M, K)
    a16 = ops.QAct(N, K).quantize(x, layout=16)
    a1 = ops.QAct(N, K).quantize(x, layout=1)
    y_naive = ops.mul_mat_q(W, a16, which=0).cpu().numpy()
    y_mfma = ops.mul_mat_q(W, a16, which=1).cpu().numpy()
    y_gemv = ops.mul_mat_q(W, a1, which=2).cpu().numpy()
    want = port.mul_mat_q(qt, wq, x.cpu().numpy())
    for y in (y_naive, y_mfma, y_gemv):
        assert rel_max_err(y, want) <= TOL


def test_mfma_operand_layout_is_transpose_safe(torch, ops, port):
    """Asymmetric, structured operands: a row<->column or k-permutation slip cannot cancel out."""
    M, K, N = 32, 128, 32
    w = np.zeros((M, K), dtype=np.float32)
    x = np.zeros((N, K), dtype=np.float32)
    for m in range(M):
        w[m] = np.linspace(-1, 1, K) * (m + 1) / M + 0.01 * np.sin(np.arange(K) * (m + 3))
    for n in range(N):
        x[n] = np.cos(np.arange(K) * 0.37 * (n + 1)) * (1 + n)
    for qt in (oracle.Q4_0, oracle.Q4_1):
        wq = port.quantize_q4(qt, w)
        W = ops.QTensor(qt, wq, M, K)
        a = ops.QAct(N, K).quantize(dev(torch, x), layout=16)
        got = ops.mul_mat_q(W, a, which=1).cpu().numpy()
        want = port.mul_mat_q(qt, wq, x)
        assert rel_max_err(got, want) <= TOL
        assert rel_max_err(got.T, want) > 1e-2       # and the check can tell a transpose


# ---- BASELINE.json full sizes: sampled oracle rows + size-independent exact properties -------------
FULL = [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096)]


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K", FULL)
@pytest.mark.parametrize("N", [1, 512])
def test_llama7b_shapes_sampled_rows_vs_oracle(torch, ops, port, nm, qt, M, K, N):
    rng = np.random.default_rng(M + K + N)
    wq = make_weights(port, qt, M, K, 31 + M % 97)
    x = make_x(N, K, 41)
    W = ops.QTensor(qt, wq, M, K)
    got = ops.mul_mat(W, dev(torch, x)).cpu().numpy()
    rows = np.sort(rng.choice(M, size=24, replace=False))
    cols = np.sort(rng.choice(N, size=min(N, 16), replace=False))
    want = port.mul_mat_q(qt, wq[rows], x[cols])
    assert rel_max_err(got[np.ix_(cols, rows)], want) <= TOL


@pytest.mark.parametrize("N", [1, 4, 512])
def test_exact_properties_at_full_size(torch, ops, port, N):
    """Properties that hold BIT-FOR-BIT for the reference's algorithm, checked at 7B size:
       (1) power-of-two homogeneity: mul_mat(W, 2^k x) == 2^k mul_mat(W, x)  (amax, d scale exactly; q unchanged)
       (2) zero activations give exactly zero
       (3) outputs are per-row independent: permuting W's rows permutes y's columns
       (4) outputs are per-column independent: a column computed alone (same kernel family) is identical
    """
    M, K = 4096, 4096
    qt = oracle.Q4_0
    wq = make_weights(port, qt, M, K, 5)
    x = dev(torch, make_x(N, K, 6))
    W = ops.QTensor(qt, wq, M, K)
    y = ops.mul_mat(W, x).clone()
    y8 = ops.mul_mat(W, x * 8.0).clone()
    assert torch.equal(y8, y * 8.0)
    y0 = ops.mul_mat(W, torch.zeros_like(x))
    assert torch.count_nonzero(y0).item() == 0
    perm = np.random.default_rng(0).permutation(M)
    Wp = ops.QTensor(qt, wq[perm], M, K)
    yp = ops.mul_mat(Wp, x)
    assert torch.equal(yp, y[:, torch.from_numpy(perm).cuda()])
    if N >= 16:
        ysub = ops.mul_mat(W, x[100:132].contiguous())
        assert torch.equal(ysub, y[100:132])


# ---- every tile configuration of both MFMA kernels returns the same bits (7B, 13B and 65B shapes) ------------
OLD_CFGS = list(range(14))                    # gemm_q4_mfma.hip FL_GEMM_CONFIGS (round 1, 16x16x32 MFMA)
NEW_CFGS = [100, 101, 102, 103, 104, 105, 106, 108, 116]   # gemm_q4_mfma32.hip FL_GEMM32_CONFIGS (32x32x32 MFMA); 116 = two tile shapes in one launch
LLAMA_SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096),        # 7B
                (5120, 5120), (13824, 5120), (5120, 13824),                       # 13B
                (8192, 8192), (22016, 8192), (8192, 22016)]                       # 65B


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K", LLAMA_SHAPES)
def test_tile_configurations_are_bit_identical_and_match_oracle(torch, ops, port, nm, qt, M, K):
    """N = 512 (BASELINE.json's n_batch) at every LLaMA matrix shape: each output accumulates its per-block terms in K
    order whatever the tile shape and whichever MFMA computes the block dots, so all 23 configurations -- and the one
    pick_config chooses -- must agree BIT FOR BIT; sampled rows are checked against the oracle."""
    from fastllama_amd import hip
    from harness import synth
    L = hip.load()
    N = 512
    blocks = synth.synth_q4(M, K, qt, 11 + M % 13)
    W = ops.QTensor(qt, blocks, M, K)
    x = dev(torch, make_x(N, K, 17))
    a = ops.QAct(N, K).quantize(x)
    try:
        L.fl_debug_set(0, 12)
        ref = ops.mul_mat_q(W, a).clone()
        for cfg in [-1, -2] + OLD_CFGS + NEW_CFGS:
            L.fl_debug_set(0, cfg)
            y = ops.mul_mat_q(W, a)
            assert torch.equal(y, ref), (M, K, cfg, int((y != ref).sum()))
    finally:
        L.fl_debug_set(0, -1)
    rng = np.random.default_rng(M + K)
    rows = np.sort(rng.choice(M, size=16, replace=False))
    cols = np.sort(rng.choice(N, size=12, replace=False))
    want = port.mul_mat_q(qt, blocks[torch.from_numpy(rows).cuda()].cpu().numpy(), x.cpu().numpy()[cols])
    assert rel_max_err(ref.cpu().numpy()[np.ix_(cols, rows)], want) <= TOL
    W.free()


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K,N", [(48, 192, 17), (200, 1408, 9), (130, 256, 70), (264, 320, 33), (256, 352, 33), (40, 96, 20),
                                   (33, 32, 16), (1000, 4096, 100)])
def test_tile_configurations_ragged_shapes(torch, ops, port, nm, qt, M, K, N):
    """Row / column / K tails (M % 32, N % 32, K/32 odd or not a multiple of the 8-block loop body) in every configuration."""
    from fastllama_amd import hip
    L = hip.load()
    wq = make_weights(port, qt, M, K, 5 + M)
    W = ops.QTensor(qt, wq, M, K)
    x = make_x(N, K, 6 + N)
    a = ops.QAct(N, K).quantize(dev(torch, x))
    want = port.mul_mat_q(qt, wq, x, strict=False)
    try:
        L.fl_debug_set(0, 12)
        ref = ops.mul_mat_q(W, a).clone()
        assert rel_max_err(ref.cpu().numpy(), want) <= TOL
        for cfg in OLD_CFGS + NEW_CFGS:
            L.fl_debug_set(0, cfg)
            y = torch.full((N, (M + 3) // 4 * 4), 7.0, device="cuda")[:, :M]
            ops.mul_mat_q(W, a, out=y)
            assert torch.equal(y, ref), (M, K, N, cfg)
    finally:
        L.fl_debug_set(0, -1)


# ---- the fused forms of the prefill GEMM against GEMM + the separate op kernel --------------------------------
@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("E,D,N,n_past", [(256, 32, 40, 9), (4096, 128, 512, 0), (4096, 128, 96, 160), (5120, 128, 512, 512)])
def test_gemm_qkv_rope_epilogue_equals_gemm_plus_rope_kv(torch, ops, port, nm, qt, E, D, N, n_past):
    """wq|wk|wv matmul with rope on Q / K and the K / V cache stores as its epilogue == the plain matmul followed by
    rope_kv_kernel, bit for bit, at n_past > 0 (position 0 has the identity rotation) in both kernels' configurations."""
    from fastllama_amd import hip
    from harness import synth
    L = hip.load()
    n_ctx = 1024
    W = ops.QTensor(qt, synth.synth_q4(3 * E, E, qt, 3), 3 * E, E)
    x = dev(torch, make_x(N, E, 4))
    a = ops.QAct(N, E).quantize(x)
    rt = np.empty((n_ctx, D // 2, 2), np.float32)
    L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), n_ctx, D)
    rd = dev(torch, rt)
    qkv = ops.mul_mat_q(W, a).contiguous()
    kc0, vc0 = torch.zeros((n_ctx, E), device="cuda"), torch.zeros((E, n_ctx), device="cuda")
    hip.check(L.fl_debug_rope_kv(qkv.data_ptr(), 3 * E, N, E, D, n_past, n_ctx, rd.data_ptr(), kc0.data_ptr(), vc0.data_ptr(), None))
    try:
        for cfg in (-1, -2, 10, 12, 100, 101, 106, 116):
            L.fl_debug_set(0, cfg)
            y = torch.zeros((N, 3 * E), device="cuda")
            kc, vc = torch.zeros_like(kc0), torch.zeros_like(vc0)
            hip.check(L.fl_debug_gemm_qkv(W.handle, a.handle, y.data_ptr(), 3 * E, rd.data_ptr(), kc.data_ptr(), vc.data_ptr(), E, D,
                                          n_past, n_ctx, None))
            assert torch.equal(y[:, :E], qkv[:, :E]), cfg                 # roped Q
            assert torch.equal(kc, kc0) and torch.equal(vc, vc0), cfg     # roped K rows, transposed V columns
            assert not y[:, E:].any()                                     # K / V never land in y
    finally:
        L.fl_debug_set(0, -1)


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("F,K,N", [(704, 256, 40), (11008, 4096, 512), (13824, 5120, 100), (352, 256, 33)])
def test_gemm_silu_epilogue_equals_gemm_plus_silu_mul_quant(torch, ops, port, nm, qt, F, K, N):
    """woven w1|w3 matmul whose epilogue writes Q8_0(silu(w1 x) * (w3 x)) == plain matmul + silu_mul_quant_kernel:
    the exported Q8_0 bytes (quants, d, s) are identical."""
    from fastllama_amd import hip
    from harness import synth
    L = hip.load()
    W = ops.QTensor(qt, synth.synth_q4(2 * F, K, qt, 8), 2 * F, K)     # rows woven by 16: group 2p = w1, 2p+1 = w3
    x = dev(torch, make_x(N, K, 9))
    a = ops.QAct(N, K).quantize(x)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
    sd = dev(torch, s.view(np.int16))
    h13 = ops.mul_mat_q(W, a).contiguous()
    want = ops.QAct(N, F)
    hip.check(L.fl_quantize_q8_layout(want.handle, h13.data_ptr(), 2 * F, N, F, 16, None))    # bookkeeping (N, layout)
    want.N, want.K = N, F
    # woven = True: h13 = [w1 x 16 | w3 x 16 | ...]
    hip.check(L.fl_debug_silu_mul_quant_woven(h13.data_ptr(), 2 * F, N, F, sd.data_ptr(), want.handle, 16, None))
    wb = want.export().cpu().numpy()
    try:
        for cfg in (-1, -2, 10, 12, 100, 101, 106, 116):
            L.fl_debug_set(0, cfg)
            out = ops.QAct(N, F)
            hip.check(L.fl_debug_gemm_silu(W.handle, a.handle, sd.data_ptr(), out.handle, None))
            out.N, out.K = N, F
            assert np.array_equal(out.export().cpu().numpy(), wb), cfg
    finally:
        L.fl_debug_set(0, -1)


@pytest.mark.parametrize("nm,qt", Q4)
def test_gemm_residual_epilogue(torch, ops, port, nm, qt):
    """y = mul_mat + resid (the ggml_add after wo / w2) fused into the store == matmul then add."""
    from fastllama_amd import hip
    from harness import synth
    L = hip.load()
    M, K, N = 4096, 4096, 200
    W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 2), M, K)
    a = ops.QAct(N, K).quantize(dev(torch, make_x(N, K, 3)))
    r = dev(torch, make_x(N, M, 4))
    base = ops.mul_mat_q(W, a)
    try:
        for cfg in (-1, -2, 12, 100, 101, 106, 116):
            L.fl_debug_set(0, cfg)
            y = torch.empty((N, M), device="cuda")
            hip.check(L.fl_debug_mul_mat_q_resid(W.handle, a.handle, y.data_ptr(), M, r.data_ptr(), M, None))
            assert torch.equal(y, base + r), cfg
    finally:
        L.fl_debug_set(0, -1)


def test_mul_mat_argument_errors(torch, ops, port):
    from fastllama_amd import hip
    wq = make_weights(port, oracle.Q4_0, 16, 64, 1)
    W = ops.QTensor(oracle.Q4_0, wq, 16, 64)
    a = ops.QAct(4, 128).quantize(dev(torch, make_x(4, 128, 1)))
    with pytest.raises(hip.FastLlamaHipError):       # K mismatch
        ops.mul_mat_q(W, a)
    L = hip.load()
    assert L.fl_mul_mat_q_f32(W.handle, None, 64, None, 16, 1, None) == hip.FL_EINVAL
    assert L.fl_quantize_row_q8_0(dev(torch, make_x(1, 64, 1)[0]).data_ptr(), 1, 48, None) == hip.FL_EINVAL
