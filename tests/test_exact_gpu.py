"""GPU: the reference-order ("exact") kernels, fastllama_amd/csrc/exact_kernels.hip.

The bar here is not a tolerance: every result must equal the reference's x86 build BIT FOR BIT --
  * mul_mat_q (N <= 8 wave kernel, N >= 9 tile kernel) against the C restatement of ggml_vec_dot_q4_{0,1}_q8_0's AVX2 branch
    (oracle/q4_oracle.c, pinned bit-exact against the compiled reference in tests/test_oracle_pinning.py),
  * the f32 attention matmuls against the restatement of ggml_vec_dot_f32 as compiled into mul_mat_f32
    (orc_vec_dot_f32_mm, pinned by tests/test_llama_eval_oracle.py: the numpy eval built on it equals the reference's logits),
  * whole models (prefill, chunked ingest, decode steps) against the LIVE reference through its own C-ABI.
The full 7B x 32 layers x n_batch 512 run is tests/test_parity_7b_gpu.py::test_llama7b_full_model_exact_mode_is_bit_identical.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from harness import ggjt, llama_capi
from util import bits, make_weights, make_x

pytestmark = pytest.mark.gpu
Q4 = [("q40", oracle.Q4_0), ("q41", oracle.Q4_1)]
TEXT = "The quick brown fox jumps over the lazy dog; 0123456789 times!? And then some more text follows here, enough of it."


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


@pytest.fixture(scope="module")
def reflib():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    return llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ mul_mat_q ----------------------------------------
@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K,N", [(48, 192, 1), (200, 1408, 1), (33, 64, 2), (130, 256, 3), (264, 320, 5), (1000, 4096, 8),
                                   (48, 192, 9), (200, 1408, 17), (130, 256, 70), (264, 320, 33), (40, 96, 20), (33, 64, 16),
                                   (1000, 4096, 100), (4096, 4096, 1), (704, 11008, 1), (512, 11008, 48)])
def test_mul_mat_q_exact_is_bit_identical_to_the_reference_order(torch, ops, port, nm, qt, M, K, N):
    """Row / column / K tails included (M % 16, N % 16, odd block counts); N <= 8 takes the wave kernel, N >= 9 the tile kernel."""
    wq = make_weights(port, qt, M, K, 5 + M)
    W = ops.QTensor(qt, wq, M, K)
    x = make_x(N, K, 6 + N)
    x[0, :32] = 0.0                                             # an all-zero block: d_x = 0
    a = ops.QAct(N, K).quantize(dev(torch, x))
    want = port.mul_mat_q(qt, wq, x, strict=False)
    y = torch.full((N, (M + 3) // 4 * 4), 7.0, device="cuda")[:, :M]
    ops.mul_mat_q(W, a, which=3, out=y)
    got = y.cpu().numpy()
    assert np.array_equal(bits(got), bits(want)), (M, K, N, int((bits(got) != bits(want)).sum()), float(np.abs(got - want).max()))
    W.free()


@pytest.mark.parametrize("nm,qt", Q4)
def test_mul_mat_q_exact_large_scales_and_signs(torch, ops, port, nm, qt):
    """Weights and activations spanning many orders of magnitude: the 16x / (1/16) bookkeeping of the Q4_0 layout (q4_layout.h)
    must stay exact, and the residual add must be the plain f32 add that follows the matmul (lib/llama.cpp:407)."""
    from fastllama_amd import hip
    rng = np.random.default_rng(9)
    M, K, N = 96, 512, 24
    w = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-12, 6, (M, 1)))).astype(np.float32)
    wq = port.quantize_q4(qt, w)
    x = (rng.standard_normal((N, K)) * np.exp(rng.uniform(-10, 10, (N, 1)))).astype(np.float32)
    W = ops.QTensor(qt, wq, M, K)
    a = ops.QAct(N, K).quantize(dev(torch, x))
    want = port.mul_mat_q(qt, wq, x)
    got = ops.mul_mat_q(W, a, which=3).cpu().numpy()
    assert np.array_equal(bits(got), bits(want))
    a1 = ops.QAct(8, K).quantize(dev(torch, x[:5]))
    got1 = ops.mul_mat_q(W, a1, which=3).cpu().numpy()
    assert np.array_equal(bits(got1), bits(want[:5]))


# ------------------------------------------------------------------ attention matmuls --------------------------------
def _abt_exact(torch, A, B, alpha, causal, n_past):
    """C[z][m][n] = alpha * dot(A[z][m], B[z][n]) through fl_debug_gemm_f32_abt_exact."""
    from fastllama_amd import hip
    L = hip.load()
    Z, M, K = A.shape
    Nn = B.shape[1]
    Ad, Bd = dev(torch, A), dev(torch, B)
    Cd = torch.full((Z, M, Nn), 7.0, device="cuda")
    hip.check(L.fl_debug_gemm_f32_abt_exact(C.c_void_p(Ad.data_ptr()), K, M * K, C.c_void_p(Bd.data_ptr()), K, Nn * K,
                                            C.c_void_p(Cd.data_ptr()), Nn, M * Nn, M, Nn, K, Z, alpha, causal, n_past, None))
    torch.cuda.synchronize()
    return Cd.cpu().numpy()


@pytest.mark.parametrize("K", [128, 32, 64, 96, 100, 31, 7, 3, 45, 77, 255])
def test_attention_dot_has_the_reference_lane_order(torch, ops, port, K):
    """Every leftover form of the compiled ggml_vec_dot_f32: n % 32 in {0, 1..3 (scalar FMAs), 4..7 (one chunk of 4 + FMAs),
    >= 8 (chunks of 8, of 4, FMAs)}."""
    rng = np.random.default_rng(K)
    A = rng.standard_normal((3, 20, K)).astype(np.float32)
    B = rng.standard_normal((3, 37, K)).astype(np.float32)
    alpha = np.float32(1.0) / np.sqrt(np.float32(128.0), dtype=np.float32)
    got = _abt_exact(torch, A, B, float(alpha), 0, 0)
    for z in range(3):
        want = (port.mul_mat_f32(B[z], A[z]) * alpha).astype(np.float32)      # [M, Nn]
        assert np.array_equal(bits(got[z]), bits(want)), (K, z)


@pytest.mark.parametrize("n_past,N", [(0, 40), (0, 64), (13, 33), (100, 1), (95, 1), (64, 50)])
def test_attention_pv_skips_only_zero_steps(torch, ops, port, n_past, N):
    """KQV (causal_mode 2): probabilities are zero beyond n_past + m, 32-element steps made of zeros are skipped -- same bits as
    the full-length dot the reference computes."""
    rng = np.random.default_rng(n_past + N)
    P, D, Z = n_past + N, 32, 2
    pr = np.abs(rng.standard_normal((Z, N, P))).astype(np.float32)
    for m in range(N):
        pr[:, m, n_past + m + 1:] = 0.0
    vt = rng.standard_normal((Z, D, P)).astype(np.float32)
    got = _abt_exact(torch, pr, vt, 1.0, 2, n_past)
    for z in range(Z):
        want = port.mul_mat_f32(vt[z], pr[z])
        assert np.array_equal(bits(got[z]), bits(want)), (n_past, N, z)


# ------------------------------------------------------------------ whole models -------------------------------------
def _ref_logits(reflib, path, text, n_ctx, n_batch):
    ref = llama_capi.Session(reflib, path, n_ctx=n_ctx, n_batch=n_batch, all_logits=True)
    ref.perplexity(text)
    return ref.logits().copy()


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("cfgname,ntext", [("TINY", 40), ("SMALL", 63), ("SMALL", 110)])
def test_model_exact_mode_equals_reference_bit_for_bit(tmp_path, torch, port, reflib, nm, qt, cfgname, ntext):
    """Prefill of one batch (N >= 9 kernels) on small models whose head_dim (32 / 64) and key counts exercise every leftover
    form: all positions' logits equal the LIVE reference's, bit for bit."""
    from harness.flmodel import FlModel
    cfg = getattr(ggjt, cfgname)
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=4321)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qt, tensors)
    text = TEXT[:ntext]
    toks = ggjt.text_tokens(text)
    want = _ref_logits(reflib, path, text, 128, 128).reshape(len(toks), cfg["n_vocab"])
    m = FlModel(cfg, qt, tensors, n_ctx=128, max_batch=128)
    m.set_exact(True)
    got = m.eval(toks, all_logits=True)
    assert np.array_equal(bits(got), bits(want)), float(np.abs(got - want).max())
    m.set_exact(False)
    fast = m.eval(toks, all_logits=True)
    assert np.array_equal(bits(fast[0]), bits(want[0])) or np.abs(fast[0] - want[0]).max() <= 1e-5 * np.abs(want).max()
    m.free()


def test_model_exact_mode_chunked_ingest_and_decode_steps(tmp_path, torch, port, reflib):
    """The session pattern: n_batch-8 chunks (N <= 8 kernels on the prompt), then greedy decode steps (N = 1, hipGraph replay)."""
    from harness.flmodel import FlModel
    cfg, qt = ggjt.TINY, oracle.Q4_0
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=99)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qt, tensors)
    prompt = "abcdefghijklmnopqrstuvw"
    toks = ggjt.text_tokens(" " + prompt)
    ref = llama_capi.Session(reflib, path, n_ctx=64, n_batch=8)
    assert ref.ingest(prompt)
    m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=8)
    m.set_exact(True)
    n_past, lg = 0, None
    for i in range(0, len(toks), 8):
        lg = m.eval(toks[i:i + 8], n_past=n_past)
        n_past += len(toks[i:i + 8])
    for _ in range(6):
        ok, _ = ref.generate(1, temp=0.0)
        want = ref.logits()
        assert np.array_equal(bits(lg[-1]), bits(want))
        tok = int(np.argmax(want))
        lg = m.eval([tok], n_past=n_past)
        n_past += 1
    m.free()
