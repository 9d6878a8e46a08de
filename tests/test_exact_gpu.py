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
@pytest.mark.parametrize("M,K", [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096), (5120, 5120), (13824, 5120), (5120, 13824),
                                 (8192, 8192), (22016, 8192), (8192, 22016)])
def test_exact_tile_kernel_at_llama_shapes(torch, ops, port, nm, qt, M, K):
    """N = 512 at every LLaMA matrix shape (7B / 13B / 65B): the K = 4 MFMA form equals the v_dot4 form bit for bit over the whole
    output, and sampled rows / columns equal the oracle."""
    from harness import synth
    N = 512
    blocks = synth.synth_q4(M, K, qt, 11 + M % 13)
    W = ops.QTensor(qt, blocks, M, K)
    x = dev(torch, make_x(N, K, 17))
    a = ops.QAct(N, K).quantize(x)
    y_mfma = ops.mul_mat_q(W, a, which=3).clone()
    y_valu = ops.mul_mat_q(W, a, which=4)
    assert torch.equal(y_mfma, y_valu), int((y_mfma != y_valu).sum())
    rng = np.random.default_rng(M + K)
    rows = np.sort(rng.choice(M, size=16, replace=False))
    cols = np.sort(rng.choice(N, size=12, replace=False))
    want = port.mul_mat_q(qt, blocks[torch.from_numpy(rows).cuda()].cpu().numpy(), x.cpu().numpy()[cols])
    assert np.array_equal(bits(y_mfma.cpu().numpy()[np.ix_(cols, rows)]), bits(want))
    W.free()


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K,N", [(8200, 320, 300), (33000, 96, 75)])
def test_exact_tile_kernel_persistent_workgroups_on_ragged_shapes(torch, ops, port, nm, qt, M, K, N):
    """More 64 x 64 tiles than residency slots (645 / 1032 against 512: workgroups walk several tiles, the next tile's first K-step
    requested during the last one of the current tile) with ragged last row and column tiles and an odd number of K-steps: the whole
    output equals the oracle's, bit for bit."""
    wq = make_weights(port, qt, M, K, 3 + M)
    W = ops.QTensor(qt, wq, M, K)
    x = make_x(N, K, 9 + N)
    a = ops.QAct(N, K).quantize(dev(torch, x))
    want = port.mul_mat_q(qt, wq, x, strict=False)
    y = torch.full((N, (M + 3) // 4 * 4), 7.0, device="cuda")[:, :M]
    ops.mul_mat_q(W, a, which=3, out=y)
    got = y.cpu().numpy()
    assert np.array_equal(bits(got), bits(want)), (int((bits(got) != bits(want)).sum()), float(np.abs(got - want).max()))
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


def test_model_exact_mode_deep_context(tmp_path, torch, port, reflib):
    """A prompt of three n_batch evals (512 + 512 + 40 tokens): the later chunks' attention runs over 1024+ keys -- several
    512-key pieces of the V.P kernel, leftover forms included -- and the last logits still equal the LIVE reference's bits."""
    from harness.flmodel import FlModel
    cfg, qt = ggjt.SMALL, oracle.Q4_0
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=7)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qt, tensors)
    rng = np.random.default_rng(11)
    prompt = bytes(rng.integers(33, 127, size=1062).astype(np.uint8)).decode()
    toks = ggjt.text_tokens(" " + prompt)
    assert len(toks) == 1064
    ref = llama_capi.Session(reflib, path, n_ctx=1100, n_batch=512, n_threads=8)
    assert ref.ingest(prompt)
    ok, _ = ref.generate(1, temp=0.0)
    want = ref.logits().copy()
    ref.close()
    m = FlModel(cfg, qt, tensors, n_ctx=1100, max_batch=512)
    m.set_exact(True)
    n_past, lg = 0, None
    for i in range(0, len(toks), 512):
        lg = m.eval(toks[i:i + 512], n_past=n_past)
        n_past += len(toks[i:i + 512])
    assert np.array_equal(bits(lg[-1]), bits(want))
    m.free()


@pytest.mark.parametrize("exact", [True, False])
def test_split_prefill_eval_equals_the_single_eval(torch, port, exact):
    """fl_model_set_graph bit 8: a prefill of >= 256 tokens runs as two halves on two streams, the first ending on a 32-key boundary
    (eval_split, model.cpp; off by default -- measured slower, profiles/r04_split_eval.txt): every logit, the embedding of the last token and the K/V cache (through a decode step on top) equal the unsplit
    eval's -- in the reference-order mode because nothing of a token's arithmetic depends on its batch once ggml_vec_dot_f32's
    32-element steps end where the half ends, in the fast mode because its kernels are batch-split invariant too.  Positions off
    the 32-key grid, ragged sizes and all-logits included."""
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg, qt = ggjt.SMALL, oracle.Q4_1
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=17)
    rng = np.random.default_rng(4)
    toks = rng.integers(3, 259, 1200).astype(np.int32)
    outs = []
    for mode in (0, 256):                                    # bit 8 of fl_model_set_graph: default (one eval) | split
        m = FlModel(cfg, qt, tensors, n_ctx=1300, max_batch=512)
        m.set_exact(exact)
        hip.check(L.fl_model_set_graph(m.h, 1 | mode))
        a, ea = m.eval(toks[:512], n_past=0, all_logits=True, embeddings=True)
        b = m.eval(toks[512:512 + 300], n_past=512)                         # last logits only
        c = m.eval(toks[812:812 + 37], n_past=812)                          # below the split threshold
        d, ed = m.eval(toks[849:849 + 351], n_past=849, all_logits=True, embeddings=True)   # n_past off the 32-key grid
        e = m.eval(toks[5:6], n_past=1200)                                  # a decode step over the whole cache
        outs.append((a, ea, b, c, d, ed, e))
        m.free()
    for x, y in zip(*outs):
        assert np.array_equal(bits(x), bits(y))


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


@pytest.mark.parametrize("nm,qt", Q4)
def test_exact_decode_forms_agree_with_each_other_and_the_reference(tmp_path, torch, port, reflib, nm, qt):
    """Decode behind a 300-token context (key counts with every leftover form, past the 256-position switch to the two-launch
    attention): the fused single-token kernels (producer/chain-wave matmuls with their prologues, one-launch attention), the
    same with the split attention, and the generic per-op sequence return the same bits in exact mode -- the reference's."""
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=31)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qt, tensors)
    rng = np.random.default_rng(1)
    prompt = bytes(rng.integers(33, 127, size=297).astype(np.uint8)).decode()
    toks = ggjt.text_tokens(" " + prompt)
    ref = llama_capi.Session(reflib, path, n_ctx=512, n_batch=512)
    assert ref.ingest(prompt)
    want = []
    for _ in range(5):
        ok, _ = ref.generate(1, temp=0.0)
        want.append(ref.logits().copy())
    ref.close()
    m = FlModel(cfg, qt, tensors, n_ctx=512, max_batch=512)
    m.set_exact(True)
    for mode, what in [(1, "fused, automatic attention form"), (1 | 8, "fused, one-launch attention"), (1 | 16, "fused, split attention"),
                       (1 | 2, "generic per-op kernels"), (0, "no hipGraph")]:
        hip.check(L.fl_model_set_graph(m.h, mode))
        lg = m.eval(toks, n_past=0)
        n_past = len(toks)
        for i in range(5):
            assert np.array_equal(bits(lg[-1]), bits(want[i])), (what, i, float(np.abs(lg[-1] - want[i]).max()))
            lg = m.eval([int(np.argmax(want[i]))], n_past=n_past)
            n_past += 1
    m.free()


def test_prepare_builds_the_operand_copies_and_a_failed_copy_warns(torch, port, monkeypatch):
    """fl_model_prepare builds the derived weight copies of the reference-order kernels up front (fl_model_prepared says which are
    resident); when a copy cannot be allocated the kernel family that reads the primary layout runs -- the SAME bits -- and the warning
    handler is told once per kind, with the bytes that were missing (no silent cliffs)."""
    import ctypes as C
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg, qt = ggjt.SMALL, oracle.Q4_0
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=3)
    toks = np.random.default_rng(2).integers(3, 259, 40).astype(np.int32)
    lines = []
    CB = C.CFUNCTYPE(None, C.c_char_p)
    cb = CB(lambda s: lines.append(s.decode()))
    L.fl_set_warn_handler(C.cast(cb, C.c_void_p))
    try:
        m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=40)
        assert L.fl_model_prepared(m.h) == 0
        assert m.prepare(3) == 3 and not lines
        want = m.eval(toks, all_logits=True)
        want1 = m.eval([int(toks[7])], n_past=40)
        m.free()
        monkeypatch.setenv("FL_TEST_FAIL_DERIVED", "3")
        m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=40)
        assert m.prepare(3) == 0
        assert len(lines) == 2 and "f16 operand copies" in lines[0] and "decode copies" in lines[1], lines
        got = m.eval(toks, all_logits=True)
        got1 = m.eval([int(toks[7])], n_past=40)
        assert len(lines) == 2                                        # told once, not per eval
        m.free()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got1.view(np.uint32), want1.view(np.uint32))
    finally:
        L.fl_set_warn_handler(None)


# ------------------------------------------------------------------ the single-token (decode) kernels, op by op ------------
from oracle import llama_eval as le  # noqa: E402


@pytest.fixture()
def exact_hooks():
    """fl_debug_set(2, 1): the single-token test hooks run the exact-mode kernels"""
    from fastllama_amd import hip
    L = hip.load()
    L.fl_debug_set(2, 1)
    yield L
    L.fl_debug_set(2, 0)


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K", [(48, 64), (300, 256), (4096, 4096), (12288, 4096), (1024, 8192), (4608, 8192), (32, 2560)])
def test_exact_gemv_with_norm_prologue(torch, ops, port, exact_hooks, nm, qt, M, K):
    """rms_norm * w -> Q8_0 -> mul_mat in one launch (producer / chain waves): the oracle's bits."""
    from fastllama_amd import hip
    L = exact_hooks
    rng = np.random.default_rng(M + K + qt)
    wq = port.quantize_q4(qt, (rng.standard_normal((M, K)) * 0.05).astype(np.float32))
    W = ops.QTensor(qt, wq, M, K)
    x = (rng.standard_normal((1, K)) * 1.7).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    xd, nd = dev(torch, x), dev(torch, nw)
    y = torch.full((M,), 3.0, device="cuda")
    yn = torch.zeros((1, K), device="cuda")
    hip.check(L.fl_debug_gemv_norm(W.handle, xd.data_ptr(), nd.data_ptr(), yn.data_ptr(), y.data_ptr(), None))
    cur = le.rms_norm_mul(x, nw)
    assert np.array_equal(bits(yn.cpu().numpy()), bits(cur))
    want = port.mul_mat_q(qt, wq, cur, strict=False)[0]
    assert np.array_equal(bits(y.cpu().numpy()), bits(want)), float(np.abs(y.cpu().numpy() - want).max())


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("F,K,E", [(64, 64, 48), (704, 256, 256), (11008, 4096, 4096), (13824, 5120, 512), (4608, 8192, 8192)])
def test_exact_feed_forward_pair_and_quant_prologue(torch, ops, port, exact_hooks, nm, qt, F, K, E):
    """(a) woven w1|w3: norm prologue + the two dots + silu*mul epilogue in one launch; (b) w2 with the Q8_0 prologue
    (+ residual); (c) w2 with the silu*mul prologue on an f32 [w1 x | w3 x] vector -- each the oracle's bits."""
    from fastllama_amd import hip
    L = exact_hooks
    rng = np.random.default_rng(F + K + qt)
    w1 = port.quantize_q4(qt, (rng.standard_normal((F, K)) * 0.05).astype(np.float32))
    w3 = port.quantize_q4(qt, (rng.standard_normal((F, K)) * 0.05).astype(np.float32))
    rb = w1.shape[1]
    woven = np.empty((2 * F, rb), np.uint8)
    wv = woven.reshape(F // 16, 2, 16, rb)
    wv[:, 0] = w1.reshape(F // 16, 16, rb)
    wv[:, 1] = w3.reshape(F // 16, 16, rb)
    W = ops.QTensor(qt, woven, 2 * F, K)
    x = (rng.standard_normal((1, K)) * 1.3).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
    xd, nd, sd = dev(torch, x), dev(torch, nw), dev(torch, s.view(np.int16))
    cur = le.rms_norm_mul(x, nw)
    h1, h3 = port.mul_mat_q(qt, w1, cur, strict=False), port.mul_mat_q(qt, w3, cur, strict=False)
    want_act = (le.silu(h1) * h3).astype(np.float32)
    # the forms of (a) (q4_kernels.h): automatic; 2 = the two groups of a feature as two workgroups meeting in a workspace (launched three
    # times: its slots must be back at zero after every launch); 1 = one workgroup per feature pair, groups in turn
    for form, reps in [(0, 2), (2, 3), (1, 1)]:
        L.fl_debug_set(5, form)
        for _ in range(reps):
            act = torch.full((F,), 9.0, device="cuda")
            hip.check(L.fl_debug_gemv_norm_silu(W.handle, xd.data_ptr(), nd.data_ptr(), sd.data_ptr(), act.data_ptr(), None))
            assert np.array_equal(bits(act.cpu().numpy()), bits(want_act[0])), form
    L.fl_debug_set(5, 0)
    w2 = port.quantize_q4(qt, (rng.standard_normal((E, F)) * 0.05).astype(np.float32))
    W2 = ops.QTensor(qt, w2, E, F)
    res = rng.standard_normal(E).astype(np.float32)
    rd = dev(torch, res)
    y = torch.empty(E, device="cuda")
    hip.check(L.fl_debug_gemv_quant(W2.handle, act.data_ptr(), y.data_ptr(), rd.data_ptr(), None))
    want = (port.mul_mat_q(qt, w2, want_act, strict=False)[0] + res).astype(np.float32)
    assert np.array_equal(bits(y.cpu().numpy()), bits(want))
    h13 = np.concatenate([h1, h3], axis=1)
    hip.check(L.fl_debug_gemv_silu(W2.handle, dev(torch, h13).data_ptr(), sd.data_ptr(), y.data_ptr(), None, None))
    assert np.array_equal(bits(y.cpu().numpy()), bits(port.mul_mat_q(qt, w2, want_act, strict=False)[0]))


@pytest.fixture()
def stream_form(exact_hooks):
    """fl_debug_set(6, 1): every reference-order N = 1 matmul takes the one-wave-per-row-group form (gemv1_q4_exact_stream.hip),
    whatever its row count (automatic: from three row groups per CU on)"""
    exact_hooks.fl_debug_set(6, 1)
    yield exact_hooks
    exact_hooks.fl_debug_set(6, -1)
    exact_hooks.fl_debug_set(7, 0)


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("M,K", [(48, 64), (300, 256), (130, 2560), (12288, 4096), (1024, 8192), (2000, 5120), (72, 6656), (100, 416)])
def test_stream_form_with_every_prologue(torch, ops, port, stream_form, nm, qt, M, K):
    """One wave per row group over the whole of K, four row groups sharing a prologue: rms_norm prologue (+ the normalised vector), plain Q8_0
    prologue + residual, ready-made Q8_0 operand + residual -- the oracle's bits at ragged row counts, partial quads and padded rounds."""
    from fastllama_amd import hip
    L = stream_form
    rng = np.random.default_rng(M * 7 + K + qt)
    wq = port.quantize_q4(qt, (rng.standard_normal((M, K)) * 0.05).astype(np.float32))
    W = ops.QTensor(qt, wq, M, K)
    x = (rng.standard_normal((1, K)) * 1.7).astype(np.float32)
    res = rng.standard_normal(M).astype(np.float32)
    xd, rd = dev(torch, x), dev(torch, res)
    if K <= 8192:
        nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
        cur = le.rms_norm_mul(x, nw)
        want_n = port.mul_mat_q(qt, wq, cur, strict=False)[0]
        # row groups per workgroup: automatic (one workgroup per CU), 1, 3 (a 4-wave workgroup with an idle wave), 7 (8 waves), 12
        for per_wg in (0, 1, 3, 7, 12):
            L.fl_debug_set(7, per_wg)
            y = torch.full((M,), 3.0, device="cuda")
            yn = torch.zeros((1, K), device="cuda")
            hip.check(L.fl_debug_gemv_norm(W.handle, xd.data_ptr(), dev(torch, nw).data_ptr(), yn.data_ptr(), y.data_ptr(), None))
            assert np.array_equal(bits(yn.cpu().numpy()), bits(cur)), per_wg
            assert np.array_equal(bits(y.cpu().numpy()), bits(want_n)), per_wg
        L.fl_debug_set(7, 0)
    want = (port.mul_mat_q(qt, wq, x, strict=False)[0] + res).astype(np.float32)
    y = torch.full((M,), 3.0, device="cuda")
    hip.check(L.fl_debug_gemv_quant(W.handle, xd.data_ptr(), y.data_ptr(), rd.data_ptr(), None))
    assert np.array_equal(bits(y.cpu().numpy()), bits(want))
    a = ops.QAct(1, K).quantize(xd, layout=1)
    y = torch.full((M,), 3.0, device="cuda")
    hip.check(L.fl_debug_gemv_q8(W.handle, a.handle, y.data_ptr(), rd.data_ptr(), None))
    assert np.array_equal(bits(y.cpu().numpy()), bits(want))


@pytest.mark.parametrize("nm,qt", Q4)
@pytest.mark.parametrize("F,K,E", [(64, 64, 48), (704, 256, 256), (11008, 4096, 4096), (13824, 5120, 512), (4608, 8192, 2048), (96, 416, 32)])
def test_stream_form_writes_the_q8_operand_of_w2(torch, ops, port, stream_form, nm, qt, F, K, E):
    """Woven w1|w3 as workgroups of 32 whole features: (a) the f32 features silu(w1 x) * (w3 x); (b) their Q8_0 blocks written by the
    matmul's workgroups -- the reference's quantize_row_q8_0 of the f32 features, byte for byte -- and w2 on that operand (+ residual)."""
    from fastllama_amd import hip
    L = stream_form
    rng = np.random.default_rng(F + K + qt + 5)
    w1 = port.quantize_q4(qt, (rng.standard_normal((F, K)) * 0.05).astype(np.float32))
    w3 = port.quantize_q4(qt, (rng.standard_normal((F, K)) * 0.05).astype(np.float32))
    rb = w1.shape[1]
    woven = np.empty((2 * F, rb), np.uint8)
    wv = woven.reshape(F // 16, 2, 16, rb)
    wv[:, 0] = w1.reshape(F // 16, 16, rb)
    wv[:, 1] = w3.reshape(F // 16, 16, rb)
    W = ops.QTensor(qt, woven, 2 * F, K)
    x = (rng.standard_normal((1, K)) * 1.3).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    s = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
    xd, nd, sd = dev(torch, x), dev(torch, nw), dev(torch, s.view(np.int16))
    cur = le.rms_norm_mul(x, nw)
    h1, h3 = port.mul_mat_q(qt, w1, cur, strict=False), port.mul_mat_q(qt, w3, cur, strict=False)
    want_act = (le.silu(h1) * h3).astype(np.float32)
    L.fl_debug_set(5, 0)
    a = ops.QAct(1, F)
    for per_wg in (0, 4, 8, 12):                                # blocks of 32 features per workgroup: automatic (one workgroup per CU), 1, 2, 3
        L.fl_debug_set(7, per_wg)
        act = torch.full((F,), 9.0, device="cuda")
        hip.check(L.fl_debug_gemv_norm_silu(W.handle, xd.data_ptr(), nd.data_ptr(), sd.data_ptr(), act.data_ptr(), None))
        assert np.array_equal(bits(act.cpu().numpy()), bits(want_act[0])), per_wg
        for _ in range(2):
            hip.check(L.fl_debug_gemv_norm_silu_q8(W.handle, xd.data_ptr(), nd.data_ptr(), sd.data_ptr(), a.handle, None))
            a.N, a.K = 1, F
            got = a.export().cpu().numpy()
            assert np.array_equal(got[0], port.quantize_row_q8_0(want_act[0]).reshape(-1)), per_wg
    L.fl_debug_set(7, 0)
    w2 = port.quantize_q4(qt, (rng.standard_normal((E, F)) * 0.05).astype(np.float32))
    W2 = ops.QTensor(qt, w2, E, F)
    res = rng.standard_normal(E).astype(np.float32)
    y = torch.empty(E, device="cuda")
    for form in (1 << 30, 1):                                   # w2 in the K-sliced form and in the one-wave-per-row-group form
        L.fl_debug_set(6, form)
        hip.check(L.fl_debug_gemv_q8(W2.handle, a.handle, y.data_ptr(), dev(torch, res).data_ptr(), None))
        want = (port.mul_mat_q(qt, w2, want_act, strict=False)[0] + res).astype(np.float32)
        assert np.array_equal(bits(y.cpu().numpy()), bits(want)), form


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("D,H,n_past", [(32, 4, 0), (32, 4, 9), (128, 32, 0), (128, 32, 1), (128, 32, 31), (128, 8, 130), (128, 4, 511),
                                        (64, 5, 37), (128, 3, 290), (96, 2, 515), (128, 4, 62), (128, 4, 60), (128, 4, 35), (128, 4, 38), (128, 3, 283),
                                        (128, 2, 255), (64, 3, 285)])
def test_exact_decode_attention(torch, ops, port, exact_hooks, D, H, n_past, split):
    """The single-token attention (one launch, and the two-launch form for long contexts): rope + KV store, K.q and V.p in
    ggml_vec_dot_f32's order (leftover forms included: P = n_past + 1 keys), fp16-table soft_max, Q8_0 of the result --
    the oracle's block bytes."""
    from fastllama_amd import hip
    L = exact_hooks
    n_ctx, E, P = 1024, H * D, n_past + 1
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
    ed, rd, qd, kd, vd = dev(torch, e.view(np.int16)), dev(torch, rt), dev(torch, qkv), dev(torch, kc), dev(torch, vc)
    a = ops.QAct(1, E)
    hip.check(L.fl_quantize_q8_layout(a.handle, qd.data_ptr(), 3 * E, 1, E, 1, None))
    a.N, a.K = 1, E
    scale = np.float32(1.0) / np.sqrt(np.float32(D))
    if split:
        sc = torch.full((H, n_ctx), float("nan"), device="cuda")
        hip.check(L.fl_debug_decode_attention_split(qd.data_ptr(), E, D, H, n_past, n_ctx, rd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                                    ed.data_ptr(), float(scale), sc.data_ptr(), a.handle, None, None))
    else:
        hip.check(L.fl_debug_decode_attention(qd.data_ptr(), E, D, H, n_past, n_ctx, rd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                              ed.data_ptr(), float(scale), a.handle, None))
    q_r = le.rope(qkv[:, :E], n_past, H)
    k_r = le.rope(qkv[:, E:2 * E], n_past, H)
    kc2, vc2 = kd.cpu().numpy(), vd.cpu().numpy()
    assert np.array_equal(bits(kc2[n_past]), bits(k_r[0])) and np.array_equal(vc2[:, n_past], qkv[0, 2 * E:])
    want = np.empty((1, E), np.float32)
    for h in range(H):
        sl = slice(h * D, (h + 1) * D)
        s = (port.mul_mat_f32(np.ascontiguousarray(kc2[:P, sl]), np.ascontiguousarray(q_r[:, sl])) * scale).astype(np.float32)
        p = le.soft_max_rows(s)
        want[:, sl] = port.mul_mat_f32(np.ascontiguousarray(vc2[sl, :P]), p)
    assert np.array_equal(a.export().cpu().numpy()[0], port.quantize_row_q8_0(want[0]))


@pytest.mark.parametrize("D,H,N,n_past", [(128, 4, 64, 0), (128, 3, 100, 37), (128, 2, 512, 0), (64, 5, 70, 11), (96, 2, 33, 200), (32, 4, 40, 0),
                                          (128, 2, 9, 500), (128, 2, 200, 312), (32, 3, 2, 5),
                                          # contexts beyond one 512-key piece of the V.P kernel: the 8 chains of a wave are carried across pieces
                                          (128, 2, 70, 600), (64, 3, 33, 1000), (128, 1, 512, 512), (128, 2, 100, 1947), (32, 2, 40, 1003),
                                          (128, 1, 64, 448), (96, 2, 31, 993),
                                          # ... with the probabilities compact between soft_max and V.P: whole and ragged last pieces, P = n_ctx (the row's
                                          # factor sits in the float a score held), ragged query blocks, every leftover form again
                                          (128, 2, 512, 1536), (128, 1, 40, 984), (64, 2, 100, 413), (128, 1, 33, 480), (32, 2, 70, 1951), (128, 2, 9, 504),
                                          (96, 1, 48, 977), (128, 1, 5, 2043),
                                          # beyond 2048 keys: f32 probabilities again (the loop form of soft_max), nine and ten pieces of 256 keys
                                          (64, 1, 40, 2260), (128, 1, 33, 2535)])
def test_exact_prefill_attention_mfma_forms(torch, ops, port, D, H, N, n_past):
    """K.Q and V.P of a batch on the f32-input MFMA (its k = 0, 1 chain is the reference's fma chain): the same bits as the
    one-half-wave-per-dot kernel and as the oracle -- every leftover form (P % 32), causal tiles, ragged last query block, and
    contexts of several 512-key pieces; beyond 512 keys also the form the model runs (round 6): probabilities as fp16 table values + one
    factor per row between the launches, K.Q and soft_max as one launch up to 1024 keys."""
    from fastllama_amd import hip
    L = hip.load()
    E, P = H * D, n_past + N
    n_ctx = 512 if P <= 512 else (P + 63) // 64 * 64
    rng = np.random.default_rng(D + H + N + n_past)
    qkv = rng.standard_normal((N, 3 * E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32)
    vc = np.full((E, n_ctx), np.nan, np.float32)              # whatever lies beyond the context must not leak in (nor be multiplied by a zero)
    kc[:P] = rng.standard_normal((P, E))
    vc[:, :P] = rng.standard_normal((E, P))
    e = np.empty(1 << 16, np.uint16)
    L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    ed, qd, kd, vd = dev(torch, e.view(np.int16)), dev(torch, qkv), dev(torch, kc), dev(torch, vc)
    scale = np.float32(1.0) / np.sqrt(np.float32(D))
    outs = []
    compact_forms = ((2,) if 512 < P <= 2048 else ()) + ((3,) if 512 < P <= 1024 else ())
    # (beyond 512 keys V.P has two forms -- eight waves per workgroup, the default, and four: 12 / 11 stand for 2 / 1 with the four-wave form)
    four_wave = ((12,) if 512 < P <= 2048 else ()) + ((11,) if P > 512 else ())
    for which in four_wave + compact_forms + (0, 1):
        L.fl_debug_set(8, 4 if which >= 10 else 8)
        which %= 10
        att = torch.full((H, N, n_ctx), float("nan"), device="cuda")
        ao = torch.full((N, E), 3.0, device="cuda")
        hip.check(L.fl_debug_attn_exact(qd.data_ptr(), 3 * E, D, H, N, n_past, n_ctx, E, kd.data_ptr(), vd.data_ptr(), ed.data_ptr(),
                                        float(scale), att.data_ptr(), ao.data_ptr(), which, None))
        torch.cuda.synchronize()
        outs.append(ao.cpu().numpy())
    L.fl_debug_set(8, 8)
    for o in outs[:-1]:
        assert np.array_equal(bits(o), bits(outs[-1])), int((bits(o) != bits(outs[-1])).sum())
    outs = outs[-2:]
    want = np.empty((N, E), np.float32)
    for h in range(H):
        sl = slice(h * D, (h + 1) * D)
        s = (port.mul_mat_f32(np.ascontiguousarray(kc[:P, sl]), np.ascontiguousarray(qkv[:, sl])) * scale).astype(np.float32)
        s[np.arange(P)[None, :] > (n_past + np.arange(N))[:, None]] = -np.inf
        want[:, sl] = port.mul_mat_f32(np.ascontiguousarray(vc[sl, :P]), le.soft_max_rows(s))
    assert np.array_equal(bits(outs[1]), bits(want))
    # the form the model runs for N >= 9: P.V writes the Q8_0 operand of the wo matmul itself (att still holds the probabilities) ...
    a = ops.QAct(N, E)
    hip.check(L.fl_debug_attn_pv_exact_q8(att.data_ptr(), n_ctx, D, H, N, n_past, vd.data_ptr(), E, a.handle, 0, None))
    a.N, a.K = N, E
    got_q = a.export().cpu().numpy()
    for n in range(N):
        assert np.array_equal(got_q[n], port.quantize_row_q8_0(want[n])), n
    if compact_forms:                                         # ... from compact probabilities
        att = torch.full((H, N, n_ctx), float("nan"), device="cuda")
        ao = torch.full((N, E), 3.0, device="cuda")
        hip.check(L.fl_debug_attn_exact(qd.data_ptr(), 3 * E, D, H, N, n_past, n_ctx, E, kd.data_ptr(), vd.data_ptr(), ed.data_ptr(),
                                        float(scale), att.data_ptr(), ao.data_ptr(), 2, None))
        a = ops.QAct(N, E)
        hip.check(L.fl_debug_attn_pv_exact_q8(att.data_ptr(), n_ctx, D, H, N, n_past, vd.data_ptr(), E, a.handle, 1, None))
        a.N, a.K = N, E
        assert np.array_equal(a.export().cpu().numpy(), got_q)
    # ... and its f16 fragment copy, which the reference-order GEMM reads (q4_layout.h XH16): a matmul on it equals the oracle's
    if N >= 9:
        wq = port.quantize_q4(oracle.Q4_0, (rng.standard_normal((80, E)) * 0.05).astype(np.float32))
        W = ops.QTensor(oracle.Q4_0, wq, 80, E)
        y = torch.empty((N, 80), device="cuda")
        L.fl_set_op_mode(1)
        try:
            hip.check(L.fl_mul_mat_q(W.handle, a.handle, y.data_ptr(), 80, None))
        finally:
            L.fl_set_op_mode(-1)
        assert np.array_equal(bits(y.cpu().numpy()), bits(port.mul_mat_q(oracle.Q4_0, wq, want, strict=False)))
