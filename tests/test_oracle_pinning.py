"""CPU: the C restatement (oracle/q4_oracle.c) is pinned to the reference.

(1) against the committed golden vectors, which were produced by the reference itself
    (tests/golden/make_golden.py -> oracle/_ref), and
(2) when oracle/_ref is available (this container: /root/reference present), live against the
    compiled reference on fresh random inputs -- bit-for-bit in both cases.
"""
import numpy as np
import pytest

import oracle
from util import bits, golden


@pytest.fixture(scope="module")
def port():
    return oracle.Port()


@pytest.fixture(scope="module")
def ref():
    if not oracle.have_ref():
        try:
            oracle.build()
        except Exception:
            pass
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return oracle.Ref()


@pytest.mark.parametrize("tag", ["s", "m"])
def test_port_q8_matches_golden(port, tag):
    g = golden()
    for n, row in enumerate(g[f"{tag}_x"]):
        assert np.array_equal(port.quantize_row_q8_0(row), g[f"{tag}_q8"][n]), (tag, n)


@pytest.mark.parametrize("nm,qt", [("q40", oracle.Q4_0), ("q41", oracle.Q4_1)])
def test_port_weight_quantizer_and_dequant_match_golden(port, nm, qt):
    g = golden()
    wq = port.quantize_q4(qt, g["s_w"])
    assert np.array_equal(wq, g[f"s_{nm}"])
    deq = port.dequantize(qt, wq, g["s_w"].shape[1])
    assert np.array_equal(bits(deq), bits(g[f"s_{nm}_deq"]))


@pytest.mark.parametrize("tag", ["s", "m"])
@pytest.mark.parametrize("nm,qt", [("q40", oracle.Q4_0), ("q41", oracle.Q4_1)])
def test_port_vec_dot_and_mul_mat_match_golden(port, tag, nm, qt):
    g = golden()
    wq, x = g[f"{tag}_{nm}"], g[f"{tag}_x"]
    K = x.shape[1]
    y = port.mul_mat_q(qt, wq, x, n_threads=2)
    assert np.array_equal(bits(y), bits(g[f"{tag}_{nm}_y"]))
    # mul_mat is, by construction, vec_dot per (row, column): lib/ggml.c:8160-8162
    assert np.array_equal(bits(g[f"{tag}_{nm}_y"]), bits(g[f"{tag}_{nm}_vd"]))
    for n in range(x.shape[0]):
        for m in range(0, wq.shape[0], 7):
            v = port.vec_dot(qt, K, wq[m], g[f"{tag}_q8"][n])
            assert v.tobytes() == g[f"{tag}_{nm}_vd"][n, m].tobytes()


def test_mul_mat_rejects_odd_block_count(port):
    # assert(nb % 2 == 0), lib/ggml.c:2372 -> K must be a multiple of 64
    w = np.zeros((4, 96), dtype=np.float32)
    wq = port.quantize_q4(oracle.Q4_0, w)
    with pytest.raises(ValueError):
        port.mul_mat_q(oracle.Q4_0, wq, np.zeros((1, 96), dtype=np.float32))


@pytest.mark.parametrize("qt", [oracle.Q4_0, oracle.Q4_1])
@pytest.mark.parametrize("seed", [0, 1])
def test_port_matches_live_reference(port, ref, qt, seed):
    rng = np.random.default_rng(100 + seed)
    M, K, N = 40, 1024, 5
    w = rng.normal(0, 0.02, (M, K)).astype(np.float32)
    x = rng.normal(0, 1, (N, K)).astype(np.float32)
    x[0, :64] = 0
    x[1, 100] = -7e20
    a, b = port.quantize_q4(qt, w), ref.quantize_q4(qt, w)
    assert np.array_equal(a, b)
    for n in range(N):
        assert np.array_equal(port.quantize_row_q8_0(x[n]), ref.quantize_row_q8_0(x[n], qt))
    assert np.array_equal(bits(port.dequantize(qt, a, K)),
                          bits(np.stack([ref.dequantize_row(qt, r, K) for r in a])))
    ya, yb = port.mul_mat_q(qt, a, x, n_threads=2), ref.mul_mat_q(qt, a, x, n_threads=3)
    assert np.array_equal(bits(ya), bits(yb))


@pytest.mark.parametrize("qt", [oracle.Q4_0, oracle.Q4_1])
@pytest.mark.parametrize("reference", [False, True])
def test_row_quantizer_flavours_pinned_to_reference_table(port, ref, qt, reference):
    """quantize_fns[type].quantize_row_q (the SIMD flavour of this x86 build) and .quantize_row_q_reference
    (lib/ggml.c:1731-1745) against the restatements the GPU kernels are tested with -- incl. x.5 ties, where rint and
    roundf part ways."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(32 * 40).astype(np.float32)
    x[32:64] = 0.0
    x[64:96] = np.concatenate([[7.0, -7.0], (np.arange(30) % 14 - 7) + 0.5]).astype(np.float32)
    x[96:128] = np.concatenate([[0.0, 15.0], np.arange(30) % 15 + 0.5]).astype(np.float32)
    x[128:160] *= 1e-30
    x[160:192] *= 1e30
    assert np.array_equal(port.quantize_row_q4(qt, x, reference), ref.quantize_row_q4(qt, x, reference))
    if qt == oracle.Q4_0:   # the two flavours do differ on ties (this is what the two table slots are for)
        assert not np.array_equal(port.quantize_row_q4(qt, x, True), port.quantize_row_q4(qt, x, False))


def test_reference_result_independent_of_thread_count(ref, port):
    # each output element is computed by exactly one thread in a fixed order (lib/ggml.c:8127-8163)
    wq = port.quantize_q4(oracle.Q4_0, np.random.default_rng(5).normal(0, .02, (37, 512)).astype(np.float32))
    x = np.random.default_rng(6).normal(0, 1, (4, 512)).astype(np.float32)
    y1 = ref.mul_mat_q(oracle.Q4_0, wq, x, n_threads=1)
    y4 = ref.mul_mat_q(oracle.Q4_0, wq, x, n_threads=4)
    assert np.array_equal(bits(y1), bits(y4))


@pytest.mark.parametrize("qtype", [2, 3])
@pytest.mark.parametrize("r", [4, 16, 32, 72])
def test_lora_merge_restatement_pinned_to_reference(port, ref, qtype, r):
    """oracle orc_lora_add (vec_dot_f32 lane order, dequantize + add + SIMD quantizer) == the reference's own
    ggml_mul_mat(loraA, loraB) -> ggml_add_inplace(W_q4, BA) graph, byte for byte; attach, detach (sign -1) and the
    cached-matrix form."""
    rng = np.random.default_rng(100 * qtype + r)
    M, K = 48, 256
    wq = port.quantize_q4(qtype, (rng.standard_normal((M, K)) * 0.05).astype(np.float32))
    a = (rng.standard_normal((K, r)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((M, r)) * 0.1).astype(np.float32)
    w_ref, ba_ref = ref.lora_add(qtype, wq, K, a=a, b=b)
    w_orc, ba_orc = port.lora_add(qtype, wq, K, a=a, b=b)
    assert np.array_equal(ba_ref.view(np.uint32), ba_orc.view(np.uint32))
    assert np.array_equal(w_ref, w_orc) and not np.array_equal(w_ref, wq)
    w_ref2, _ = ref.lora_add(qtype, w_ref, K, a=a, b=b, sign=-1.0)            # detach without mmap: W - BA, requantized
    w_orc2, _ = port.lora_add(qtype, w_orc, K, a=a, b=b, sign=-1.0)
    assert np.array_equal(w_ref2, w_orc2)
    w_ref3, _ = ref.lora_add(qtype, wq, K, ba=ba_ref)                         # cached adapter
    w_orc3, _ = port.lora_add(qtype, wq, K, ba=ba_ref)
    assert np.array_equal(w_ref3, w_orc3) and np.array_equal(w_ref3, w_ref)
