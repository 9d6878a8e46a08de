"""GPU: LoRA attach / detach on the resident Q4 weights.

Byte-exact against the CPU restatement of the reference's merge (oracle.Port.lora_add, itself pinned byte for byte to the
reference's ggml graph in tests/test_oracle_pinning.py), and end to end through the llama_* C-ABI against the reference
library on the same model + adapter files."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from harness import ggjt, llama_capi

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fast_mode")]   # (conftest.py: the fast kernels, explicitly)
OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastllama_amd", "libfastllama_hip.so")
TEXT = "The quick brown fox jumps over the lazy dog; 0123456789 times!?"


@pytest.fixture(scope="module")
def port():
    return oracle.Port()


def download(L, m, name, rows, K, qtype):
    from fastllama_amd import hip
    out = np.empty((rows, K // 32 * ggjt.BLOCK_BYTES[qtype]), np.uint8)
    hip.check(L.fl_model_tensor_download(m.h, name.encode(), out.ctypes.data_as(C.c_void_p)), "download " + name)
    return out


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("qtype", [ggjt.Q4_0, ggjt.Q4_1])
@pytest.mark.parametrize("r", [8, 40])
def test_lora_merge_bit_exact_vs_oracle(port, qtype, r):
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    E, F = cfg["n_embd"], ggjt.n_ff_of(cfg["n_embd"], cfg["n_mult"])
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=99)
    m = FlModel(cfg, qtype, tensors, n_ctx=64, max_batch=64)
    rng = np.random.default_rng(r + qtype)
    cases = {"layers.1.attention.wk.weight": (E, E), "layers.0.feed_forward.w3.weight": (F, E),
             "layers.2.feed_forward.w2.weight": (E, F), "layers.2.attention.wo.weight": (E, E), "output.weight": (cfg["n_vocab"], E)}
    for name, (M, K) in cases.items():
        orig = np.ascontiguousarray(tensors[name][2]).reshape(M, -1)
        assert np.array_equal(download(L, m, name, M, K, qtype), orig)
        a = (rng.standard_normal((K, r)) * 0.05).astype(np.float32)
        b = (rng.standard_normal((M, r)) * 0.05).astype(np.float32)
        hip.check(L.fl_model_lora_apply(m.h, name.encode(), None, ptr(a), ptr(b), r, 1.0, 1), "lora_apply")
        want, ba = port.lora_add(qtype, orig, K, a=a, b=b)
        got = download(L, m, name, M, K, qtype)
        assert np.array_equal(got, want), name
        assert not np.array_equal(got, orig)
        hip.check(L.fl_model_lora_apply(m.h, name.encode(), None, ptr(a), ptr(b), r, -1.0, 0), "lora_apply(-1)")   # lossy detach
        want2, _ = port.lora_add(qtype, want, K, a=a, b=b, sign=-1.0)
        assert np.array_equal(download(L, m, name, M, K, qtype), want2), name
    # the neighbours inside the fused tensors were not touched
    for name in ("layers.1.attention.wq.weight", "layers.1.attention.wv.weight", "layers.0.feed_forward.w1.weight"):
        M, K = (E, E) if "attention" in name else (F, E)
        assert np.array_equal(download(L, m, name, M, K, qtype), np.ascontiguousarray(tensors[name][2]).reshape(M, -1))
    # cached form == uncached form when BA is the same matrix
    name, (M, K) = "layers.0.attention.wq.weight", (E, E)
    orig = np.ascontiguousarray(tensors[name][2]).reshape(M, -1)
    a = (rng.standard_normal((K, r)) * 0.05).astype(np.float32)
    b = (rng.standard_normal((M, r)) * 0.05).astype(np.float32)
    want, ba = port.lora_add(qtype, orig, K, a=a, b=b)
    hip.check(L.fl_model_lora_apply(m.h, name.encode(), ptr(ba), None, None, 0, 1.0, 1), "lora_apply(cached)")
    assert np.array_equal(download(L, m, name, M, K, qtype), want)
    # restore puts every original back, byte for byte
    hip.check(L.fl_model_lora_restore(m.h), "restore")
    for name, (M, K) in list(cases.items()) + [("layers.0.attention.wq.weight", (E, E))]:
        assert np.array_equal(download(L, m, name, M, K, qtype), np.ascontiguousarray(tensors[name][2]).reshape(M, -1)), name
    assert L.fl_model_lora_apply(m.h, b"layers.0.attention_norm.weight", None, ptr(a), ptr(b), r, 1.0, 0) != 0
    m.free()


@pytest.mark.parametrize("qtype", [ggjt.Q4_0, ggjt.Q4_1])
def test_derived_weight_copies_follow_the_weights_through_merge_and_restore(port, qtype):
    """The reference-order kernels read DERIVED copies of the nibbles (q4_layout.h: the f16 fragment copy H16 for N >= 9, the QWD copy for
    N = 1), built on first use.  A LoRA merge rewrites the nibbles of record; the copies must follow: after the merge the model's logits
    (a 96-token batch and a decode step) equal, bit for bit, those of a FRESH model created from the merged tensor -- and after the restore
    those of the original."""
    import torch
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    E = cfg["n_embd"]
    N, r, name = 96, 8, "layers.1.attention.wo.weight"
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=4)
    toks = np.random.default_rng(2).integers(3, 259, N).astype(np.int32)
    rng = np.random.default_rng(qtype)
    a = (rng.standard_normal((E, r)) * 0.05).astype(np.float32)
    b = (rng.standard_normal((E, r)) * 0.05).astype(np.float32)

    def run(model):
        pre = model.eval(toks, all_logits=True).copy()
        dec = model.eval([int(toks[5])], n_past=N).copy()
        return pre, dec

    m = FlModel(cfg, qtype, tensors, n_ctx=128, max_batch=N)
    m.set_exact(True)
    base = run(m)                                               # (builds both derived copies)
    hip.check(L.fl_model_lora_apply(m.h, name.encode(), None, ptr(a), ptr(b), r, 1.0, 1), "lora_apply")
    merged = run(m)
    merged_blocks = download(L, m, name, E, E, qtype)
    hip.check(L.fl_model_lora_restore(m.h), "restore")
    restored = run(m)
    m.free()
    t2 = dict(tensors)
    ent = list(t2[name])
    ent[2] = merged_blocks.reshape(np.asarray(ent[2]).shape)
    t2[name] = tuple(ent)
    m2 = FlModel(cfg, qtype, t2, n_ctx=128, max_batch=N)
    m2.set_exact(True)
    fresh = run(m2)
    m2.free()
    for got, want in zip(merged, fresh):
        assert np.array_equal(got.view(np.int32), want.view(np.int32))
    for got, want in zip(restored, base):
        assert np.array_equal(got.view(np.int32), want.view(np.int32))
    assert not np.array_equal(base[0], merged[0])
    torch.cuda.empty_cache()


def test_lora_merge_under_tensor_parallel_slices(port):
    """rank 1 of 2: the row shard (wq, w1) takes B's rows, the K shard (wo, w2) takes A's rows -- against the oracle
    applied to the full tensor and then sliced the way fl_model_set_tensor slices."""
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg, qtype, G, rank, r = ggjt.SMALL, ggjt.Q4_0, 2, 1, 8
    E, F = cfg["n_embd"], ggjt.n_ff_of(cfg["n_embd"], cfg["n_mult"])
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=5)
    m = FlModel(cfg, qtype, tensors, n_ctx=64, max_batch=64, tp_rank=rank, tp_size=G)
    rng = np.random.default_rng(0)
    for name, (M, K), by_rows in (("layers.0.attention.wv.weight", (E, E), True), ("layers.1.feed_forward.w1.weight", (F, E), True),
                                  ("layers.1.attention.wo.weight", (E, E), False), ("layers.2.feed_forward.w2.weight", (E, F), False)):
        orig = np.ascontiguousarray(tensors[name][2]).reshape(M, -1)
        a = (rng.standard_normal((K, r)) * 0.05).astype(np.float32)
        b = (rng.standard_normal((M, r)) * 0.05).astype(np.float32)
        hip.check(L.fl_model_lora_apply(m.h, name.encode(), None, ptr(a), ptr(b), r, 1.0, 0), "lora_apply")
        full, _ = port.lora_add(qtype, orig, K, a=a, b=b)
        bs = ggjt.BLOCK_BYTES[qtype]
        if by_rows:
            want = full[rank * (M // G):(rank + 1) * (M // G)]
            got = download(L, m, name, M // G, K, qtype)
        else:
            kb = K // 32 // G
            want = full.reshape(M, K // 32, bs)[:, rank * kb:(rank + 1) * kb].reshape(M, -1)
            got = download(L, m, name, M, K // G, qtype)
        assert np.array_equal(got, want), name
    m.free()


@pytest.mark.parametrize("cached", [False, True])
@pytest.mark.parametrize("use_mmap", [False, True])
def test_llama_attach_detach_lora_matches_reference(tmp_path_factory, port, cached, use_mmap):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    cfg, qtype, r, alpha = ggjt.SMALL, ggjt.Q4_0, 8, 16
    E, F = cfg["n_embd"], ggjt.n_ff_of(cfg["n_embd"], cfg["n_mult"])
    d = tmp_path_factory.mktemp("lora")
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=4321)
    mpath, lpath = str(d / "m.bin"), str(d / "adapter.bin")
    ggjt.write_ggjt(mpath, cfg, qtype, tensors)
    rng = np.random.default_rng(7)
    adapters = {}
    for l in range(cfg["n_layer"]):
        for sub, (M, K) in (("attention.wq", (E, E)), ("attention.wv", (E, E)), ("feed_forward.w2", (E, F))):
            a = (rng.standard_normal((K, r)) * 0.08 * alpha / r).astype(np.float32)
            b = (rng.standard_normal((M, r)) * 0.08).astype(np.float32)
            adapters[f"layers.{l}.{sub}.weight"] = (b @ a.T).astype(np.float32) if cached else (a, b)
    ggjt.write_lora(lpath, adapters, r, alpha, cached=cached)
    text = TEXT[:48]
    res = {}
    for tag, lib in (("ref", os.path.join(oracle.REF_DIR, "pyfastllama.so")), ("ours", OURS)):
        s = llama_capi.Session(llama_capi.LlamaLib(lib), mpath, n_ctx=128, n_batch=64, all_logits=True, use_mmap=use_mmap)
        assert not s.detach_lora()                                   # nothing attached yet
        p0, l0 = s.perplexity(text), s.logits().copy()
        assert s.attach_lora(lpath)
        assert not s.attach_lora(lpath)                              # "already attached"
        p1, l1 = s.perplexity(text), s.logits().copy()
        assert s.detach_lora()
        p2, l2 = s.perplexity(text), s.logits().copy()
        assert not s.detach_lora()
        assert s.attach_lora(lpath)                                  # and again after a detach
        p3 = s.perplexity(text)
        res[tag] = (p0, l0, p1, l1, p2, l2, p3)
        s.close()
    (rp0, rl0, rp1, rl1, rp2, rl2, rp3), (p0, l0, p1, l1, p2, l2, p3) = res["ref"], res["ours"]
    assert abs(rp1 - rp0) / rp0 > 1e-3                               # the adapter really changes the model
    for mine, ref in ((p0, rp0), (p1, rp1), (p2, rp2), (p3, rp3)):
        assert abs(mine - ref) / ref <= 3e-2, (mine, ref)
    n = len(ggjt.text_tokens(text))
    for mine, ref in ((l0, rl0), (l1, rl1), (l2, rl2)):
        mine, ref = mine.reshape(n, -1), ref.reshape(n, -1).astype(np.float64)
        per_pos = np.max(np.abs(mine - ref), axis=1) / np.max(np.abs(ref))
        assert per_pos[0] <= 1e-5 and per_pos.max() <= 5e-2, per_pos
    if use_mmap:
        assert np.array_equal(l2, l0) and p2 == p0                   # originals restored exactly
    else:
        assert not np.array_equal(l2, l0)                            # W - BA re-quantized: close, not identical
