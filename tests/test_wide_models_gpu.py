"""GPU: the eval at the WIDTHS of BASELINE.json's configs 4 and 5 -- LLaMA-13B (n_embd 5120, 40 heads, n_ff 13824) and LLaMA-65B
(n_embd 8192, 64 heads, n_ff 22016; shapes from /root/reference/lib/llama.cpp:129-140) -- as 2-layer models with the real
vocabulary, against the LIVE reference (oracle/_ref through its own C-ABI), and the tensor-parallel split of those widths at the
degrees the configs name (13B: 2 and 4 shards, 65B: 8 shards) as G logical shards on one device.

Kernel shapes at these widths are covered bit for bit in tests/test_kernels_gpu.py (LLAMA_SHAPES); this file covers what only a
model exercises: 40 / 64 heads, the head -> shard mapping, K-block shards of wo / w2 with odd block counts (13824 / 4 / 32 = 108,
22016 / 8 / 32 = 86), the row-split lm-head of 32000 / G rows -- and, in the default (reference-order) mode, the ROW split of every matmul, whose shards
return the reference's logits bit for bit.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle
from harness import ggjt, llama_capi
from util import bits

pytestmark = pytest.mark.gpu

WIDTHS = {
    "13B": dict(n_embd=5120, n_head=40, n_mult=256),
    "65B": dict(n_embd=8192, n_head=64, n_mult=256),
}


@pytest.fixture(scope="module")
def reflib():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    return llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))


def _cfgs(width, n_vocab, n_layer=2):
    w = WIDTHS[width]
    cfg = dict(n_vocab=n_vocab, n_embd=w["n_embd"], n_mult=w["n_mult"], n_head=w["n_head"], n_layer=n_layer)
    scfg = dict(n_embd=w["n_embd"], n_head=w["n_head"], n_layer=n_layer, n_ff=ggjt.n_ff_of(w["n_embd"], w["n_mult"]), n_vocab=n_vocab)
    return cfg, scfg


@pytest.mark.parametrize("width", ["13B", "65B"])
def test_wide_model_logits_vs_reference(tmp_path_factory, reflib, width):
    """N = 128 tokens in one batch.  Exact mode: every logit bit-identical to the reference.  Fast mode: the assertions of
    test_llama7b_width_logits (bounded deviation on random weights after 2 layers, greedy agreement, perplexity)."""
    import torch
    from harness import synth
    from harness.flmodel import FlModel
    qtype, N = ggjt.Q4_0, 128
    cfg, scfg = _cfgs(width, 32000)
    assert scfg["n_ff"] == synth.MODELS[width]["n_ff"]
    gen = lambda: synth.synth_model_tensors(scfg, qtype, seed=77)
    path = str(tmp_path_factory.mktemp("wide") / f"w{width}.bin")
    ggjt.write_ggjt_stream(path, cfg, qtype, gen())
    rng = np.random.default_rng(5)
    text = bytes(rng.integers(33, 127, size=N - 2).astype(np.uint8)).decode()
    toks = [1] + [b + 3 for b in (" " + text).encode()]
    assert len(toks) == N
    ref = llama_capi.Session(reflib, path, n_ctx=256, n_batch=N, n_threads=min(32, os.cpu_count() or 8), all_logits=True)
    assert ref.ingest(text) and ref.generate(1, temp=0.0)[0]
    want = ref.logits().reshape(N, cfg["n_vocab"]).copy()
    ref.close()
    os.remove(path)
    m = FlModel(scfg, qtype, gen(), n_ctx=256, max_batch=N)
    m.set_exact(True)
    got_x = m.eval(toks, all_logits=True)
    assert np.array_equal(bits(got_x), bits(want)), float(np.abs(got_x - want).max() / np.abs(want).max())
    m.set_exact(False)
    got = m.eval(toks, all_logits=True)
    m.free()
    torch.cuda.empty_cache()
    per_pos = np.max(np.abs(got.astype(np.float64) - want), axis=1) / np.max(np.abs(want))
    assert per_pos.max() <= 5e-2, (per_pos[0], per_pos.max())     # (at these widths a Q8_0 flip already moves position 0 by 1e-2)
    assert np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want) <= 2e-2
    assert np.mean(np.argmax(got, axis=1) == np.argmax(want, axis=1)) >= 0.9


@pytest.mark.usefixtures("fast_mode")
@pytest.mark.parametrize("width,G", [("13B", 2), ("13B", 4), ("65B", 8)])
def test_tensor_parallel_shards_at_config_widths(width, G):
    """FAST mode (conftest.py fast_mode).  The Megatron split of SURVEY.md 8(e) at the widths and degrees of BASELINE configs 4 / 5, as G shards on ONE device
    (fl_comm_create_local): every shard ends with the same logits (row-split lm-head + all-gather), and they agree with the
    unsharded model up to the order of the G partial sums and the Q8_0 flips that order causes downstream."""
    import torch
    from fastllama_amd import hip
    from harness import synth
    from harness.flmodel import FlModel
    L = hip.load()
    qtype, N = ggjt.Q4_0, 48
    _, scfg = _cfgs(width, 2048)
    assert scfg["n_head"] % G == 0 and (scfg["n_ff"] // G) % 32 == 0 and scfg["n_vocab"] % G == 0
    tensors = list(synth.synth_model_tensors(scfg, qtype, seed=5))
    toks = np.random.default_rng(3).integers(3, 259, N).astype(np.int32)
    full = FlModel(scfg, qtype, tensors, n_ctx=64, max_batch=N)
    want = full.eval(toks, all_logits=True)
    want_dec = full.eval([int(toks[3])], n_past=N)
    full.free()
    comms = (C.c_void_p * G)()
    hip.check(L.fl_comm_create_local(G, comms), "fl_comm_create_local")
    shards = []
    for r in range(G):
        m = FlModel(scfg, qtype, tensors, n_ctx=64, max_batch=N, tp_rank=r, tp_size=G)
        m.set_comm(C.c_void_p(comms[r]))
        shards.append(m)
    del tensors
    got, got_dec, errs = [None] * G, [None] * G, []

    def run(r):
        try:
            got[r] = shards[r].eval(toks, all_logits=True)
            got_dec[r] = shards[r].eval([int(toks[3])], n_past=N)
        except Exception as e:           # a failing shard must not leave the others waiting forever
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs and all(not t.is_alive() for t in th), errs
    for r in range(1, G):
        assert np.array_equal(got[r], got[0]) and np.array_equal(got_dec[r], got_dec[0])
    scale = np.max(np.abs(want))
    per_pos = np.max(np.abs(got[0].astype(np.float64) - want), axis=1) / scale
    assert per_pos.max() <= 5e-2, (per_pos[0], per_pos.max())
    assert np.max(np.abs(got_dec[0].astype(np.float64) - want_dec)) / np.max(np.abs(want_dec)) <= 5e-2
    for m in shards:
        m.free()
    for r in range(G):
        L.fl_comm_destroy(C.c_void_p(comms[r]))
    torch.cuda.empty_cache()


def _run_shards(shards, fn):
    """fn(r) on every shard at once (the single-process group rendezvouses on the host); a failing shard must not leave the others waiting"""
    out, errs = [None] * len(shards), []

    def run(r):
        try:
            out[r] = fn(r)
        except Exception as e:
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(len(shards))]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs and all(not t.is_alive() for t in th), errs
    return out


@pytest.mark.parametrize("qname,qtype", [("q40", ggjt.Q4_0), ("q41", ggjt.Q4_1)])
@pytest.mark.parametrize("width,G", [("13B", 2), ("13B", 4), ("65B", 8)])
def test_row_split_tensor_parallel_is_bit_identical_to_the_reference(tmp_path_factory, reflib, width, G, qname, qtype):
    """The default (reference-order) mode shards EVERY matmul by output rows -- the reference's own split across threads
    (/root/reference/lib/ggml.c:8127-8135) -- and all-gathers the Q8_0 operands of wo / w2 and the output rows; nothing is summed
    across ranks.  At the widths and degrees of BASELINE configs 4 / 5, as G shards on one device: every shard's logits of a
    48-token batch equal the LIVE reference's bit for bit; a 5-token eval (wave kernels, QA16 operands without the H16 copy) and
    single-token decode steps equal the unsharded model's (itself the reference's, tests/test_exact_gpu.py)."""
    import torch
    from fastllama_amd import hip
    from harness import synth
    from harness.flmodel import FlModel
    if qtype == ggjt.Q4_1 and (width, G) != ("13B", 4):
        pytest.skip("Q4_1 at one width / degree")
    L = hip.load()
    assert L.fl_default_exact() == 1
    N, V = 48, 2048
    cfg, scfg = _cfgs(width, V)
    assert scfg["n_head"] % G == 0 and (scfg["n_ff"] // G) % 32 == 0 and V % G == 0
    gen = lambda: synth.synth_model_tensors(scfg, qtype, seed=21)
    path = str(tmp_path_factory.mktemp("tprows") / f"w{width}.bin")
    ggjt.write_ggjt_stream(path, cfg, qtype, gen())
    rng = np.random.default_rng(8)
    text = bytes(rng.integers(33, 127, size=N - 2).astype(np.uint8)).decode()
    toks = np.array([1] + [b + 3 for b in (" " + text).encode()], dtype=np.int32)
    assert len(toks) == N
    ref = llama_capi.Session(reflib, path, n_ctx=128, n_batch=N, n_threads=min(32, os.cpu_count() or 8), all_logits=True)
    assert ref.ingest(text) and ref.generate(1, temp=0.0)[0]
    want = ref.logits().reshape(N, V).copy()
    ref.close()
    os.remove(path)
    tensors = list(gen())
    full = FlModel(scfg, qtype, tensors, n_ctx=128, max_batch=N)
    assert np.array_equal(bits(full.eval(toks, all_logits=True)), bits(want))
    want5 = full.eval(toks[:5], n_past=N, all_logits=True)
    want_dec = [full.eval([int(toks[3 + i])], n_past=N + 5 + i) for i in range(3)]
    full.free()
    comms = (C.c_void_p * G)()
    hip.check(L.fl_comm_create_local(G, comms), "fl_comm_create_local")
    shards = []
    for r in range(G):
        m = FlModel(scfg, qtype, tensors, n_ctx=128, max_batch=N, tp_rank=r, tp_size=G)
        m.set_comm(C.c_void_p(comms[r]))
        shards.append(m)
    del tensors
    got = _run_shards(shards, lambda r: shards[r].eval(toks, all_logits=True))
    for r in range(G):
        assert np.array_equal(bits(got[r]), bits(want)), (r, int((bits(got[r]) != bits(want)).sum()))
    got5 = _run_shards(shards, lambda r: shards[r].eval(toks[:5], n_past=N, all_logits=True))
    for r in range(G):
        assert np.array_equal(bits(got5[r]), bits(want5)), r
    for i in range(3):
        gd = _run_shards(shards, lambda r: shards[r].eval([int(toks[3 + i])], n_past=N + 5 + i))
        for r in range(G):
            assert np.array_equal(bits(gd[r]), bits(want_dec[i])), (r, i)
    for m in shards:
        m.free()
    for r in range(G):
        L.fl_comm_destroy(C.c_void_p(comms[r]))
    torch.cuda.empty_cache()
