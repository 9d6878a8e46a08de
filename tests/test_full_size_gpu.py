"""GPU: BASELINE.json's configs 4 and 5 at their REAL size against the LIVE reference (oracle/_ref through its own C-ABI).

tests/test_wide_models_gpu.py runs the 13B / 65B WIDTHS as 2-layer models with short batches; this file closes what that leaves open
(missing 2):

* LLaMA-13B at full depth -- 40 layers (/root/reference/lib/llama.cpp:129-140), n_batch 512, the same GGJT file through the compiled
  reference and the GPU: every one of the 512 x 32000 logits bit-identical in the default (reference-order) mode.
* LLaMA-65B width (n_embd 8192, 64 heads, n_ff 22016) with 8 layers at n_ctx = 2048: a prompt of 2040 tokens ingested as four n_batch
  evals (the last one's queries attend over 2040 keys: the piece-by-piece V.P loop of the reference-order attention, every leftover
  form of ggml_vec_dot_f32), then greedy decode steps at positions 2040 .. 2046 (the split decode attention over 2041 .. 2047 keys, 64
  heads) -- every step's logits equal the reference's bits, unsharded AND as G = 8 row-split shards (config 5's degree) on one device.
* Q4_1 at 65B width (skipped in test_wide_models_gpu.py's row-split matrix): the unsharded 2-layer model, N = 128, once.
"""
import ctypes as C
import os
import threading
import time

import numpy as np
import pytest

import oracle
from harness import ggjt, llama_capi
from util import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reflib():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    return llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))


def _prompt(n_tokens, seed):
    """a random printable-ASCII prompt whose llama_ingest tokenisation is exactly n_tokens (BOS + the inserted space + the bytes)"""
    rng = np.random.default_rng(seed)
    text = bytes(rng.integers(33, 127, size=n_tokens - 2).astype(np.uint8)).decode()
    toks = np.array([1] + [b + 3 for b in (" " + text).encode()], dtype=np.int32)
    assert len(toks) == n_tokens
    return text, toks


def test_llama13b_full_depth_nbatch512_vs_reference(tmp_path_factory, reflib):
    """BASELINE config 4's model on one GPU: 40 layers, n_embd 5120, 40 heads, n_ff 13824, n_vocab 32000, one n_batch = 512 eval."""
    import torch
    from harness import synth
    from harness.flmodel import FlModel
    qtype = ggjt.Q4_0
    scfg = dict(synth.MODELS["13B"])
    cfg = dict(n_vocab=scfg["n_vocab"], n_embd=scfg["n_embd"], n_mult=256, n_head=scfg["n_head"], n_layer=scfg["n_layer"])
    assert scfg["n_layer"] == 40 and ggjt.n_ff_of(cfg["n_embd"], cfg["n_mult"]) == scfg["n_ff"]
    gen = lambda: synth.synth_model_tensors(scfg, qtype, seed=4321)
    path = str(tmp_path_factory.mktemp("m13b") / "llama13b_q4_0.bin")
    ggjt.write_ggjt_stream(path, cfg, qtype, gen())
    text, toks = _prompt(512, 13)
    t0 = time.time()
    # (n_ctx 1024: the reference's session refuses a prompt that fills its context, lib/bridge.cpp:186-238.  extra_mem: its eval pool is
    #  1024 MiB + 20 MiB per batch token for a 13B model (include/model_type.hpp, lib/llama.cpp:173) -- the graph of a 512-token eval at
    #  this width needs ~13 GB, and the reference dereferences the NULL tensor it gets when the pool is full; allocate_extra_mem is its knob)
    ref = llama_capi.Session(reflib, path, n_ctx=1024, n_batch=512, n_threads=min(32, os.cpu_count() or 8), all_logits=True, extra_mem=24 << 30)
    assert ref.ingest(text) and ref.generate(1, temp=0.0)[0]
    want = ref.logits().reshape(512, cfg["n_vocab"]).copy()
    t_ref = time.time() - t0
    ref.close()
    os.remove(path)
    m = FlModel(scfg, qtype, gen(), n_ctx=512, max_batch=512)
    assert m.L.fl_model_get_exact(m.h) == 1                     # the library's default mode
    assert m.prepare(1) & 1                                     # the f16 operand copies (37 GB here) built outside the eval
    got = m.eval(toks, all_logits=True)
    m.free()
    torch.cuda.empty_cache()
    n_diff = int((bits(got) != bits(want)).sum())
    print(f"13B x 40 layers, n_batch 512: {n_diff} of {want.size} logits differ; reference {t_ref:.1f} s (load + eval)")
    assert n_diff == 0


def _run_shards(shards, fn):
    out, errs = [None] * len(shards), []

    def run(r):
        try:
            out[r] = fn(r)
        except Exception as e:           # a failing shard must not leave the others waiting forever
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(len(shards))]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs and all(not t.is_alive() for t in th), errs
    return out


def test_llama65b_width_nctx2048_deep_context_vs_reference(tmp_path_factory, reflib):
    """BASELINE config 5's width and context: 8 layers of n_embd 8192 / 64 heads / n_ff 22016, n_ctx 2048, n_batch 512."""
    import torch
    from fastllama_amd import hip
    from harness import synth
    from harness.flmodel import FlModel
    L = hip.load()
    qtype, NL, V, NCTX, G, STEPS = ggjt.Q4_0, 8, 32000, 2048, 8, 7
    w = synth.MODELS["65B"]
    scfg = dict(n_embd=w["n_embd"], n_head=w["n_head"], n_layer=NL, n_ff=w["n_ff"], n_vocab=V)
    cfg = dict(n_vocab=V, n_embd=w["n_embd"], n_mult=256, n_head=w["n_head"], n_layer=NL)
    assert ggjt.n_ff_of(cfg["n_embd"], cfg["n_mult"]) == scfg["n_ff"] and V % G == 0
    gen = lambda: synth.synth_model_tensors(scfg, qtype, seed=6565)
    path = str(tmp_path_factory.mktemp("m65w") / "w65_q4_0.bin")
    ggjt.write_ggjt_stream(path, cfg, qtype, gen())
    NP = 2040
    text, toks = _prompt(NP, 65)
    chunks = [(i, min(NP, i + 512)) for i in range(0, NP, 512)]            # 512 + 512 + 512 + 504 tokens
    # ---- the reference: llama_ingest evaluates all but the last n_batch, the first llama_generate step the last one -----------
    t0 = time.time()
    ref = llama_capi.Session(reflib, path, n_ctx=NCTX, n_batch=512, n_threads=min(32, os.cpu_count() or 8), all_logits=True, extra_mem=32 << 30)
    assert ref.ingest(text)
    want_steps, want_last_chunk = [], None
    for i in range(STEPS + 1):
        # greedy through top_k = 1 at temp 1: the reference's temp <= 0 shortcut returns the argmax's distance from the START of the logits
        # buffer (lib/bridge.cpp:38-41) -- with all_logits that is (rows - 1) * n_vocab + token, and the next eval reads tok_embeddings out of bounds
        ok, _ = ref.generate(1, top_k=1.0, temp=1.0)
        assert ok
        lg = ref.logits()
        if i == 0:                                                        # all_logits: the rows of the last prompt chunk (n_past 1536)
            want_last_chunk = lg.reshape(-1, V).copy()
            assert want_last_chunk.shape[0] == chunks[-1][1] - chunks[-1][0]
            want_steps.append(want_last_chunk[-1].copy())
        else:
            want_steps.append(lg.reshape(-1, V)[-1].copy())
    t_ref = time.time() - t0
    ref.close()
    os.remove(path)
    follow = [int(np.argmax(x)) for x in want_steps]                       # the reference's greedy tokens (temp 0)

    def run_model(evalfn):
        """the same evals through `evalfn(tokens, n_past, all_logits)`; returns (rows of the last chunk, per-step logits)"""
        last = None
        for a, b in chunks:
            last = evalfn(toks[a:b], a, True)
        steps = [last[-1]]
        for i in range(STEPS):
            steps.append(evalfn(np.array([follow[i]], np.int32), NP + i, False)[-1])
        return last, steps

    # ---- unsharded -----------------------------------------------------------------------------------------------------------
    tensors = list(gen())
    m = FlModel(scfg, qtype, tensors, n_ctx=NCTX, max_batch=512)
    assert m.L.fl_model_get_exact(m.h) == 1
    last, steps = run_model(lambda t, p, al: m.eval(t, n_past=p, all_logits=al))
    m.free()
    nd_chunk = int((bits(last) != bits(want_last_chunk)).sum())
    nd_steps = [int((bits(s) != bits(wnt)).sum()) for s, wnt in zip(steps, want_steps)]
    print(f"65B width x {NL} layers, n_ctx {NCTX}: last prompt chunk (n_past 1536, 504 rows) {nd_chunk} logits differ; decode steps at "
          f"positions {NP}..{NP + STEPS - 1}: {nd_steps[1:]} differ; reference {t_ref:.1f} s")
    assert nd_chunk == 0 and not any(nd_steps), (nd_chunk, nd_steps)
    # ---- G = 8 row-split shards on one device (the reference's split across threads, lib/ggml.c:8127-8135) ------------------------
    comms = (C.c_void_p * G)()
    hip.check(L.fl_comm_create_local(G, comms), "fl_comm_create_local")
    shards = []
    for r in range(G):
        s = FlModel(scfg, qtype, tensors, n_ctx=NCTX, max_batch=512, tp_rank=r, tp_size=G)
        s.set_comm(C.c_void_p(comms[r]))
        shards.append(s)
    del tensors
    last_g, steps_g = run_model(lambda t, p, al: _run_shards(shards, lambda r: shards[r].eval(t, n_past=p, all_logits=al))[G - 1])
    for s in shards:
        s.free()
    for r in range(G):
        L.fl_comm_destroy(C.c_void_p(comms[r]))
    torch.cuda.empty_cache()
    nd_chunk_g = int((bits(last_g) != bits(want_last_chunk)).sum())
    nd_steps_g = [int((bits(s) != bits(wnt)).sum()) for s, wnt in zip(steps_g, want_steps)]
    print(f"  as {G} row-split shards: last chunk {nd_chunk_g}, steps {nd_steps_g[1:]}")
    assert nd_chunk_g == 0 and not any(nd_steps_g), (nd_chunk_g, nd_steps_g)


def test_llama65b_width_q4_1_vs_reference(tmp_path_factory, reflib):
    """Q4_1 once at 65B width (2 layers, N = 128, real vocabulary): default mode bit-identical to the reference, prefill and three
    decode steps."""
    import torch
    from harness import synth
    from harness.flmodel import FlModel
    qtype, NL, V, N = ggjt.Q4_1, 2, 32000, 128
    w = synth.MODELS["65B"]
    scfg = dict(n_embd=w["n_embd"], n_head=w["n_head"], n_layer=NL, n_ff=w["n_ff"], n_vocab=V)
    cfg = dict(n_vocab=V, n_embd=w["n_embd"], n_mult=256, n_head=w["n_head"], n_layer=NL)
    gen = lambda: synth.synth_model_tensors(scfg, qtype, seed=4141)
    path = str(tmp_path_factory.mktemp("m65q41") / "w65_q4_1.bin")
    ggjt.write_ggjt_stream(path, cfg, qtype, gen())
    text, toks = _prompt(N, 41)
    ref = llama_capi.Session(reflib, path, n_ctx=256, n_batch=N, n_threads=min(32, os.cpu_count() or 8), all_logits=True)
    assert ref.ingest(text)
    want = []
    for i in range(4):
        assert ref.generate(1, top_k=1.0, temp=1.0)[0]        # (greedy: see the note on temp <= 0 above)
        want.append(ref.logits().reshape(-1, V).copy())
    ref.close()
    os.remove(path)
    m = FlModel(scfg, qtype, gen(), n_ctx=256, max_batch=N)
    got = m.eval(toks, all_logits=True)
    assert np.array_equal(bits(got), bits(want[0])), int((bits(got) != bits(want[0])).sum())
    n_past = N
    for i in range(1, 4):
        lg = m.eval([int(np.argmax(want[i - 1][-1]))], n_past=n_past)
        assert np.array_equal(bits(lg[-1]), bits(want[i][-1])), i
        n_past += 1
    m.free()
    torch.cuda.empty_cache()
