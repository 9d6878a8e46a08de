"""GPU: the PRIMARY boundary.  The same ctypes driver (harness/llama_capi.py, a mirror of the reference's
interfaces/python/fastllama.py struct layouts) runs the reference (oracle/_ref/pyfastllama.so) and
libfastllama_hip.so on the same synthetic GGJT file: perplexity, logits, embeddings, ingest + greedy generate
(token stream), stop words, reset, save/load state (files are byte-compatible both ways)."""
import os

import numpy as np
import pytest

import oracle
from harness import ggjt, llama_capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "fastllama_amd", "libfastllama_hip.so")
TEXT = "The quick brown fox jumps over the lazy dog; 0123456789 times!?"


@pytest.fixture(scope="module")
def libs():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    return llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so")), llama_capi.LlamaLib(OURS)


@pytest.fixture(scope="module")
def model_file(tmp_path_factory):
    port = oracle.Port()
    cfg, qtype = ggjt.TINY, ggjt.Q4_0
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=77)
    path = str(tmp_path_factory.mktemp("api") / "tiny.bin")
    ggjt.write_ggjt(path, cfg, qtype, tensors)
    return path, cfg


def rel(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b)))


def test_all_17_symbols_exported(libs):
    ours = libs[1].lib
    for n in ("llama_create_default_context_args llama_create_context llama_load_model llama_set_stop_words "
              "llama_ingest_system_prompt llama_ingest llama_generate llama_perplexity llama_get_embeddings "
              "llama_get_logits llama_save_state llama_load_state llama_attach_lora llama_detach_lora "
              "llama_reset_model llama_free_context llama_handle_signal").split():
        assert hasattr(ours, n), n


def test_default_args_match_reference(libs):
    a, b = libs[0].lib.llama_create_default_context_args(), libs[1].lib.llama_create_default_context_args()
    for f in ("embedding_eval_enabled should_get_all_logits use_mmap use_mlock load_parallel seed n_keep n_ctx n_threads "
              "n_batch n_load_parallel_blocks last_n_tokens allocate_extra_mem").split():
        assert getattr(a, f) == getattr(b, f), f


def test_perplexity_logits_embeddings(libs, model_file):
    path, cfg = model_file
    out = []
    for lib in libs:
        s = llama_capi.Session(lib, path, n_ctx=128, n_batch=32, all_logits=True, embeddings=True)
        ppl = s.perplexity(TEXT)                       # 64 tokens -> two n_batch blocks, each eval at n_past = 0
        out.append((ppl, s.logits().reshape(-1, cfg["n_vocab"]), s.embeddings()))
        s.close()
    (p0, l0, e0), (p1, l1, e1) = out
    assert l0.shape == l1.shape == (32, cfg["n_vocab"])      # logits of the LAST block, all positions
    assert rel(l1[0], l0[0]) <= 1e-5 and rel(l1, l0) <= 5e-2
    assert abs(p1 - p0) / p0 <= 3e-2
    assert e0.shape == e1.shape == (cfg["n_embd"],) and rel(e1, e0) <= 5e-2


def test_ingest_then_greedy_generate_streams_same_tokens(libs, model_file):
    path, cfg = model_file
    res = []
    for lib in libs:
        s = llama_capi.Session(lib, path, n_ctx=96, n_batch=8)
        assert s.ingest("Once upon a time", system=True) is True
        assert s.ingest(" there was a GPU")
        ok, text = s.generate(12, temp=0.0)
        assert ok
        res.append((text, s.logits()))
        s.close()
    assert res[0][1].shape == res[1][1].shape == (cfg["n_vocab"],)
    # greedy streams agree token for token unless a rounding flip hits a near-tie; require a long common prefix
    a, b = res[0][0], res[1][0]
    common = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
    assert common >= 6, (a, b)


@pytest.mark.parametrize("with_system", [False, True])
def test_context_recycling_follows_the_reference(libs, model_file, with_system):
    """Generating past n_ctx makes FastLlama::recycle_embed_if_exceeds_context (lib/bridge.cpp:161-180) restart at
    n_past = n_keep with the system prompt and half of the recent tokens re-staged.  The greedy streams of the reference
    and of this library must still agree well beyond the recycle point (a wrong restart position or a wrong re-staged
    window changes every token after it)."""
    path, cfg = model_file
    res = []
    for lib in libs:
        s = llama_capi.Session(lib, path, n_ctx=48, n_batch=8, n_keep=8, last_n_tokens=24)
        if with_system:
            assert s.ingest("Sys", system=True)
        assert s.ingest(" abcdefghijklmnopqrstuvwxyz")
        ok, text = s.generate(40, temp=0.0)             # 1 + 28 (+ 5) prompt tokens: the context overflows after ~12-18 tokens
        assert ok
        res.append(text)
        s.close()
    a, b = res
    assert len(a) >= 36 and len(b) >= 36, (a, b)
    common = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
    assert common >= 30, (common, a, b)


def test_sampling_is_seeded_and_reset_restores_it(libs, model_file):
    path, cfg = model_file
    ours = libs[1]
    s = llama_capi.Session(ours, path, n_ctx=64, n_batch=16, seed=123)
    assert s.ingest("hello")
    ok, t1 = s.generate(8, top_k=40, top_p=0.9, temp=0.8, repeat_penalty=1.1)
    assert ok and s.reset()
    assert s.ingest("hello")
    ok, t2 = s.generate(8, top_k=40, top_p=0.9, temp=0.8, repeat_penalty=1.1)
    assert ok and t1 == t2                              # reset() re-seeds the mt19937 (lib/bridge.cpp:546)
    s.close()


def test_stop_words_cut_the_stream(libs, model_file):
    path, cfg = model_file
    ours = libs[1]
    s = llama_capi.Session(ours, path, n_ctx=64, n_batch=16)
    assert s.ingest("abc")
    ok, full = s.generate(10, temp=0.0)
    assert ok and len(full) >= 4
    stop = full[2:4].decode("latin1")
    s.reset()
    assert s.ingest("abc")
    ok, cut = s.generate(10, temp=0.0, stop_words=[stop])
    assert ok and cut == full[:2]
    s.close()


def test_state_files_are_interchangeable_with_the_reference(libs, model_file, tmp_path):
    path, cfg = model_file
    ref, ours = libs
    a = llama_capi.Session(ref, path, n_ctx=64, n_batch=8)
    assert a.ingest("state of the art")
    ok, _ = a.generate(3, temp=0.0)
    st = str(tmp_path / "ref.state")
    assert a.save_state(st)
    ok, cont_ref = a.generate(5, temp=0.0)
    b = llama_capi.Session(ours, path, n_ctx=64, n_batch=8)
    assert b.load_state(st)                              # a file written by the reference
    st2 = str(tmp_path / "ours.state")
    assert b.save_state(st2)
    assert open(st2, "rb").read() == open(st, "rb").read()   # load -> save reproduces the reference's file byte for byte
    ok, cont_ours = b.generate(5, temp=0.0)
    assert ok and cont_ours[:2] == cont_ref[:2]
    c = llama_capi.Session(ref, path, n_ctx=64, n_batch=8)
    assert c.load_state(st2)                             # and the reference reads ours
    a.close(); b.close(); c.close()


def test_error_paths(libs, tmp_path):
    ours = libs[1]
    L = ours.lib
    args = L.llama_create_default_context_args()
    ctx = L.llama_create_context(args)
    assert ctx
    assert L.llama_perplexity(ctx, b"x") == -1.0         # model not loaded
    assert not L.llama_ingest(ctx, b"x")
    assert L.llama_get_logits(ctx).size == 0
    assert not L.llama_load_model(ctx, os.fsencode(str(tmp_path / "missing.bin")))
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"notamodel" * 10)
    assert not L.llama_load_model(ctx, os.fsencode(str(bad)))
    L.llama_free_context(ctx)


def test_malformed_model_files_are_refused(libs, model_file, tmp_path):
    """Truncations at every structural boundary and corrupted directory fields: llama_load_model returns false (and says
    why through the logger) instead of reading outside the mapping; the intact file still loads afterwards."""
    path, cfg = model_file
    ours = libs[1]
    blob = open(path, "rb").read()
    cases = {"cut%d" % n: blob[:n] for n in (0, 3, 8, 20, 36, 40, 1000, len(blob) // 2, len(blob) - 1)}
    hdr = 8 + 7 * 4
    off = hdr
    for _ in range(cfg["n_vocab"]):                      # walk the vocabulary to the first tensor record
        ln = int.from_bytes(blob[off:off + 4], "little")
        off += 4 + ln + 4
    rec = bytearray(blob)
    rec[off + 12:off + 16] = (0).to_bytes(4, "little")   # ne[0] = 0
    cases["zero_dim"] = bytes(rec)
    rec = bytearray(blob)
    rec[off + 4:off + 8] = (0x7FFFFFFF).to_bytes(4, "little")   # name_len far beyond the file
    cases["name_len"] = bytes(rec)
    rec = bytearray(blob)
    rec[off:off + 4] = (9).to_bytes(4, "little")         # n_dims = 9
    cases["n_dims"] = bytes(rec)
    rec = bytearray(blob)
    rec[off + 8:off + 12] = (77).to_bytes(4, "little")   # unknown ggml type
    cases["type"] = bytes(rec)
    rec = bytearray(blob)
    rec[8:12] = (0x7FFFFFFF).to_bytes(4, "little")       # n_vocab
    cases["n_vocab"] = bytes(rec)
    L = ours.lib
    for name, data in cases.items():
        f = tmp_path / (name + ".bin")
        f.write_bytes(data)
        s_args = L.llama_create_default_context_args()
        s_args.n_ctx, s_args.n_batch = 32, 8
        msgs = []
        cbs = [llama_capi.LOG_FN(lambda f_, fl, m, ml: msgs.append(m[:ml])) for _ in range(3)]
        cbs += [llama_capi.RESET_FN(lambda: None), llama_capi.PROGRESS_FN(lambda t, d, tot: None)]
        s_args.logger = llama_capi.Logger(*cbs)
        ctx = L.llama_create_context(s_args)
        assert ctx
        assert not L.llama_load_model(ctx, os.fsencode(str(f))), name
        assert msgs, name                                # ... and the logger was told why
        L.llama_free_context(ctx)
    s = llama_capi.Session(ours, path, n_ctx=32, n_batch=8)
    assert s.ingest("ok")
    s.close()


def _ref_python_binding():
    """The reference's own interfaces/python/fastllama.py: from the reference tree when it is present (this container), else
    its bytecode oracle/_ref/fastllama.pyc (compiled from where it lies by oracle/Makefile; the GPU box has no /root/reference)."""
    import importlib.machinery
    import importlib.util
    src = "/root/reference/interfaces/python/fastllama.py"
    pyc = os.path.join(oracle.REF_DIR, "fastllama.pyc")
    if os.path.isfile(src):
        spec = importlib.util.spec_from_file_location("ref_fastllama", src)
    elif os.path.isfile(pyc):
        spec = importlib.util.spec_from_loader("ref_fastllama", importlib.machinery.SourcelessFileLoader("ref_fastllama", pyc))
    else:
        pytest.skip("neither the reference tree nor oracle/_ref/fastllama.pyc is present")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.filterwarnings("ignore::pytest.PytestUnraisableExceptionWarning")
def test_unmodified_reference_python_binding_runs_on_this_library(libs, model_file):
    """fastllama.Model(path, library_path=libfastllama_hip.so) -- the reference's Python surface, unmodified -- loads, ingests,
    generates and computes perplexity / logits ON THE GPU, with the values the plain ctypes driver gets from the same
    library and the same token stream the reference library produces.

    The filtered warning: the synthetic vocabulary is byte-level and the weights are random, so the greedy continuation contains
    lone UTF-8 continuation bytes (0x80..0xBF).  The reference's utf8_len (include/tokenizer.hpp:15-20) counts such a byte as a
    complete 1-byte character, so its TokenBuffer streams it (only a lead byte with missing continuation bytes is held back,
    include/token_buffer.hpp:118-130) and the binding's callback (interfaces/python/fastllama.py:365) fails to decode it -- with
    EITHER library behind it: the byte streams of the two libraries are compared raw, byte for byte, below."""
    import signal
    ref_py = _ref_python_binding()
    path, cfg = model_file
    old = signal.getsignal(signal.SIGINT)
    try:
        m = ref_py.Model(path, num_threads=4, n_ctx=128, n_batch=32, should_get_all_logits=True, library_path=OURS)
        ppl = m.perplexity(TEXT)
        lg = np.array(m.get_logits(), dtype=np.float32)
        s = llama_capi.Session(libs[1], path, n_ctx=128, n_batch=32, all_logits=True)
        assert ppl == pytest.approx(s.perplexity(TEXT), rel=1e-6)
        assert np.array_equal(lg, s.logits())
        s.close()
        m.reset()
        assert m.ingest("The quick brown fox")
        out = []
        assert m.generate(num_tokens=12, temp=0.0, streaming_fn=lambda t: out.append(t))
        rs = llama_capi.Session(libs[1], path, n_ctx=128, n_batch=32)          # the same library through the plain ctypes driver
        assert rs.ingest("The quick brown fox")
        ok, stream = rs.generate(12, temp=0.0)
        printable = lambda t: "".join(ch for ch in t if 32 <= ord(ch) < 127)       # (the binding drops invalid UTF-8 bytes its own way)
        assert ok and len(out) > 0 and printable("".join(out)) == printable(stream.decode("utf-8", "replace"))
        rr = llama_capi.Session(libs[0], path, n_ctx=128, n_batch=32)          # the REFERENCE library, same driver: the same raw bytes,
        assert rr.ingest("The quick brown fox")                                 # invalid UTF-8 included (nothing is held back that it streams)
        ok_r, stream_r = rr.generate(12, temp=0.0)
        assert ok_r and stream_r == stream
        rr.close()
        rs.close()
    finally:
        signal.signal(signal.SIGINT, old)


@pytest.mark.parametrize("example", ["perplexity", "example"])
def test_reference_c_examples_run_on_the_gpu(libs, tmp_path, example):
    """The reference's examples/c programs, compiled unmodified against include/fastllama.h and linked against
    libfastllama_hip.so (oracle/Makefile `clients`), run to completion on the GPU with a model at the path they hard-code."""
    import subprocess
    exe = os.path.join(oracle.REF_DIR, "ex_" + example)
    if not os.path.isfile(exe):
        pytest.skip("oracle/_ref/ex_* not built")
    port = oracle.Port()
    cfg, qtype = ggjt.TINY, ggjt.Q4_0
    os.makedirs(tmp_path / "models" / "7B")
    path = str(tmp_path / "models" / "7B" / "ggml-model-q4_0.bin")
    ggjt.write_ggjt(path, cfg, qtype, ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=77))
    text = (TEXT + " ") * 12
    (tmp_path / "test.txt").write_text(text)
    run = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    if example == "example":
        # its 450-character system prompt is 450 tokens in this byte-level vocabulary, more than the n_keep = 200 it sets:
        # the program must stop at ITS `return 2` with the reference's own refusal (lib/bridge.cpp:205-209), after a
        # successful load on the GPU
        assert run.returncode == 2 and "Ingesting, please wait" in run.stdout and "exceeds 'n_keep'" in run.stdout + run.stderr, \
            (run.returncode, run.stdout[-400:], run.stderr[-400:])
        return
    assert run.returncode == 0, (run.returncode, run.stdout[-500:], run.stderr[-500:])
    if example == "perplexity":
        got = float(run.stdout.rsplit("Total Perplexity:", 1)[1].split()[0])
        s = llama_capi.Session(libs[1], path, n_ctx=512, n_batch=512, n_threads=16, n_keep=200)
        assert got == pytest.approx(s.perplexity(text), rel=1e-4)
        s.close()


@pytest.mark.parametrize("container", ["ggmf", "ggml"])
def test_older_containers_load_to_the_same_model(libs, tmp_path_factory, container):
    """GGMF v1 (no tensor alignment) and the unversioned GGML container (no version word, no vocabulary scores) --
    include/file_loader.hpp:94-250 -- load through llama_load_model to the model the GGJT file gives: identical logits
    from this library, and the reference itself reads the same files to ITS GGJT logits (so the writer is right)."""
    port = oracle.Port()
    cfg, qtype = ggjt.TINY, ggjt.Q4_1
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=31)
    d = tmp_path_factory.mktemp("cont")
    paths = {c: str(d / f"tiny_{c}.bin") for c in ("ggjt", container)}
    for c, pth in paths.items():
        ggjt.write_ggjt(pth, cfg, qtype, tensors, container=c)
    assert os.path.getsize(paths[container]) < os.path.getsize(paths["ggjt"])          # no alignment padding (, no scores)
    for lib in libs:
        out = {}
        for c, pth in paths.items():
            s = llama_capi.Session(lib, pth, n_ctx=64, n_batch=32, all_logits=True)
            assert s.perplexity(TEXT[:30]) > 0
            out[c] = s.logits()
            s.close()
        assert out["ggjt"].size == 31 * cfg["n_vocab"] and np.array_equal(out[container], out["ggjt"])


def test_vocabulary_size_not_a_multiple_of_four(tmp_path_factory):
    """n_vocab = 323 (e.g. the 32001-token vocabularies of Alpaca / Vicuna style checkpoints): the lm-head GEMM writes rows
    of 16-byte-aligned stride; prefill (N >= 9), small batches and decode all return n_vocab dense logits that agree
    with the oracle eval."""
    from harness.flmodel import FlModel
    from oracle import llama_eval as le
    port = oracle.Port()
    cfg = dict(ggjt.TINY, n_vocab=323)
    qtype = ggjt.Q4_0
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=5)
    toks = ggjt.text_tokens(TEXT[:20])
    want, _ = le.eval_tokens(le.Weights(cfg, qtype, tensors), le.KV(cfg["n_layer"], 64, cfg["n_embd"]), toks, 0, port)
    m = FlModel(cfg, qtype, tensors, n_ctx=64, max_batch=32)
    got = m.eval(toks, all_logits=True)                       # MFMA GEMM path
    assert got.shape == (len(toks), 323) and rel(got, want.astype(np.float64)) <= 5e-2
    assert rel(got[0], want[0].astype(np.float64)) <= 1e-5
    last = m.eval(toks, all_logits=False)                     # last row only
    assert np.array_equal(last[0], got[-1])
    small = m.eval(toks[:5], all_logits=True)                 # N <= 8: GEMV path
    assert rel(small, want[:5].astype(np.float64)) <= 5e-2
    one = m.eval([toks[5]], n_past=5)                         # decode hipGraph
    assert np.isfinite(one).all() and one.shape == (1, 323)
    m.free()


@pytest.mark.parametrize("n_parts", [2, 4])
def test_multi_part_checkpoint_merges_to_the_same_model(tmp_path_factory, n_parts):
    """<path>, <path>.1, ... (tok_embeddings / wo / w2 split by columns, the other matrices by rows) load to exactly the
    model of the single file: same logits bit for bit (file_loader.hpp:377-453, tensor/utils.hpp:93-112)."""
    port = oracle.Port()
    cfg, qtype = ggjt.SMALL, ggjt.Q4_1
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=77)
    d = tmp_path_factory.mktemp("parts")
    one, many = str(d / "one.bin"), str(d / "many.bin")
    ggjt.write_ggjt(one, cfg, qtype, tensors)
    ggjt.write_ggjt_parts(many, cfg, qtype, tensors, n_parts)
    lib = llama_capi.LlamaLib(OURS)
    text = "multi part checkpoints merge on load"
    a = llama_capi.Session(lib, one, n_ctx=64, n_batch=64, all_logits=True)
    b = llama_capi.Session(lib, many, n_ctx=64, n_batch=64, all_logits=True)
    assert a.perplexity(text) == b.perplexity(text)
    assert np.array_equal(a.logits(), b.logits())
    a.close(); b.close()
    os.remove(many + ".1")
    with pytest.raises(RuntimeError):
        llama_capi.Session(lib, many, n_ctx=64, n_batch=64)


def test_perplexity_then_generate_samples_from_current_logits(libs, model_file):
    """llama_perplexity keeps its block's logits in HBM; a llama_generate that follows with nothing staged samples from the last
    row of those logits, exactly as the reference does with its m_logits (lib/bridge.cpp:262-266) -- the host copy must be
    refreshed first, also when perplexity was the very first call after load.  (Sampled with temp > 0 and top_k = 1, i.e. the
    arg-max of the last row through the sampling branch: the reference's temp <= 0 branch returns the arg-max's offset from the
    START of m_logits, lib/bridge.cpp:39-42, which is not a token id once m_logits holds more than one row.)"""
    path, cfg = model_file
    first = []
    for lib in libs:
        s = llama_capi.Session(lib, path, n_ctx=128, n_batch=32)
        assert s.perplexity(TEXT[:30]) > 0
        last_row = s.logits().reshape(-1, cfg["n_vocab"])[-1]
        ok, text = s.generate(1, top_k=1, top_p=1.0, temp=0.8)
        assert ok
        first.append((text, int(np.argmax(last_row))))
        s.close()
    assert first[0] == first[1] and len(first[1][0]) > 0, first   # the same first token from both libraries: the last row's arg-max


def test_exact_mode_through_the_primary_boundary(libs, model_file, tmp_path, monkeypatch):
    """FL_EXACT=1 (the mode that meets north_star's tolerance, read when the model is created): through the 17-symbol boundary the
    library returns the reference's BITS -- all-logits perplexity, a greedy stream with n_batch-8 ingest and context recycling,
    and a session state file (n_past, RNG, token windows, logits, the whole f32 K/V cache) that is byte-identical to the file
    the reference itself writes after the same calls."""
    path, cfg = model_file
    ref, ours = libs
    monkeypatch.setenv("FL_EXACT", "1")
    out = []
    for lib in libs:
        s = llama_capi.Session(lib, path, n_ctx=128, n_batch=32, all_logits=True, embeddings=True)
        ppl = s.perplexity(TEXT)
        out.append((ppl, s.logits(), s.embeddings()))
        s.close()
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert np.array_equal(out[0][2].view(np.uint32), out[1][2].view(np.uint32))
    assert abs(out[0][0] - out[1][0]) / out[0][0] <= 1e-5     # (the row softmax is summed in f64 on the device: DESIGN.md)
    res = []
    for i, lib in enumerate(libs):
        s = llama_capi.Session(lib, path, n_ctx=48, n_batch=8, n_keep=8, last_n_tokens=24)
        assert s.ingest("Sys", system=True) and s.ingest(" abcdefghijklmnopqrstuvwxyz")
        ok, text = s.generate(40, temp=0.0)                   # overflows the context: recycling included
        st = str(tmp_path / f"s{i}.state")
        assert ok and s.save_state(st)
        res.append((text, s.logits(), open(st, "rb").read()))
        s.close()
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))

    def masked(blob):                                         # mem_per_token = ggml_used_mem(ctx0) / N of the reference's first eval:
        n = int.from_bytes(blob[4:12], "little")              # an allocator statistic of the CPU graph, not a result (we store 1)
        off = 4 + 8 + n
        return blob[:off] + b"\0" * 8 + blob[off + 8:]
    assert masked(res[0][2]) == masked(res[1][2])             # n_past, RNG, token windows, logits, K/V cache: byte-identical
