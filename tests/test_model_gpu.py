"""GPU: the device-resident eval (fl_model_eval) against the reference's Model::eval, end to end.

The reference is run LIVE through its own C-ABI (oracle/_ref/pyfastllama.so ships with the snapshot) on a
synthetic GGJT model written by harness/ggjt.py; the same tensors are fed to fl_model.  What is asserted, and why the
north star's 1e-3 on logits holds only up to the first rounding flip, is in check_logits below; the configuration the
headline is quoted on (7B, 32 layers, n_batch 512) and the flip accounting live in tests/test_parity_7b_gpu.py.
"""
import os

import numpy as np
import pytest

import oracle
from harness import ggjt, llama_capi

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fast_mode")]   # (conftest.py: the fast kernels, explicitly)


def check_logits(got, want, what=""):
    """Parity criterion for whole-model logits on the deliberately TINY synthetic models.

    The reference path is discontinuous in two places: activations are re-quantized to int8 before every
    matmul, and soft_max rounds (score - max) to fp16 to index an exp table (lib/ggml.c:8569-8572).  An f32
    summation-ORDER difference of 1e-7 upstream (our MFMA / wave reductions vs the AVX2 lane order) therefore
    occasionally flips one rounding; with a few hundred channels and a dozen keys nothing averages it out, and
    every later position that attends to the affected token moves by ~1e-2.  tests/test_llama_eval_oracle.py
    shows the CPU restatement of the reference behaves identically against the reference itself.  Hence:
    position 0 (single key, no attention freedom; two or three layers) must agree to round-off, every position is
    bounded.  tests/test_parity_7b_gpu.py measures the same thing stage by stage at 7B width (matmul stages 5e-7, a
    handful of one-quantum flips per 262144 quants) and end to end on the full 32-layer model."""
    per_pos = np.max(np.abs(got.astype(np.float64) - want), axis=1) / np.max(np.abs(want))
    assert per_pos[0] <= 1e-5, (what, per_pos[0])
    assert per_pos.max() <= 5e-2, (what, per_pos.max())
    rel_l2 = np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want)
    assert rel_l2 <= 2e-2, (what, rel_l2)


TEXT = "The quick brown fox jumps over the lazy dog; 0123456789 times!?"


def relerr(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b)))


@pytest.fixture(scope="module")
def port():
    return oracle.Port()


@pytest.fixture(scope="module")
def reflib():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    return llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))


def build(tmp_path_factory, port, cfg, qtype, tag):
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=4321)
    path = str(tmp_path_factory.mktemp("m") / f"{tag}.bin")
    ggjt.write_ggjt(path, cfg, qtype, tensors)
    return tensors, path


@pytest.mark.parametrize("qtype", [ggjt.Q4_0, ggjt.Q4_1])
@pytest.mark.parametrize("cfgname", ["TINY", "SMALL"])
def test_prefill_all_logits_match_reference(tmp_path_factory, port, reflib, qtype, cfgname):
    from harness.flmodel import FlModel
    cfg = getattr(ggjt, cfgname)
    tensors, path = build(tmp_path_factory, port, cfg, qtype, cfgname)
    text = TEXT[:40] if cfgname == "TINY" else TEXT
    toks = ggjt.text_tokens(text)
    ref = llama_capi.Session(reflib, path, n_ctx=128, n_batch=128, all_logits=True, embeddings=True)
    ppl = ref.perplexity(text)
    want = ref.logits().reshape(len(toks), cfg["n_vocab"])
    want_emb = ref.embeddings()
    assert np.isfinite(ppl) and np.isfinite(want).all()
    m = FlModel(cfg, qtype, tensors, n_ctx=128, max_batch=128)
    got, emb = m.eval(toks, n_past=0, all_logits=True, embeddings=True)   # N >= 9: MFMA path
    assert got.shape == want.shape
    check_logits(got, want, cfgname)
    assert relerr(emb, want_emb) <= 5e-2
    # perplexity computed from OUR logits with the reference's rule (second half of the window, bridge.cpp:398-409)
    N = len(toks)
    nll, cnt = 0.0, 0
    for j in range(N >> 1, N - 1):
        l = got[j].astype(np.float64)
        p = np.exp(l - l.max())
        p /= p.sum()
        nll -= np.log(p[toks[j + 1]])
        cnt += 1
    assert abs(np.exp(nll / cnt) - ppl) / ppl < 3e-2      # tiny random model: a single flip moves ppl by ~1 %


@pytest.mark.parametrize("qtype", [ggjt.Q4_0, ggjt.Q4_1])
def test_chunked_prefill_and_decode_match_reference(tmp_path_factory, port, reflib, qtype):
    """n_past > 0: the reference ingests in n_batch blocks (lazy, bridge.cpp:215-231) and then decodes token by
    token; we replay the same eval() calls: chunks of 8 (wave-dot GEMV path, N <= 8), then N = 1 steps."""
    from harness.flmodel import FlModel
    cfg = ggjt.TINY
    tensors, path = build(tmp_path_factory, port, cfg, qtype, "dec")
    prompt = "abcdefghijklmnopqrstuvw"                 # ingest() prepends a space (bridge.cpp:193)
    toks = ggjt.text_tokens(" " + prompt)
    ref = llama_capi.Session(reflib, path, n_ctx=64, n_batch=8, all_logits=False)
    assert ref.ingest(prompt)
    m = FlModel(cfg, qtype, tensors, n_ctx=64, max_batch=8)
    n_past, lg = 0, None
    for i in range(0, len(toks), 8):
        blk = toks[i:i + 8]
        lg = m.eval(blk, n_past=n_past)
        n_past += len(blk)
    greedy, errs = [], []
    for step in range(4):
        ok, _ = ref.generate(1, temp=0.0)              # evaluates the pending block, samples argmax (bridge.cpp:39-42)
        assert ok
        want = ref.logits()
        assert want.size == cfg["n_vocab"]
        errs.append(relerr(lg[0], want))
        tok = int(np.argmax(want))                     # follow the reference's token so both stay on one path
        greedy.append(tok)
        if tok == 2:
            break
        lg = m.eval([tok], n_past=n_past)              # decode step, N = 1
        n_past += 1
    assert max(errs) <= 5e-2, errs


def test_eval_argument_errors(port):
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    cfg = ggjt.TINY
    tensors = ggjt.synth_tensors(cfg, ggjt.Q4_0, port.quantize_q4)
    m = FlModel(cfg, ggjt.Q4_0, tensors, n_ctx=32, max_batch=16)
    with pytest.raises(hip.FastLlamaHipError):
        m.eval(list(range(3, 20)), n_past=0)           # N > max_batch
    with pytest.raises(hip.FastLlamaHipError):
        m.eval([5] * 8, n_past=30)                     # beyond n_ctx
    for bad in (-1, cfg["n_vocab"]):
        with pytest.raises(hip.FastLlamaHipError):
            m.eval([5, bad, 7], n_past=0)              # token outside the vocabulary
        with pytest.raises(hip.FastLlamaHipError):
            m.eval([bad], n_past=3)                    # ... on the single-token (hipGraph) path too
    assert np.isfinite(m.eval([5, 6, 7], n_past=0)).all()       # the model is still usable afterwards
    t2 = dict(tensors)
    del t2["layers.1.attention.wo.weight"]
    with pytest.raises(hip.FastLlamaHipError):
        FlModel(cfg, ggjt.Q4_0, t2, n_ctx=32, max_batch=16)   # finalize reports the missing tensor


def test_llama7b_width_logits(tmp_path_factory, port, reflib):
    """The real width: n_embd 4096, 32 heads, n_ff 11008, vocabulary 32000, Q4_0 (2 layers to bound the CPU
    reference's run time), 96-token prefill, both modes of the library:
      * exact mode (reference-order kernels): every logit equals the reference's bit for bit;
      * fast mode (MFMA kernels, f32 terms added in their own order): a 1e-7 difference either vanishes or flips one Q8_0 /
        fp16-table rounding, and a random net amplifies each flip -- a few 1e-2 of max|logit| after two layers.  (At this width
        the reference does NOT deviate from itself across batch splits: its f32 dots have no remainder loops,
        tests/test_parity_7b_gpu.py; it does at toy widths, tests/test_llama_eval_oracle.py.)  What a user observes agrees:
        perplexity within 0.5 %, greedy token at >= 90 % of positions."""
    from harness.flmodel import FlModel
    cfg = dict(n_vocab=32000, n_embd=4096, n_mult=256, n_head=32, n_layer=2)
    qtype = ggjt.Q4_0
    tensors, path = build(tmp_path_factory, port, cfg, qtype, "w7b")
    rng = np.random.default_rng(3)
    text = bytes(rng.integers(32, 127, size=95).astype(np.uint8)).decode()
    toks = ggjt.text_tokens(text)
    ref = llama_capi.Session(reflib, path, n_ctx=128, n_batch=128, n_threads=16, all_logits=True)
    ppl = ref.perplexity(text)
    want = ref.logits().reshape(len(toks), cfg["n_vocab"])
    os.remove(path)
    m = FlModel(cfg, qtype, tensors, n_ctx=128, max_batch=128)
    m.set_exact(True)
    got_x = m.eval(toks, n_past=0, all_logits=True)
    assert np.array_equal(got_x.view(np.uint32), np.ascontiguousarray(want, np.float32).view(np.uint32))
    m.set_exact(False)
    got = m.eval(toks, n_past=0, all_logits=True)
    per_pos = np.max(np.abs(got.astype(np.float64) - want), axis=1) / np.max(np.abs(want))
    assert per_pos.max() <= 5e-2, per_pos.max()
    assert np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want) <= 2e-2
    assert np.mean(np.argmax(got, axis=1) == np.argmax(want, axis=1)) >= 0.9
    N = len(toks)
    nll = 0.0
    for j in range(N >> 1, N - 1):
        l = got[j].astype(np.float64)
        nll -= (l[toks[j + 1]] - l.max()) - np.log(np.exp(l - l.max()).sum())
    assert abs(np.exp(nll / (N - 1 - (N >> 1))) - ppl) / ppl < 5e-3


def test_rccl_world_of_one_allreduce():
    """the RCCL wrapper itself on the GPU box: id, communicator, in-place f32 sum over a world of one"""
    import ctypes as C
    import torch
    from fastllama_amd import hip
    L = hip.load()
    hip.require_device(0)
    idb = (C.c_ubyte * 128)()
    hip.check(L.fl_comm_unique_id(idb), "fl_comm_unique_id")
    c = L.fl_comm_create(idb, 0, 1)
    assert c, L.fl_last_error().decode()
    c = C.c_void_p(c)
    assert L.fl_comm_rank(c) == 0 and L.fl_comm_size(c) == 1
    x = torch.randn(4096 * 7, device="cuda")
    want = x.clone()
    hip.check(L.fl_comm_allreduce_sum_f32(c, x.data_ptr(), x.numel(), None), "allreduce")
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    L.fl_comm_destroy(c)


def test_rccl_allreduce_inside_a_hipgraph():
    """ncclAllReduce captured into a hipGraph and replayed (what the tensor-parallel decode graph does per layer), on the
    world of one a 1-GPU box offers: the capture / instantiate / replay path of RCCL itself."""
    import ctypes as C
    import torch
    from fastllama_amd import hip
    L = hip.load()
    hip.require_device(0)
    idb = (C.c_ubyte * 128)()
    hip.check(L.fl_comm_unique_id(idb), "fl_comm_unique_id")
    c = C.c_void_p(L.fl_comm_create(idb, 0, 1))
    assert c, L.fl_last_error().decode()
    x = torch.randn(4096, device="cuda")
    want = x.clone()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        hip.check(L.fl_comm_debug_graph_allreduce(c, x.data_ptr(), x.numel(), 3, C.c_void_p(st.cuda_stream)), "graph allreduce")
    torch.cuda.synchronize()
    assert torch.equal(x, want)                  # sum over one rank, three times
    L.fl_comm_destroy(c)


def test_peer_exchange_two_processes_on_one_gpu(tmp_path):
    """The one-shot small-message exchange (peer-mapped buffers over hipIpc, one kernel per rank: publish, wait, add in rank
    order) between two PROCESSES -- on one GPU, which is all this box has and which RCCL refuses; the handles travel through
    files.  All-reduces of several sizes (alternating slots), an all-gather, an all-reduce captured in a hipGraph and replayed:
    every rank ends with the rank-order f32 sum, bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "harness.p2p_worker", str(r), "2", str(tmp_path), "0"], cwd=root, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("peer exchange workers did not finish (a rank spinning on a peer that never ran?)")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    o = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(2)]
    for k, n in enumerate((4096, 1, 8192, 16384, 333, 4096)):
        xs = [np.random.default_rng(100 * k + r).standard_normal(n).astype(np.float32) for r in range(2)]
        want = (xs[0] + xs[1]).astype(np.float32)
        for r in range(2):
            assert np.array_equal(o[r][f"ar{k}"].view(np.uint32), want.view(np.uint32)), (k, r)
    want = np.concatenate([np.random.default_rng(7 + r).standard_normal(1000).astype(np.float32) for r in range(2)])
    for r in range(2):
        assert np.array_equal(o[r]["ag"], want)
    xs = [np.random.default_rng(55 + r).standard_normal(4096).astype(np.float32) for r in range(2)]
    s1 = (xs[0] + xs[1]).astype(np.float32)
    want = ((s1 + s1) + (s1 + s1)).astype(np.float32)            # three replays: S, 2S, 4S (exact doublings)
    for r in range(2):
        assert np.array_equal(o[r]["graph"].view(np.uint32), want.view(np.uint32)), r
    print("peer all-reduce of 4096 floats: %.1f / %.1f us per call" % (float(o[0]["us_per_allreduce"]), float(o[1]["us_per_allreduce"])))


def test_two_process_tensor_parallel_over_the_peer_exchange(tmp_path):
    """The tensor-parallel eval across two real PROCESSES on this box's one GPU: every collective of the (small) model --
    two all-reduces per layer, the logits all-gather, eager in prefill and captured in the decode hipGraph -- goes through the
    peer-mapped exchange.  Same checks as the two-GPU RCCL test below."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "harness.tp_worker", str(r), "2", str(tmp_path / "unused"), str(tmp_path / f"out{r}.npz"),
                               str(tmp_path)], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("tensor-parallel workers did not finish")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    o = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(2)]
    for k in ("pre", "dec_graph", "dec_graph2", "dec_plain"):
        assert np.array_equal(o[0][k], o[1][k]), k                      # every rank ends with the full, identical logits
    assert np.array_equal(o[0]["dec_graph2"], o[0]["dec_plain"])       # graph replay == plain launches
    check_logits(o[0]["pre"], o[0]["full_pre"].astype(np.float64), "tp2 (peer exchange) prefill vs unsharded")
    assert relerr(o[0]["dec_graph"], o[0]["full_dec"]) <= 5e-2


def _run_tp_workers(tmp_path, tag, extra_env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / tag
    d.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FL_EXACT="1", **(extra_env or {}))      # (this module's models are fast-mode ones: conftest.py)
    env.pop("FL_FAST", None)
    procs = [subprocess.Popen([sys.executable, "-m", "harness.tp_worker", str(r), "2", str(d / "unused"), str(d / f"out{r}.npz"), str(d)],
                              cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("tensor-parallel workers did not finish")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [np.load(str(d / f"out{r}.npz")) for r in range(2)]


def test_two_process_row_split_decode_with_the_exchanges_folded_into_the_producers(tmp_path):
    """VERDICT r4 item 5.  The default (reference-order) mode shards every matmul by output rows; its decode token used to run 11 kernels + 4
    collectives per layer.  Over a communicator with the peer-mapped exchange the four exchanges of a layer are now the TAILS of the launches
    that produce the data (tp_tail.h): five launches per layer (six with the two-launch attention of long contexts), counted from the captured
    hipGraph.  Two processes on this box's one GPU; every logit of the prefill and of a run of decode tokens -- graph replays, the split
    attention's graph, plain launches -- equals the UNSHARDED model's bit for bit on both ranks, and the collective sequence (FL_TP_FOLD=0)
    gives the same bits."""
    o = _run_tp_workers(tmp_path, "fold")
    L = int(o[0]["n_layer"])
    for r in range(2):
        assert int(o[r]["folded"]) == 1
        assert np.array_equal(o[r]["seq"].view(np.uint32), o[0]["full_seq"].view(np.uint32)), r
        assert np.array_equal(o[r]["pre"].view(np.uint32), o[0]["full_pre"].view(np.uint32)), r
        assert np.array_equal(o[r]["dec_graph"].view(np.uint32), o[0]["full_dec"].view(np.uint32)), r
        # get_rows + L x (wq|wk|wv, attention, wo, w1|w3, w2) + lm-head + the logits' gather (exchange, permute)
        assert int(o[r]["graph_nodes"]) == 5 * L + 4, int(o[r]["graph_nodes"])
        assert int(o[r]["graph_nodes_split"]) == 6 * L + 4, int(o[r]["graph_nodes_split"])
    assert int(o[0]["full_graph_nodes"]) == 5 * L + 2
    c = _run_tp_workers(tmp_path, "collectives", {"FL_TP_FOLD": "0"})
    for r in range(2):
        assert int(c[r]["folded"]) == 0
        assert np.array_equal(c[r]["seq"].view(np.uint32), o[0]["full_seq"].view(np.uint32)), r
        assert int(c[r]["graph_nodes"]) > 9 * L
    print("decode graph nodes per layer: folded %d (split attention %d), collective sequence %.1f; us per token (tiny model, two ranks on one GPU): "
          "folded %.0f / %.0f, collectives %.0f / %.0f" % ((int(o[0]["graph_nodes"]) - 4) // L, (int(o[0]["graph_nodes_split"]) - 4) // L,
          (int(c[0]["graph_nodes"]) - 4) / L, float(o[0]["us_per_token"]), float(o[1]["us_per_token"]), float(c[0]["us_per_token"]), float(c[1]["us_per_token"])))


def test_row_split_decode_fails_loudly_when_a_peer_never_arrives(tmp_path):
    """The folded exchange's failure mode: a rank whose peer never publishes its epoch.  Every wait is bounded (FL_P2P_TIMEOUT_MS here 1 s;
    20 s by default), each one given up is counted (after the first, the communicator's later exchanges do not wait again), and fl_model_eval
    returns an error for that token instead of hanging the GPU's queue."""
    o = _run_tp_workers(tmp_path, "dies", {"FL_TP_WORKER_MODE": "peer_never_arrives", "FL_P2P_TIMEOUT_MS": "1000"})
    assert int(o[0]["folded"]) == 1 and int(o[1]["folded"]) == 1
    assert "gave up waiting for a peer" in str(o[0]["error"]), str(o[0]["error"])
    assert float(o[0]["seconds"]) < 60, float(o[0]["seconds"])
    print("a decode token whose peer never arrives: error after %.1f s (bound per wait: 1 s; 13 exchanges in the token)" % float(o[0]["seconds"]))


def test_bench_tensor_parallel_leg_as_two_processes_on_one_gpu():
    """bench.py's world > 1 code -- replica leg, tensor-parallel leg (communicator, sharded model, timing, the JSON line with
    `replicas` beside the headline) -- run the way the driver launches it, with two ranks on this box's one GPU: gloo for the
    process group, the peer exchange as the communicator, a model whose messages fit it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FL_BENCH_DEVICE="0", FL_BENCH_BACKEND="gloo", FL_BENCH_P2P_ONLY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", "bench.py", "--gpus", "2", "--model", "tiny", "--n-batch", "32", "--steps", "3",
                        "--warmup", "1", "--decode-steps", "8"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "tp_error" not in d
    assert d["config"]["parallelism"].startswith("tp2") and d["tp_small_message_path"].startswith("peer-mapped")
    assert d["replicas"]["scaling"] == "weak" and d["replicas"]["prefill_tokens_per_s"] > 0
    assert d["value"] > 0 and d["decode_tokens_per_s"] > 0 and d["config"]["global_batch_tokens"] == 32
    # the headline leg is the default mode = the row split: its decode runs with the exchanges folded into the producing launches
    assert d["tp_decode"]["exchanges_folded_into_producers"] is True and d["tp_decode"]["launches_per_layer"] <= 7, d["tp_decode"]


def test_rccl_two_ranks_tensor_parallel(tmp_path):
    """Two processes, two GPUs, RCCL over xGMI: the Megatron split with the row-split lm-head + all-gather, prefill and decode
    (collectives captured in the decode hipGraph, then plain launches) against the unsharded model.  Needs two devices: the
    gpurun box has one (skipped there); `harness/tp_worker.py` is the same program for a multi-GPU node."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the tensor-parallel path is covered on one GPU by test_tensor_parallel_shards_on_one_gpu)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idf = str(tmp_path / "id.bin")
    procs = [subprocess.Popen([sys.executable, "-m", "harness.tp_worker", str(r), "2", idf, str(tmp_path / f"out{r}.npz")], cwd=root)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    o = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(2)]
    for k in ("pre", "dec_graph", "dec_graph2", "dec_plain"):
        assert np.array_equal(o[0][k], o[1][k]), k                      # every rank ends with the full, identical logits
    assert np.array_equal(o[0]["dec_graph2"], o[0]["dec_plain"])       # graph replay == plain launches
    check_logits(o[0]["pre"], o[0]["full_pre"].astype(np.float64), "tp2 (RCCL) prefill vs unsharded")
    assert relerr(o[0]["dec_graph"], o[0]["full_dec"]) <= 5e-2


@pytest.mark.parametrize("cfgname,G", [("SMALL", 2), ("TINY", 4)])
@pytest.mark.parametrize("qtype", [ggjt.Q4_0, ggjt.Q4_1])
def test_tensor_parallel_shards_on_one_gpu(tmp_path_factory, port, qtype, cfgname, G):
    """the Megatron split (wq/wk/wv/w1/w3 by rows, wo/w2 by K blocks, partial sums all-reduced) run as G
    logical shards on ONE device -- each shard its own fl_model and host thread, the collective being the
    single-process group of fl_comm_create_local.  Checked against the unsharded model on the same GPU (the numpy
    restatement of the same split is pinned to the reference on the CPU in tests/test_llama_eval_oracle.py)."""
    import ctypes as C
    import threading
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = getattr(ggjt, cfgname)
    tensors, _ = build(tmp_path_factory, port, cfg, qtype, f"tp{cfgname}")
    toks = ggjt.text_tokens(TEXT[:33])
    full = FlModel(cfg, qtype, tensors, n_ctx=64, max_batch=64)
    want = full.eval(toks, all_logits=True)
    want_dec = full.eval([toks[3]], n_past=len(toks))
    comms = (C.c_void_p * G)()
    hip.check(L.fl_comm_create_local(G, comms), "fl_comm_create_local")
    shards = []
    for r in range(G):
        m = FlModel(cfg, qtype, tensors, n_ctx=64, max_batch=64, tp_rank=r, tp_size=G)
        m.set_comm(C.c_void_p(comms[r]))
        shards.append(m)
    got, got_dec, errs = [None] * G, [None] * G, []

    def run(r):
        try:
            got[r] = shards[r].eval(toks, all_logits=True)
            got_dec[r] = shards[r].eval([toks[3]], n_past=len(toks))
        except Exception as e:           # a failing shard must not leave the others waiting forever
            errs.append(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs and all(not t.is_alive() for t in th), errs
    for r in range(1, G):                # every shard holds the full logits (lm-head is replicated), identical bits
        assert np.array_equal(got[r], got[0]) and np.array_equal(got_dec[r], got_dec[0])
    check_logits(got[0], want.astype(np.float64), f"tp{G} prefill vs unsharded")
    assert relerr(got_dec[0], want_dec) <= 5e-2
    for m in shards:
        m.free()
    for r in range(G):
        L.fl_comm_destroy(C.c_void_p(comms[r]))


@pytest.mark.parametrize("qtype", [ggjt.Q4_0])
def test_long_context_path_switches_agree(tmp_path_factory, port, qtype):
    """A 1400-token context evaluated in different chunkings walks through every attention path: the one-launch prefill
    kernel in pair mode (n_past = 0), in single-block mode (n_past > 0), the key-tiled form once n_past + N no
    longer fits LDS (> 960 keys), the N <= 8 path and the single-token decode kernel.  The KV cache they leave behind
    must be interchangeable: a fixed probe token evaluated after each chunking gives logits that agree to the usual
    cross-path bound, and the K/V state agrees bit for bit where the producing kernels are bit-identical by design."""
    import ctypes as C
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    tensors, _ = build(tmp_path_factory, port, cfg, qtype, "long")
    rng = np.random.default_rng(5)
    toks = rng.integers(3, 259, 1400).astype(np.int32)
    n_ctx = 1536
    E, Ln = cfg["n_embd"], cfg["n_layer"]

    def run(chunks):
        m = FlModel(cfg, qtype, tensors, n_ctx=n_ctx, max_batch=512)
        n_past = 0
        for c in chunks:
            m.eval(toks[n_past:n_past + c], n_past=n_past)
            n_past += c
        assert n_past == 1400
        k = np.empty((Ln, n_ctx, E), np.float32)
        v = np.empty((Ln, E, n_ctx), np.float32)
        hip.check(L.fl_model_kv_read(m.h, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)), "kv_read")
        lg = m.eval([int(toks[7])], n_past=1400)[0].copy()
        m.free()
        return k, v, lg

    ka, va, la = run([512, 448, 440])                    # pair mode; single-block mode (P = 960); key-tiled form
    kb, vb, lb = run([500, 300, 300, 300])               # odd block counts, other switch points
    kc_, vc_, lc = run([512, 512, 368, 8])               # ... and an N <= 8 tail
    # layer 0 K/V depend only on the token embeddings and the (bit-identical) wqkv + rope epilogue vs kernel pair
    assert np.array_equal(ka[0, :1400], kb[0, :1400]) and np.array_equal(va[0, :, :1400], vb[0, :, :1400])
    assert np.array_equal(ka[0, :1392], kc_[0, :1392])
    for other in (lb, lc):
        assert relerr(other, la) <= 5e-2
    # deeper layers: same values up to the rounding-flip noise of the algorithm
    d = np.abs(ka[:, :1400] - kb[:, :1400]).max() / np.abs(ka[:, :1400]).max()
    assert d <= 5e-2, d


@pytest.mark.gpu
def test_prefill_attention_forms_are_interchangeable(tmp_path_factory, port):
    """The three forms of prefill attention -- score rows in LDS, key-tiled through a scratch buffer (what deep contexts
    get), three kernels -- are bit-identical per output, so a 1400-token prompt ingested in 512-token chunks leaves the same
    K/V cache and the same logits whichever form runs: default (LDS form, then the key-tiled one from the second chunk
    on), key-tiled forced everywhere (set_graph bit 5), three kernels everywhere (bit 2)."""
    import ctypes as C
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    tensors, _ = build(tmp_path_factory, port, cfg, ggjt.Q4_0, "forms")
    toks = np.random.default_rng(6).integers(3, 259, 1400).astype(np.int32)
    n_ctx = 1536
    E, Ln = cfg["n_embd"], cfg["n_layer"]

    def run(mode):
        m = FlModel(cfg, ggjt.Q4_0, tensors, n_ctx=n_ctx, max_batch=512)
        hip.check(L.fl_model_set_graph(m.h, 1 | mode))
        n_past, last = 0, None
        for c in (512, 448, 431, 9):
            last = m.eval(toks[n_past:n_past + c], n_past=n_past, all_logits=True).copy()
            n_past += c
        k = np.empty((Ln, n_ctx, E), np.float32)
        v = np.empty((Ln, E, n_ctx), np.float32)
        hip.check(L.fl_model_kv_read(m.h, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)), "kv_read")
        m.free()
        return k[:, :1400], v[:, :, :1400], last

    k0, v0, l0 = run(0)
    for mode in (32, 4):
        k1, v1, l1 = run(mode)
        assert np.array_equal(k0.view(np.uint32), k1.view(np.uint32)), mode
        assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)), mode
        assert np.array_equal(l0.view(np.uint32), l1.view(np.uint32)), mode
    assert np.isfinite(l0).all() and np.ptp(l0) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,total", [(512, 1400), (100, 731), (37, 75), (512, 512), (64, 65)])
def test_pipelined_ingest_equals_chunk_by_chunk_evals(tmp_path_factory, port, chunk, total):
    """fl_model_ingest keeps two consecutive chunks of a prompt in flight on two streams (chunk c+1 waits for chunk c layer by
    layer, where its attention reads the K/V cache) and runs the lm-head for the last chunk only.  The evals are the same: K/V
    cache and final logits are bit-identical to fl_model_eval chunk by chunk, from n_past = 0 and behind an existing context,
    twice in a row (the second set of work buffers and its events are reused), and a decode step afterwards sees the same state."""
    import ctypes as C
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    tensors, _ = build(tmp_path_factory, port, cfg, ggjt.Q4_0, "pipe")
    toks = np.random.default_rng(chunk + total).integers(3, 259, total + 40).astype(np.int32)
    n_ctx = 1536
    E, Ln = cfg["n_embd"], cfg["n_layer"]

    def kv(m, upto):
        k = np.empty((Ln, n_ctx, E), np.float32)
        v = np.empty((Ln, E, n_ctx), np.float32)
        hip.check(L.fl_model_kv_read(m.h, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)), "kv_read")
        return k[:, :upto].copy(), v[:, :, :upto].copy()

    def run(pipelined, mode=1):
        m = FlModel(cfg, ggjt.Q4_0, tensors, n_ctx=n_ctx, max_batch=512)
        hip.check(L.fl_model_set_graph(m.h, mode))
        out = []
        past = 0
        for lo, hi in ((0, 40), (40, 40 + total)):                     # a short context first, then the long prompt behind it
            if pipelined:
                lg = m.ingest(toks[lo:hi], chunk, n_past=past)
            else:
                for i in range(lo, hi, chunk):
                    lg = m.eval(toks[i:min(i + chunk, hi)], n_past=past + i - lo)
            past += hi - lo
            out.append(lg.copy())
        out.append(m.eval(toks[:1], n_past=past).copy())                # decode step on top
        k, v = kv(m, past + 1)
        m.free()
        return out, k, v

    a, ka, va = run(False)
    for mode in (1, 1 | 128):                                           # two streams; the one-stream form wide / sharded models take
        b, kb, vb = run(True, mode)
        assert np.array_equal(ka.view(np.uint32), kb.view(np.uint32)) and np.array_equal(va.view(np.uint32), vb.view(np.uint32))
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.isfinite(a[-1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [0, 1])
def test_decode_attention_split_switch_is_invisible(tmp_path_factory, port, graph):
    """Decode switches to the two-launch attention at position 256 (second hipGraph capture).  Tokens decoded across the
    switch give bit-identical logits with the split forced on from the start, forced off, and in the default mode --
    as hipGraph replays and as plain launches."""
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    cfg = ggjt.SMALL
    tensors, _ = build(tmp_path_factory, port, cfg, 2, "split")
    toks = np.random.default_rng(11).integers(3, 259, 270).astype(np.int32)

    def run(mode):
        m = FlModel(cfg, 2, tensors, n_ctx=512, max_batch=256)
        hip.check(L.fl_model_set_graph(m.h, graph | mode))
        m.eval(toks[:250], n_past=0)
        out = [m.eval(toks[p:p + 1], n_past=p)[0].copy() for p in range(250, 270)]
        m.free()
        return np.stack(out)

    default, never, always = run(0), run(8), run(16)
    assert np.array_equal(default.view(np.uint32), never.view(np.uint32))
    assert np.array_equal(default.view(np.uint32), always.view(np.uint32))
    assert np.isfinite(default).all() and np.ptp(default) > 0
