"""CPU: the numpy restatement of Model::eval (oracle/llama_eval.py) is pinned against the reference run
through its own C-ABI, and the tensor-parallel split reproduces the unsharded eval -- in-process and across
two real processes with the gloo backend (the N > 1 exchange path, checked without a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle
from oracle import llama_eval as le
from harness import ggjt, llama_capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXT = "The quick brown fox jumps over the lazy dog; 0123456789 times!?"


@pytest.fixture(scope="module")
def port():
    return oracle.Port()


def per_position_err(got, want):
    return np.max(np.abs(got.astype(np.float64) - want), axis=1) / np.max(np.abs(want))


@pytest.mark.parametrize("qtype", [ggjt.Q4_0, ggjt.Q4_1])
@pytest.mark.parametrize("cfgname,ntext", [("TINY", 40), ("SMALL", 63)])
def test_numpy_eval_matches_reference(tmp_path, port, qtype, cfgname, ntext):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    cfg = getattr(ggjt, cfgname)
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=4321)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qtype, tensors)
    text = TEXT[:ntext]
    toks = ggjt.text_tokens(text)
    lib = llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))
    ref = llama_capi.Session(lib, path, n_ctx=128, n_batch=128, all_logits=True)
    ref.perplexity(text)
    want = ref.logits().reshape(len(toks), cfg["n_vocab"])
    got, _ = le.eval_tokens(le.Weights(cfg, qtype, tensors), le.KV(cfg["n_layer"], 128, cfg["n_embd"]), toks, 0, port)
    # every op of the restatement follows the reference's build bit for bit, the f32 attention dots included
    # (ggml_vec_dot_f32's AVX2 lane order and its compiled leftover loop, oracle/q4_oracle.c:orc_vec_dot_f32_mm; these
    # models have head_dim 32 / 64 and 41..64 keys, so body, reduction and all three leftover forms are exercised)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), per_position_err(got, want)


def test_decode_steps_match_reference(tmp_path, port):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    cfg, qtype = ggjt.TINY, ggjt.Q4_0
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=99)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qtype, tensors)
    prompt = "abcdefghijklmnopqrstuvw"
    toks = ggjt.text_tokens(" " + prompt)
    lib = llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))
    ref = llama_capi.Session(lib, path, n_ctx=64, n_batch=8)
    assert ref.ingest(prompt)
    w, kv = le.Weights(cfg, qtype, tensors), le.KV(cfg["n_layer"], 64, cfg["n_embd"])
    n_past, lg = 0, None
    for i in range(0, len(toks), 8):
        lg, _ = le.eval_tokens(w, kv, toks[i:i + 8], n_past, port)
        n_past += len(toks[i:i + 8])
    errs = []
    for _ in range(3):
        ok, _ = ref.generate(1, temp=0.0)
        want = ref.logits()
        errs.append(float(np.max(np.abs(lg[-1] - want)) / np.max(np.abs(want))))
        assert np.array_equal(lg[-1].view(np.uint32), want.view(np.uint32))   # bit for bit, chunked ingest + decode steps
        tok = int(np.argmax(want))
        lg, _ = le.eval_tokens(w, kv, [tok], n_past, port)
        n_past += 1
    assert max(errs) <= 5e-2, errs


@pytest.mark.parametrize("cfgname,G", [("SMALL", 2), ("TINY", 4)])
def test_tensor_parallel_split_in_process(port, cfgname, G):
    """Shard exactly as fl_model_set_tensor does (rows of wq/wk/wv/w1/w3, K blocks of wo/w2), sum the partials."""
    cfg, qtype = getattr(ggjt, cfgname), ggjt.Q4_0
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=5)
    toks = ggjt.text_tokens(TEXT[:30])
    w = le.Weights(cfg, qtype, tensors)
    full, _ = le.eval_tokens(w, le.KV(cfg["n_layer"], 64, cfg["n_embd"]), toks, 0, port)
    # run the G ranks in lock-step: generators would be overkill -- evaluate layer by layer through a shared
    # "all-reduce" that needs every rank's partial, so drive the ranks as coroutines via threads
    import threading
    partials, results = {}, [None] * G
    barrier = threading.Barrier(G)

    def allreduce_for(rank):
        def ar(part):
            partials[rank] = part
            barrier.wait()
            tot = partials[0].copy()
            for r in range(1, G):
                tot = (tot + partials[r]).astype(np.float32)      # fixed rank order
            barrier.wait()
            return tot
        return ar

    def run(rank):
        kv = le.KV(cfg["n_layer"], 64, cfg["n_embd"] // G)
        results[rank], _ = le.eval_tokens(w, kv, toks, 0, oracle.Port(), tp_rank=rank, tp_size=G,
                                          allreduce=allreduce_for(rank))

    th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(1, G):
        assert np.array_equal(results[0], results[r])             # every rank ends with the same logits
    err = per_position_err(results[0], full)
    assert err[0] <= 1e-5 and err.max() <= 5e-2, err              # only the order of the partial sums differs


@pytest.mark.parametrize("cfgname,G,qtype", [("SMALL", 2, ggjt.Q4_0), ("TINY", 4, ggjt.Q4_1)])
def test_row_split_tensor_parallel_is_bit_identical_in_process(port, cfgname, G, qtype):
    """The reference-order mode's split: every matmul by output rows, all-gathers of the wo / w2 operands and output rows, nothing
    summed across ranks -- every rank ends with the UNSHARDED logits, bit for bit (prefill, then a decode step on the same caches)."""
    import threading
    cfg = getattr(ggjt, cfgname)
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=6)
    toks = ggjt.text_tokens(TEXT[:24])
    w = le.Weights(cfg, qtype, tensors)
    kvf = le.KV(cfg["n_layer"], 64, cfg["n_embd"])
    full, _ = le.eval_tokens(w, kvf, toks, 0, port)
    full1, _ = le.eval_tokens(w, kvf, toks[3:4], len(toks), port)
    parts, results, results1 = {}, [None] * G, [None] * G
    barrier = threading.Barrier(G)

    def allgather_for(rank):
        def ag(a):
            parts[rank] = a
            barrier.wait()
            out = np.concatenate([parts[r] for r in range(G)], axis=1)
            barrier.wait()
            return out
        return ag

    def run(rank):
        kv = le.KV(cfg["n_layer"], 64, cfg["n_embd"] // G)
        P = oracle.Port()
        results[rank], _ = le.eval_tokens(w, kv, toks, 0, P, tp_rank=rank, tp_size=G, allgather=allgather_for(rank))
        results1[rank], _ = le.eval_tokens(w, kv, toks[3:4], len(toks), P, tp_rank=rank, tp_size=G, allgather=allgather_for(rank))

    th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(G):
        assert np.array_equal(results[r].view(np.uint32), full.view(np.uint32)), r
        assert np.array_equal(results1[r].view(np.uint32), full1.view(np.uint32)), r


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port_no, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port_no}", rank=rank, world_size=world)
    P = oracle.Port()
    cfg, qtype = ggjt.TINY, ggjt.Q4_1
    tensors = ggjt.synth_tensors(cfg, qtype, P.quantize_q4, seed=11)
    toks = ggjt.text_tokens("tensor parallel over gloo")

    def allreduce(part):
        t = torch.from_numpy(np.ascontiguousarray(part))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    w = le.Weights(cfg, qtype, tensors)
    lg, _ = le.eval_tokens(w, le.KV(cfg["n_layer"], 64, cfg["n_embd"] // world), toks, 0, P, tp_rank=rank, tp_size=world,
                           allreduce=allreduce)
    if rank == 0:
        full, _ = le.eval_tokens(w, le.KV(cfg["n_layer"], 64, cfg["n_embd"]), toks, 0, P)
        np.savez(out_path, tp=lg, full=full)
    dist.barrier()
    dist.destroy_process_group()


def _gloo_rows_worker(rank, world, port_no, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port_no}", rank=rank, world_size=world)
    P = oracle.Port()
    cfg, qtype = ggjt.TINY, ggjt.Q4_0
    tensors = ggjt.synth_tensors(cfg, qtype, P.quantize_q4, seed=12)
    toks = ggjt.text_tokens("row split over gloo")

    def allgather(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs], axis=1)

    w = le.Weights(cfg, qtype, tensors)
    lg, _ = le.eval_tokens(w, le.KV(cfg["n_layer"], 64, cfg["n_embd"] // world), toks, 0, P, tp_rank=rank, tp_size=world,
                           allgather=allgather)
    if rank == 0:
        full, _ = le.eval_tokens(w, le.KV(cfg["n_layer"], 64, cfg["n_embd"]), toks, 0, P)
        np.savez(out_path, tp=lg, full=full)
    dist.barrier()
    dist.destroy_process_group()


def test_row_split_tensor_parallel_two_processes_gloo(tmp_path):
    """world_size 2 over torch.distributed/gloo, the collective pattern of the reference-order mode (4 all-gathers per layer:
    the wo / w2 operands and their output rows): the sharded logits ARE the unsharded ones."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "tpr.npz")
    port_no = _free_port()
    mp.spawn(_gloo_rows_worker, args=(2, port_no, out), nprocs=2, join=True)
    d = np.load(out)
    assert np.array_equal(d["tp"].view(np.uint32), d["full"].view(np.uint32))


def test_tensor_parallel_two_processes_gloo(tmp_path):
    """world_size 2 over torch.distributed/gloo: the same collective call pattern (2 all-reduces per layer of
    the [N, n_embd] partial sums) the RCCL path issues on the GPUs."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "tp.npz")
    port_no = _free_port()
    mp.spawn(_gloo_worker, args=(2, port_no, out), nprocs=2, join=True)
    d = np.load(out)
    err = per_position_err(d["tp"], d["full"])
    assert err[0] <= 1e-5 and err.max() <= 5e-2, err


def test_reference_logits_depend_on_batch_split(tmp_path, port):
    """Characterisation of the parity floor: the REFERENCE evaluated on the same prompt gives
    different logits for the same position when the batch is longer, because ggml_vec_dot_f32 over the (zero
    padded) soft_max row switches between its 32-wide AVX body and its scalar tail (lib/ggml.c:2295-2330): a
    1e-7 reordering that the fp16 soft_max table and the Q8_0 re-quantization amplify.  Any re-implementation that
    reorders f32 sums (MFMA, wave reductions) sits on the same floor; only quantities before the first flip are
    bit-comparable."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    cfg, qtype = dict(ggjt.SMALL), ggjt.Q4_0
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=1)   # a seed where a flip happens (most of 1..8 do not)
    path = str(tmp_path / "m.bin")
    ggjt.write_ggjt(path, cfg, qtype, tensors)
    lib = llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))
    ref = llama_capi.Session(lib, path, n_ctx=128, n_batch=128, all_logits=True)
    long_text = TEXT + " and then some more text follows here."
    ref.perplexity(long_text)
    full = ref.logits().reshape(-1, cfg["n_vocab"]).copy()
    ref.perplexity(long_text[:33])
    part = ref.logits().reshape(-1, cfg["n_vocab"])
    dev = per_position_err(part, full[:part.shape[0]])
    assert dev[0] == 0.0                 # position 0: one key, nothing to reorder
    assert 1e-4 < dev.max() < 5e-2, dev  # the same token, the same position, the same library: not within 1e-3
