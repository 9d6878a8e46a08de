#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: tokens/s of LLaMA-7B Q4_0, n_batch=512 prefill + decode.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE Model::eval of an n_batch=512 batch at n_past=0 on synthetic LLaMA-7B-shaped Q4_0 weights
(BASELINE.json configs[1]), entirely on the device: token-embedding dequant, 32 layers (rms_norm+Q8_0, fused
wq|wk|wv Q4xQ8 GEMM, rope + KV store, KQ / soft_max / KQV, wo, rms_norm+Q8_0, fused w1|w3 GEMM, silu*mul+Q8_0,
w2), final norm and the lm-head -- i.e. the hot path (225 mul_mat_q_f32 = 129 GEMM launches after fusion) plus
everything around it; nothing is skipped or cached.  Token ids and weights are resident in HBM, logits stay in HBM.
`value` = prefill tokens/s summed over ranks IN THE LIBRARY'S DEFAULT MODE: the reference-order kernels, whose logits are
bit-identical to the reference's x86 build (tests/test_parity_7b_gpu.py).  Decode (N=1, greedy position stepping) is reported
beside it; the opt-in fast mode (FL_FAST=1: exact integer block dots, own f32 summation order, ~1e-2 on 7B logits) is timed in
the same run and reported under "fast_mode".

--gpus N > 1 (default --parallel auto): the headline is the TENSOR-PARALLEL eval of ONE batch, every matmul split by output rows
  (the reference's own split across threads, lib/ggml.c:8127-8135): four RCCL all-gathers per layer -- the Q8_0 operands of wo / w2
  and their output rows -- and one of the logits over xGMI, nothing summed across ranks, so the sharded logits are still the
  reference's bit for bit; decode replays one hipGraph with the collectives inside -- strong scaling.  (The fast mode keeps
  SURVEY.md 8e's K-block split of wo / w2 with two all-reduces per layer: "fast_mode" of the same line.)  A
  replica leg (every rank a full model and its own batch, no collective; weak scaling) is measured in the same run and
  reported beside it under "replicas".  --parallel dp measures replicas only.
  BASELINE.json configs 4 / 5: `bench.py --model 13B --gpus 2|4` and `bench.py --model 65B --gpus 8 --n-ctx 2048`.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, timed live with HIP events on the eval stream) and
`cpu_baseline` (the reference's own ggml mul_mat path timed on this box's host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PEAK_I8_TOPS = 5000.0      # dense int8 MFMA (2x the ~2.5 PF bf16 dense peak); the K=32 form this path needs is half of it


def cpu_baseline(cfg, N, qtype, budget_s=25.0):
    """Time the reference's ggml_mul_mat graph (oracle/_ref, kind "reference") or, if absent, the C
    restatement (kind "port") on one layer's 7 matmuls + the lm-head at N columns; extrapolate to the
    7*n_layer+1 matmuls of one eval.  Test infrastructure used as a reported baseline only."""
    import oracle
    from harness import synth
    E, F, V = cfg["n_embd"], cfg["n_ff"], cfg["n_vocab"]
    shapes = {"EE": (E, E, 4), "FE": (F, E, 2), "EF": (E, F, 1), "VE": (V, E, 0)}
    ncpu = os.cpu_count() or 1
    use_ref = oracle.have_ref()
    ref = oracle.Ref() if use_ref else None
    port = oracle.Port()
    host = {}
    for k, (M, K, _) in shapes.items():
        host[k] = (synth.synth_q4(M, K, qtype, 99).cpu().numpy(),
                   np.random.default_rng(1).standard_normal((N, K), dtype=np.float32))

    def run(k, T, reps=1):
        wq, x = host[k]
        if use_ref:
            ts, _ = ref.timed_mul_mat(qtype, wq, x, T, reps)
            return float(np.median(ts))
        t0 = time.perf_counter()
        port.mul_mat_q(qtype, wq, x, n_threads=T)
        return time.perf_counter() - t0

    # pick the thread count on the smallest shape (the reference's spin-wait pool degrades when
    # oversubscribed), then time every shape at that count (median of 3)
    cands = sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu})
    best_T, best_t = cands[0], float("inf")
    t_begin = time.perf_counter()
    for T in cands:
        t = run("EE", T, reps=2)
        if t < best_t:
            best_T, best_t = T, t
        if time.perf_counter() - t_begin > budget_s * 0.4:
            break
    times = {k: run(k, best_T, reps=3) for k in shapes}
    per_eval = cfg["n_layer"] * sum(times[k] * shapes[k][2] for k in shapes) + times["VE"]
    return {
        "value": N / per_eval, "unit": "tokens/s", "cores": best_T,
        "kind": "reference" if use_ref else "port",
        "sample": (f"one layer's 7 mul_mat_q_f32 + lm-head at N={N} through the reference's ggml_graph_compute "
                   f"({best_T} threads of {ncpu} logical CPUs, median of 3), extrapolated to the {7 * cfg['n_layer'] + 1} "
                   f"matmuls of one eval; attention/norm ops not included (favours the CPU)"),
        "seconds_measured": sum(times.values()) * 3, "per_shape_s": times,
    }


def cpu_config1(cfg, qtype, budget_s=12.0):
    """BASELINE.json configs[0]: the reference's CPU path at num_threads = 8, n_ctx = 512, n_batch = 128 -- the
    same bounded sample as cpu_baseline (one layer's matmuls + lm-head through the reference's ggml_graph_compute, extrapolated
    to one 128-token eval), at exactly 8 threads."""
    import oracle
    from harness import synth
    if not oracle.have_ref():
        return None
    E, F, V = cfg["n_embd"], cfg["n_ff"], cfg["n_vocab"]
    shapes = {"EE": (E, E, 4), "FE": (F, E, 2), "EF": (E, F, 1), "VE": (V, E, 0)}
    ref = oracle.Ref()
    times = {}
    for k, (M, K, _) in shapes.items():
        wq = synth.synth_q4(M, K, qtype, 99).cpu().numpy()
        x = np.random.default_rng(1).standard_normal((128, K), dtype=np.float32)
        ts, _ = ref.timed_mul_mat(qtype, wq, x, 8, 2)
        times[k] = float(np.median(ts))
    per_eval = cfg["n_layer"] * sum(times[k] * shapes[k][2] for k in shapes) + times["VE"]
    return {"value": 128 / per_eval, "unit": "tokens/s", "cores": 8, "kind": "reference", "n_batch": 128, "n_ctx": 512,
            "sample": "BASELINE.json configs[0]: one layer's 7 mul_mat_q_f32 + lm-head at N=128, 8 threads, median of 2, extrapolated to one eval"}


def cpu_end_to_end(cfg, N, qtype):
    """SURVEY.md 8(d)'s CPU baseline: the SAME synthetic 7B model as a GGJT file through the reference's own library (oracle/_ref:
    llama_load_model, then one n_batch eval of N tokens via llama_ingest + the first step of llama_generate, exactly what
    tests/test_parity_7b_gpu.py compares bit for bit) on this box's host cores.  One eval: ~10-20 s of CPU work."""
    import tempfile
    import oracle
    from harness import ggjt, llama_capi, synth
    if not oracle.have_ref():
        return None
    lib = llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))
    gcfg = dict(n_vocab=cfg["n_vocab"], n_embd=cfg["n_embd"], n_mult=256, n_head=cfg["n_head"], n_layer=cfg["n_layer"])
    nthr = min(32, os.cpu_count() or 8)
    rng = np.random.default_rng(7)
    text = bytes(rng.integers(33, 127, size=N - 2).astype(np.uint8)).decode()       # BOS + the inserted space + N - 2 bytes = N tokens
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.bin")
        ggjt.write_ggjt_stream(path, gcfg, qtype, synth.synth_model_tensors(cfg, qtype, seed=1234))
        t0 = time.perf_counter()
        ref = llama_capi.Session(lib, path, n_ctx=max(1024, 2 * N), n_batch=N, n_threads=nthr, all_logits=False)
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        ok = ref.ingest(text) and ref.generate(1, temp=0.0)[0]
        t_eval = time.perf_counter() - t0
        # ---- the reference's DECODE beside the GPU's (lib/bridge.cpp:240-330): greedy tokens one llama_generate step at a time
        #      -- each step is ONE Model::eval of one token at n_past = N + i plus the argmax -- timed per token after the warm-up step above
        n_dec = 16
        t_tok = []
        for _ in range(n_dec if ok else 0):
            t1 = time.perf_counter()
            if not ref.generate(1, temp=0.0)[0]:
                break
            t_tok.append(time.perf_counter() - t1)
        ref.close()
    if not ok:
        return None
    out = {"value": N / t_eval, "unit": "tokens/s", "cores": nthr, "kind": "reference", "seconds_eval": t_eval, "seconds_load": t_load,
           "sample": (f"the reference's own llama_ingest + first llama_generate step = ONE Model::eval of {N} tokens on the same synthetic "
                      f"7B file, {nthr} threads of {os.cpu_count()} logical CPUs (includes tokenizing {N} bytes and one argmax)")}
    if t_tok:
        med = float(np.median(t_tok))
        out["decode"] = {"value": 1.0 / med, "unit": "tokens/s", "cores": nthr, "kind": "reference", "n_past": N + 1, "tokens": len(t_tok),
                         "ms_per_token": {"median": med * 1e3, "min": min(t_tok) * 1e3, "max": max(t_tok) * 1e3},
                         "sample": (f"{len(t_tok)} greedy llama_generate steps (temp 0: one Model::eval of one token + argmax each) behind the "
                                    f"{N}-token prompt on the same synthetic 7B file, {nthr} threads; median per token")}
    return out


def self_launch(n):
    """Re-run this command line under torch.distributed.run with n ranks on 127.0.0.1; returns its exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL / peer-mapped buffers across processes
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="7B", help="7B | 13B | 30B | 65B (BASELINE.json configs 4/5: --model 13B|65B --gpus N)")
    ap.add_argument("--qtype", default="q4_0", choices=["q4_0", "q4_1"])
    ap.add_argument("--n-batch", type=int, default=512)
    ap.add_argument("--decode-steps", type=int, default=64)
    ap.add_argument("--n-ctx", type=int, default=0, help="context size (default max(1024, 2*n_batch)); the long-context "
                    "decode leg runs at its end")
    ap.add_argument("--parallel", default="auto", choices=["auto", "dp", "tp"],
                    help="world > 1: tp = ONE batch, Megatron split + RCCL (the headline, strong scaling; a replica leg is measured "
                         "and reported beside it), dp = replicas only.  auto = tp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--peer-exchange", action="store_true", help="(kept for old command lines: the peer-mapped exchange for decode-size messages is on by default since round 5; FL_P2P=0 switches it off)")
    ap.add_argument("--tp-timeout", type=int, default=300, help="seconds the tensor-parallel leg may take before the replica leg is reported alone")
    ap.add_argument("--no-fast", action="store_true", help="skip the fast-mode timings reported beside the headline")
    ap.add_argument("--no-cpu-e2e", action="store_true", help="skip cpu_baseline.end_to_end (the reference's own eval of the same 7B file)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs of BASELINE.json configs 3 / 4 / 5 on this GPU "
                    "(7B Q4_1; 13B; 65B at n_ctx 2048), reported under other_configs")
    ap.add_argument("--all-configs", action="store_true", help="(kept for old command lines: configs 3 / 4 / 5 run by default at --gpus 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU over RCCL,
        # the command line the docstring names) and pass rank 0's JSON line through.
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.peer_exchange:
        os.environ["FL_P2P"] = "1"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to measure")
    # development overrides (a 1-GPU box can still run the world > 1 code path: two replicas on one device, gloo for
    # the timing reduction): FL_BENCH_DEVICE pins the device index, FL_BENCH_BACKEND selects the process-group backend
    if "FL_BENCH_DEVICE" in os.environ:
        local = int(os.environ["FL_BENCH_DEVICE"])
    backend = os.environ.get("FL_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from fastllama_amd import hip
    from harness import synth
    from harness.flmodel import FlModel

    qtype = synth.Q4_0 if args.qtype == "q4_0" else synth.Q4_1
    cfg = dict(synth.MODELS[args.model])
    N = args.n_batch
    n_ctx = max(args.n_ctx, 2048, 2 * N)      # (2048: the deep-context legs -- prefill at n_past 1536, decode at the end of the context)
    L = hip.load()
    hip.require_device(local)
    # development: FL_BENCH_P2P_ONLY=1 builds the tensor-parallel communicator from the peer exchange alone (no RCCL), which lets
    # the tensor-parallel leg of this file run as two processes on ONE GPU -- with a model whose messages fit (--model tiny --n-batch 32)
    p2p_only = os.environ.get("FL_BENCH_P2P_ONLY") == "1"
    want_tp = world > 1 and args.parallel in ("auto", "tp") and cfg["n_head"] % world == 0 and (backend == "nccl" or p2p_only)
    wk, wk1 = synth.algorithmic_work(cfg, N, qtype), synth.algorithmic_work(cfg, 1, qtype)
    t_begin_all = time.perf_counter()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        barrier()
        return dt

    def pmc_traffic(kernel, tp):
        """(HBM bytes per launch, source file) from the committed PMC passes (profiles/r05_pmc_traffic.json, else older rounds':
        rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied) -- valid for the default 7B Q4_0
        n_batch=512 workload.  Not measured in this run: PMC counters need rocprofv3 around the process."""
        try:
            if args.model != "7B" or qtype != 2 or N != 512 or tp:
                return None, None
            for rnd in ("r06", "r05", "r04", "r03", "r02"):       # the latest committed pass that has the kernel
                path = os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")
                if os.path.exists(path):
                    with open(path) as f:
                        k = json.load(f)["kernels"]
                    if kernel in k:
                        return k[kernel]["hbm_bytes_per_launch"], "profiles/" + rnd + "_pmc_traffic.json"
            return None, None
        except Exception:
            return None, None

    KERNELS = {   # (prefill GEMM, decode GEMV) of a mode: names as rocprofv3 prints them
        "exact": ("gemm_q4_exact_h16_kernel", "gemv1_q4_exact_stream_kernel"),
        "fast": ("gemm_q4_mfma32_kernel", "gemv_q4_kernel"),
    }

    def run_leg(tp, cfg=cfg, qtype=qtype, N=N, n_ctx=n_ctx, short=False):
        """One model (a replica per rank, or this rank's tensor-parallel shard) through the prefill / decode / roofline legs, in the
        default (reference-order, "exact") mode and -- unless --no-fast -- in the fast mode.  Under tensor parallelism the split of
        wo / w2 belongs to the mode (rows / K blocks), so each mode gets its own model.
        short: prefill + decode timings only, fewer steps (the other BASELINE configs reported beside the headline)."""
        wk, wk1 = synth.algorithmic_work(cfg, N, qtype), synth.algorithmic_work(cfg, 1, qtype)
        steps = max(2, args.steps // 4) if short else args.steps
        dsteps = max(8, args.decode_steps // 4) if short else args.decode_steps
        seqs = 1 if tp else world
        shard = world if tp else 1
        r = {"seqs": seqs, "wk": wk, "wk1": wk1, "shard": shard}
        rng = np.random.default_rng(7 + (0 if tp else rank))
        toks = rng.integers(3, 259, size=N).astype(np.int32)      # uniform ids in [3, 258]
        tok1 = toks[:1].copy()

        def make_model(mode):
            if tp:                      # the split is fixed when the model is created (fl_model_create reads the mode)
                os.environ["FL_EXACT"] = "1" if mode == "exact" else "0"
            model = FlModel(cfg, qtype, synth.synth_model_tensors(cfg, qtype), n_ctx=n_ctx, max_batch=N,
                            tp_rank=rank if tp else 0, tp_size=world if tp else 1, device=local)
            os.environ.pop("FL_EXACT", None)
            comm = None
            if tp and p2p_only:
                comm = L.fl_comm_create_p2p(rank, world)
                if not comm:
                    raise SystemExit("fl_comm_create_p2p failed: " + L.fl_last_error().decode())
                mine = (ctypes.c_ubyte * 128)()
                hip.check(L.fl_comm_p2p_export(ctypes.c_void_p(comm), mine), "p2p_export")
                gathered = [None] * world
                dist.all_gather_object(gathered, bytes(mine))
                allh = b"".join(gathered)
                hip.check(L.fl_comm_p2p_import(ctypes.c_void_p(comm), (ctypes.c_ubyte * len(allh))(*allh)), "p2p_import")
                model.set_comm(ctypes.c_void_p(comm))
                r["peer_exchange"] = True
            elif tp:
                idbuf = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    raw = (ctypes.c_ubyte * 128)()
                    hip.check(L.fl_comm_unique_id(raw))
                    idbuf = torch.tensor(list(raw), dtype=torch.uint8)
                idbuf = idbuf.cuda()
                dist.broadcast(idbuf, src=0)
                raw = (ctypes.c_ubyte * 128)(*idbuf.cpu().tolist())
                comm = L.fl_comm_create(raw, rank, world)
                if not comm:
                    raise SystemExit("fl_comm_create failed: " + L.fl_last_error().decode())
                model.set_comm(ctypes.c_void_p(comm))
                r["peer_exchange"] = bool(L.fl_comm_has_p2p(ctypes.c_void_p(comm)))   # decode-size messages through peer-mapped buffers
            if tp:
                r["n_ranks_rccl"] = int(L.fl_comm_rccl_ranks(ctypes.c_void_p(comm)))
            return model, comm

        def drop(model, comm):
            barrier()
            model.free()
            if comm:
                L.fl_comm_destroy(ctypes.c_void_p(comm))
            torch.cuda.empty_cache()

        def time_mode(model, mode, full):
            """prefill + decode (+ roofline legs when `full`) of `model` in `mode`"""
            model.set_exact(mode == "exact")
            m = {}
            # what a llama_eval() caller gets: token ids in, the LAST token's logits back on the host -- inside the timed region
            lg_host = np.empty(cfg["n_vocab"], dtype=np.float32)
            prefill = lambda i: model.eval_last_logits(toks, 0, lg_host)
            dec = lambda i: model.eval_last_logits(tok1, min(128, N) + i, lg_host)
            nst = steps if mode == "exact" or short else max(2, steps // 2)
            # ---- prefill: K timed evals after W warm-ups (the first exact eval also builds the f16 fragment copies of the weights)
            for i in range(max(1, args.warmup)):
                prefill(i)
            dt = timed(prefill, nst)
            m["ms_per_step"] = dt / nst * 1e3
            m["prefill_tokens_per_s"] = N * seqs / (dt / nst)
            # ---- decode: N = 1 at n_past = 128.. (KV holds the prefill)
            for i in range(3):
                dec(i)
            # three passes of dsteps steps: the MEDIAN is the number, every pass is reported (round 5 took the better of two because a pass now
            # and then ran ~40 % slow; profiles/r06_decode_transient.md has what that was)
            def passes(fn, n, k=3):
                ts = [timed(fn, n) / n * 1e3 for _ in range(k)]
                return float(np.median(ts)), ts
            m["decode_ms"], m["decode_passes_ms"] = passes(dec, dsteps)
            m["decode_tokens_per_s"] = seqs / (m["decode_ms"] * 1e-3)
            # the same steps with the logits left in HBM (no host round trip per token): what the kernels alone sustain
            dec_nc = lambda i: model.eval_nocopy(tok1, min(128, N) + i)
            m["decode_ms_device_resident"], m["decode_passes_ms_device_resident"] = passes(dec_nc, dsteps)
            # launches of a decode token = kernel nodes of the replayed hipGraph (under the row split: are the exchanges tails of the producers?)
            m["decode_graph_nodes"] = int(hip.load().fl_model_graph_nodes(model.h))
            m["tp_decode_folded"] = bool(hip.load().fl_model_tp_folded(model.h))
            # ---- decode at the end of the context (p ~ n_ctx - 1; the K/V stream of n_past positions per layer, split attention)
            lsteps_ = min(32 if not short else 12, args.decode_steps)
            m["long_past"] = n_ctx - lsteps_ - 4
            decl = lambda i: model.eval_last_logits(tok1, m["long_past"] + i, lg_host)
            for i in range(3):
                decl(i)
            m["decode_long_ms"], m["decode_long_passes_ms"] = passes(decl, lsteps_)
            if not full:
                return m
            # ---- a later chunk of a long prompt: the same N tokens behind n_ctx - N cached positions (the reference-order attention over
            #      1536 .. 2047 keys: the piece-by-piece V.P loop; positions beyond the first eval hold zeros: fine for timing)
            deep_past = n_ctx - N
            if deep_past >= N:
                deep = lambda i: model.eval_last_logits(toks, deep_past, lg_host)
                deep(0)
                dsteps_ = max(2, nst // 2)
                dtd = timed(deep, dsteps_)
                m["prefill_deep"] = {"n_past": deep_past, "ms_per_step": dtd / dsteps_ * 1e3, "tokens_per_s": N * seqs / (dtd / dsteps_)}
            # ---- decode in the middle of the context (round 4's long-context point, n_past 988)
            if n_ctx >= 1100:
                decm = lambda i: model.eval_last_logits(tok1, 988 + i, lg_host)
                for i in range(3):
                    decm(i)
                m["decode_988_ms"], _ = passes(decm, 24)
            # ---- roofline of the dominant kernels: HIP events around every matmul launch on the eval stream
            gemm, gemv = KERNELS[mode]
            model.profile(1)
            evals = max(2, args.steps // 3)
            for i in range(evals):
                prefill(i)
            mm_ms, n_launch = model.profile(0)
            tops = wk["flops"] / shard * evals / (mm_ms * 1e-3) / 1e12
            traffic, tsrc = pmc_traffic(gemm, tp)
            note = ("ALGORITHMIC 2*M*K*N of the %d mul_mat_q_f32 (fused into %d launches) / event-timed launch durations; peak = the 5 POP/s "
                    "dense int8 MFMA rate.  " % (wk["n_matmuls"], n_launch // evals))
            if mode == "exact":
                note += ("The reference's summation order needs, per output and 32-element block, 8 separate 4-element sums (2 per "
                         "v_mfma_f32_32x32x4_2b_f16: 256 matrix cycles per 32x32 tile and block, the pipe's output rate whatever the shape) and "
                         "8 f32 fma (256+ VALU cycles); the two pipes of a gfx950 SIMD do not overlap: measured floor of the mix 232 ns per "
                         "tile and block = 0.29 POP/s chip-wide (DESIGN.md 3.7, profiles/r04_ubench_coexec5.txt)")
            else:
                note += ("v_mfma_i32_32x32x32_i8 (K = 32 = one quant block) runs at that rate; the exact per-block scaling adds 32 VALU ops + "
                         "half an f32 outer-product MFMA per 32x32 tile and block: measured instruction-mix ceiling ~1.0 POP/s (DESIGN.md 3.1)")
            m["roofline"] = {
                "kernel": "%s<Q4_%d,...>" % (gemm, qtype - 2), "bound": "mfma",
                "achieved": tops, "peak": PEAK_I8_TOPS, "unit": "TOP/s", "frac": tops / PEAK_I8_TOPS,
                "traffic": traffic, "traffic_source": tsrc,
                "launches_per_step": n_launch // evals, "avg_launch_us": mm_ms * 1e3 / max(1, n_launch),
                "algorithmic_flops_per_launch": wk["flops"] / shard / (n_launch / evals), "note": note,
            }
            if mode == "exact":
                # the ceiling the reference's summation order has on this chip: 247 ns per 32x32 tile and block on each of the 1024 SIMDs
                # (8 lane sums at the matrix pipe's output rate + 8 fma per output + the d_w x d_x outer product, the two pipes of a SIMD not
                # overlapping: DESIGN.md 3.7, profiles/r04_ubench_coexec5.txt) = 65536 ops / 247 ns x 1024
                floor_tops = 65536.0 / 247e-9 * 1024 / 1e12
                m["roofline"]["order_floor"] = {"ns_per_tile_block": 247.0, "tops": floor_tops, "frac": tops / floor_tops,
                                                "note": "event-timed launches (ramp, epilogue and partial rounds included) against the measured floor "
                                                        "of the instruction mix the reference's accumulation order needs on gfx950"}
            model.profile(1)
            for i in range(8):
                dec(i)
            mm1_ms, n1 = model.profile(0)
            gbs_ev = wk1["bytes"] / shard * 8 / (mm1_ms * 1e-3) / 1e9
            # achieved = ALGORITHMIC bytes of a token / the graph-replayed time per token with the logits left in HBM (every kernel of the
            # token, attention included: plain event-bracketed launches are SLOWER than the replayed graph they would be compared with)
            gbs = wk1["bytes"] / shard / (m["decode_ms_device_resident"] * 1e-3) / 1e9
            traffic1, tsrc1 = pmc_traffic(gemv, tp)
            m["roofline_decode"] = {
                "kernel": "%s<Q4_%d,...>" % (gemv, qtype - 2), "bound": "hbm",
                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "timing": "hipGraph replay of the whole token (logits left in HBM), all kernels",
                "event_timed_plain_launches": {"achieved": gbs_ev, "frac": gbs_ev / PEAK_HBM_GBS, "avg_launch_us": mm1_ms * 1e3 / max(1, n1),
                                               "note": "HIP events around every matmul launch outside the graph: matmul kernels only, each paying a plain launch"},
                "traffic": traffic1, "traffic_source": tsrc1,
                "launches_per_step": n1 // 8,
                "algorithmic_bytes_per_launch": wk1["bytes"] / shard / (n1 / 8),
            }
            return m

        model, comm = make_model("exact")
        r["exact"] = time_mode(model, "exact", not short)
        if not short:
            r["long_past"] = r["exact"]["long_past"]
            r["decode_long_ms"] = r["exact"]["decode_long_ms"]
            r["model_device_bytes"] = hip.load().fl_model_device_bytes(model.h)
            mem = (ctypes.c_size_t * 5)()
            if hip.load().fl_model_memory(model.h, mem) == 0:
                r["model_memory_bytes"] = dict(zip(("qw16_nibble_planes", "scale_planes", "wh16_copies", "qwd_copies", "kv_cache_and_work_buffers"), (int(v) for v in mem)))
            # ---- a prompt longer than n_batch: the session's ingest loop evaluates it n_batch tokens at a time; fl_model_ingest keeps
            #      two of those evals in flight (same results).  Reported beside the headline, which stays the single n_batch eval.
            if not tp and n_ctx >= 2 * N:
                n_long = min(n_ctx // N, 4) * N
                tl = rng.integers(3, 259, size=n_long).astype(np.int32)
                one_by_one = lambda i: [model.eval_nocopy(tl[j:j + N], j) for j in range(0, n_long, N)]
                pipelined = lambda i: model.ingest(tl, N, want_logits=False)
                lsteps = max(2, args.steps // 4)
                one_by_one(0); pipelined(0)
                t_seq, t_pipe = timed(one_by_one, lsteps) / lsteps, timed(pipelined, lsteps) / lsteps
                r["long_prompt"] = {"tokens": n_long, "n_batch": N, "tokens_per_s": n_long * seqs / t_pipe, "ms": t_pipe * 1e3,
                                    "chunk_by_chunk_tokens_per_s": n_long * seqs / t_seq,
                                    "note": "fl_model_ingest: the consecutive n_batch evals of one prompt, two in flight on two streams (one stream from 65B width on); bit-identical to chunk by chunk"}
        if not args.no_fast:
            if tp:                      # the fast mode's own split (K blocks of wo / w2, all-reduces)
                drop(model, comm)
                model, comm = make_model("fast")
            r["fast"] = time_mode(model, "fast", not short)
        drop(model, comm)
        return r

    def summary(leg, name):
        """the short form of a leg: other_configs entries"""
        t_hbm = leg["wk"]["bytes"] / leg["shard"] / (PEAK_HBM_GBS * 1e9) * 1e3
        t_hbm1 = leg["wk1"]["bytes"] / leg["shard"] / (PEAK_HBM_GBS * 1e9) * 1e3
        x = leg["exact"]
        o = {"config": name, "prefill_tokens_per_s": x["prefill_tokens_per_s"], "ms_per_step": x["ms_per_step"],
             "decode_tokens_per_s": x["decode_tokens_per_s"],
             "decode_long_context": {"n_past": x["long_past"], "tokens_per_s": leg["seqs"] / (x["decode_long_ms"] * 1e-3), "ms_per_token": x["decode_long_ms"]},
             "hbm_roofline_frac": {"prefill": t_hbm / x["ms_per_step"], "decode": t_hbm1 / x["decode_ms"]}}
        if "fast" in leg:
            f = leg["fast"]
            o["fast_mode"] = {"prefill_tokens_per_s": f["prefill_tokens_per_s"], "decode_tokens_per_s": f["decode_tokens_per_s"],
                              "decode_long_context_tokens_per_s": leg["seqs"] / (f["decode_long_ms"] * 1e-3),
                              "hbm_roofline_frac": {"prefill": t_hbm / f["ms_per_step"], "decode": t_hbm1 / f["decode_ms"]}}
        return o

    def emit(legs, tp_error, others=()):
        leg = legs["tp"] if "tp" in legs else legs["dp"]
        tp = "tp" in legs
        head = leg["exact"]
        # fraction of the HBM roofline BASELINE.json's target is written in: time to move the ALGORITHMIC bytes of one step
        # (Q4 weights once + f32 activations in and out of the 225 matmuls) at 8 TB/s / measured time per step
        shard = world if tp else 1
        t_hbm_prefill_ms = wk["bytes"] / shard / (PEAK_HBM_GBS * 1e9) * 1e3
        t_hbm_decode_ms = wk1["bytes"] / shard / (PEAK_HBM_GBS * 1e9) * 1e3
        out = {
            "metric": "tokens/sec (prefill n_batch=512 + decode) LLaMA-7B Q4_0",
            "value": head["prefill_tokens_per_s"], "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong" if tp else "weak", "vs_baseline": None,
            "dtype": "i8", "data": "synthetic",
            "config": {
                "workload": (f"LLaMA-{args.model} {args.qtype.upper()} n_batch={N} prefill; step = one full device-resident "
                             f"Model::eval (n_past=0, {wk['n_matmuls']} mul_mat_q_f32 + attention/norm/rope ops), synthetic weights, "
                             f"reference-order arithmetic (logits bit-identical to the reference's)"),
                "n_batch": N, "n_ctx": n_ctx, "global_batch_tokens": N * leg["seqs"],
                "parallelism": (f"tp{world} (ONE batch: every matmul by output rows; prefill: 4 all-gathers per layer (Q8_0 operands of wo / w2, their "
                                f"output rows) + 1 of the logits; decode: the same four exchanges as tails of the producing launches when the communicator "
                                f"has the peer-mapped exchange (tp_decode); nothing summed across ranks)" if tp else
                                f"dp{world} (one model replica and one batch per GPU, no data-path collective)"),
            },
            "mode": "exact",
            "mode_note": ("value / ms_per_step / roofline = the library's DEFAULT mode: the reference's summation order in every matmul and "
                          "attention dot -- logits bit-identical to the reference's x86 build on this configuration (tests/test_parity_7b_gpu.py), "
                          "also under tensor parallelism (tests/test_wide_models_gpu.py).  fast_mode = FL_FAST=1 / fl_model_set_exact(m, 0): exact "
                          "integer block dots, per-block f32 terms added in the kernels' own order (1e-7 per matmul; ~1e-2 on 7B logits after 32 layers)"),
            "prefill_tokens_per_s": head["prefill_tokens_per_s"],
            "decode_tokens_per_s": head["decode_tokens_per_s"], "decode_ms_per_token": head["decode_ms"], "decode_passes_ms_per_token": head["decode_passes_ms"],
            "decode_device_resident": {"tokens_per_s": leg["seqs"] / (head["decode_ms_device_resident"] * 1e-3), "ms_per_token": head["decode_ms_device_resident"], "passes_ms_per_token": head["decode_passes_ms_device_resident"],
                                       "note": "the same steps with the logits left in HBM (no host round trip per token)"},
            "timed_region": "token ids in (host), the last token's logits back on the host, per eval: prefill and decode alike; decode legs: the MEDIAN of three passes of --decode-steps steps, every pass reported",
            "decode_long_context": {"n_past": leg["long_past"], "tokens_per_s": leg["seqs"] / (leg["decode_long_ms"] * 1e-3),
                                    "ms_per_token": leg["decode_long_ms"]},
            "hbm_roofline": {"peak_GBs": PEAK_HBM_GBS,
                             "prefill": {"algorithmic_bytes_per_step": wk["bytes"] / shard, "t_hbm_ms": t_hbm_prefill_ms,
                                         "frac": t_hbm_prefill_ms / head["ms_per_step"]},
                             "decode": {"algorithmic_bytes_per_token": wk1["bytes"] / shard, "t_hbm_ms": t_hbm_decode_ms,
                                        "frac": t_hbm_decode_ms / head["decode_ms"]},
                             "note": "whole-step fractions (every kernel of the eval, not only the matmuls); BASELINE.json's target is 0.40 for prefill"},
            "prefill_deep_context": head.get("prefill_deep"),
            "decode_n_past_988": ({"tokens_per_s": leg["seqs"] / (head["decode_988_ms"] * 1e-3), "ms_per_token": head["decode_988_ms"]} if "decode_988_ms" in head else None),
            "prefill_long_prompt": leg.get("long_prompt"),
            "roofline": head["roofline"], "roofline_decode": head["roofline_decode"],
            "model_device_bytes": leg["model_device_bytes"], "model_memory_bytes": leg.get("model_memory_bytes"),
            # is the headline the tensor-parallel eval over RCCL, and how many ranks does the RCCL communicator itself count?
            "tp_ok": bool(tp) if world > 1 else None, "n_ranks_rccl": leg.get("n_ranks_rccl", 0) if world > 1 else None,
        }
        if "fast" in leg:
            f = leg["fast"]
            out["fast_mode"] = {"prefill_tokens_per_s": f["prefill_tokens_per_s"], "ms_per_step": f["ms_per_step"],
                                "decode_tokens_per_s": f["decode_tokens_per_s"], "decode_ms_per_token": f["decode_ms"],
                                "hbm_roofline_frac": {"prefill": t_hbm_prefill_ms / f["ms_per_step"], "decode": t_hbm_decode_ms / f["decode_ms"]},
                                "roofline": f["roofline"], "roofline_decode": f["roofline_decode"], "prefill_deep_context": f.get("prefill_deep"),
                                "parity": "logits ~1e-2 from the reference on this configuration (profiles/r03_parity_7b.json): opt-in, not the contract"}
            if tp:
                out["fast_mode"]["parallelism"] = f"tp{world}: wo / w2 by K blocks, 2 RCCL all-reduces of the [N, n_embd] partial sums per layer"
        if others:
            out["other_configs"] = list(others)
        if tp and "dp" in legs:
            d = legs["dp"]["exact"]
            out["replicas"] = {"scaling": "weak", "parallelism": f"dp{world}: a full replica and its own batch per GPU, no collective",
                               "prefill_tokens_per_s": d["prefill_tokens_per_s"], "ms_per_step": d["ms_per_step"],
                               "decode_tokens_per_s": d["decode_tokens_per_s"]}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, N, qtype)
                out["cpu_baseline"]["config1"] = cpu_config1(cfg, qtype)
                if not args.no_cpu_e2e and args.model == "7B":
                    e2e = cpu_end_to_end(cfg, N, qtype)
                    if e2e and "decode" in e2e:
                        out["cpu_baseline"]["decode"] = e2e.pop("decode")
                    out["cpu_baseline"]["end_to_end"] = e2e
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "unavailable",
                                       "sample": f"failed: {e!r}"}
        if tp:
            out["tp_small_message_path"] = "peer-mapped buffers (hipIpc), one kernel per rank" if leg.get("peer_exchange") else "RCCL"
            nl = int(cfg["n_layer"])
            out["tp_decode"] = {
                "exchanges_folded_into_producers": head["tp_decode_folded"], "graph_nodes_per_token": head["decode_graph_nodes"],
                "launches_per_layer": (head["decode_graph_nodes"] - 4) / nl if head["decode_graph_nodes"] else None,
                "note": ("kernel nodes of the replayed decode hipGraph; -4: token lookup, lm-head, the logits' gather (exchange + permute).  Folded: the "
                         "four exchanges of a layer are the tails of the launches that produce the data (peer-mapped fold regions, tp_tail.h): "
                         "wq|wk|wv, attention, wo, w1|w3, w2 and no collective launch; otherwise pack -> all-gather -> unpack (-> add) per exchange")}
        if tp_error:
            out["tp_error"] = tp_error + " -- the headline above is the replica leg"
        if rank == 0:
            print(json.dumps(out), flush=True)

    legs = {}
    others = []
    tp_error = None
    if world == 1 or not want_tp or args.parallel == "auto":
        legs["dp"] = run_leg(False)          # replicas (world == 1: the single-GPU measurement)
    if world == 1 and args.model == "7B" and args.qtype == "q4_0" and not args.no_other_configs:
        # BASELINE.json config 3 under the same clock as the headline; configs 4 / 5 (single-GPU form) on request
        try:
            others.append(summary(run_leg(False, qtype=synth.Q4_1, short=True), f"LLaMA-7B Q4_1 n_batch={N} (BASELINE config 3), 1 GPU"))
            others.append(summary(run_leg(False, cfg=dict(synth.MODELS["13B"]), short=True), f"LLaMA-13B Q4_0 n_batch={N} (BASELINE config 4 on 1 GPU)"))
            others.append(summary(run_leg(False, cfg=dict(synth.MODELS["65B"]), n_ctx=2048, short=True),
                                  f"LLaMA-65B Q4_0 n_batch={N} n_ctx=2048 (BASELINE config 5 on 1 GPU)"))
        except Exception as e:  # noqa: BLE001 -- never lose the headline to a side leg
            others.append({"config": "other configs", "error": repr(e)})
    if want_tp:
        # The tensor-parallel leg is the one part of this file that no 1-GPU box can rehearse with more than one rank.  It must
        # not cost the run its line: an exception or a hang (watchdog) falls back to the replica leg, measured above, and says so.
        import signal

        def on_alarm(signum, frame):
            if rank == 0 and "dp" in legs:
                emit({"dp": legs["dp"]}, f"tensor-parallel leg did not finish within {args.tp_timeout} s", others)
            os._exit(0 if "dp" in legs else 3)

        signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(args.tp_timeout)
        try:
            legs["tp"] = run_leg(True)
        except BaseException as e:  # noqa: BLE001 -- SystemExit from a failed fl_comm_create included
            tp_error = repr(e)
        try:                          # every rank must take the same branch below
            ok = torch.tensor([0 if tp_error else 1], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and not tp_error:
                tp_error = "another rank failed in the tensor-parallel leg"
        except BaseException as e:  # noqa: BLE001
            tp_error = tp_error or repr(e)
        signal.alarm(0)
        if tp_error:
            legs.pop("tp", None)
            if "dp" not in legs:
                raise SystemExit("tensor-parallel leg failed and no replica leg was requested: " + tp_error)
    emit(legs, tp_error, others)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
