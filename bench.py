#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: tokens/s of the LLaMA-7B Q4_0 hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (the 225 quantized matmuls of one Model::eval: Q8_0 activation
quantization + Q4 x Q8 block-dot matmul) over one n_batch=512 batch of synthetic activations that are
already resident in HBM, on synthetic LLaMA-7B-shaped Q4_0 weights (BASELINE.json configs[1]).
`value` = prefill tokens/s summed over all ranks; every rank owns a full replica of the model and its
own batch (the path's units -- activation columns -- are independent, so N GPUs shard tokens with no
data-path collective: weak scaling).  Decode (N=1 greedy, the wave-dot GEMV) is reported beside it.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the launch stream)
and `cpu_baseline` (the reference's own ggml path timed on this box's host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PEAK_I8_TOPS = 5000.0      # dense int8 MFMA (2x the ~2.5 PF bf16 dense peak); K=32 legacy form is half of it


def cpu_baseline(cfg, N, qtype, budget_s=25.0):
    """Time the reference's ggml_mul_mat graph (oracle/_ref, kind "reference") or, if absent, the C
    restatement (kind "port") on one layer's 7 matmuls + the lm-head at N columns; extrapolate to the
    7*n_layer+1 matmuls of one eval.  Test infrastructure used as a reported baseline only."""
    import numpy as np
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
            return min(ts)
        t0 = time.perf_counter()
        port.mul_mat_q(qtype, wq, x, n_threads=T)
        return time.perf_counter() - t0

    # pick the thread count on the smallest shape (the reference's spin-wait pool degrades when
    # oversubscribed), then time every shape once more at that count
    cands = sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu})
    best_T, best_t = cands[0], float("inf")
    t_begin = time.perf_counter()
    for T in cands:
        t = run("EE", T)
        if t < best_t:
            best_T, best_t = T, t
        if time.perf_counter() - t_begin > budget_s * 0.4:
            break
    times = {k: run(k, best_T, reps=2 if k == "EE" else 1) for k in shapes}
    per_eval = cfg["n_layer"] * sum(times[k] * shapes[k][2] for k in shapes) + times["VE"]
    return {
        "value": N / per_eval, "unit": "tokens/s", "cores": best_T,
        "kind": "reference" if use_ref else "port",
        "sample": (f"one layer's 7 mul_mat_q_f32 + lm-head at N={N} through the reference's ggml_graph_compute "
                   f"({best_T} threads of {ncpu}), extrapolated to {7 * cfg['n_layer'] + 1} matmuls"),
        "seconds_measured": sum(times.values()), "per_shape_s": times,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="7B")
    ap.add_argument("--qtype", default="q4_0", choices=["q4_0", "q4_1"])
    ap.add_argument("--n-batch", type=int, default=512)
    ap.add_argument("--decode-steps", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to measure")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from fastllama_amd import hip
    from harness import synth
    from harness.hotpath import HotPath

    qtype = synth.Q4_0 if args.qtype == "q4_0" else synth.Q4_1
    N = args.n_batch
    hp = HotPath(args.model, qtype, max_N=N, device=local)
    L = hip.load()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        barrier()
        return dt

    # ---------------- prefill (the headline): K timed steps after W warm-ups ----------------
    for _ in range(args.warmup):
        hp.step(N)
    dt = timed(lambda: hp.step(N), args.steps)
    ms_per_step = dt / args.steps * 1e3
    value = N * world / (dt / args.steps)

    # ---------------- decode leg: N = 1 ----------------
    for _ in range(3):
        hp.step(1)
    ddt = timed(lambda: hp.step(1), args.decode_steps)
    decode_ms = ddt / args.decode_steps * 1e3

    # ---------------- roofline of the dominant kernel, HIP events on the launch (null) stream ----
    def kernel_only(n, reps):
        hp.prepare(n)
        hp.step(n, quantize=False)
        e0, e1 = L.fl_event_create(), L.fl_event_create()
        torch.cuda.synchronize()
        L.fl_event_record(e0, None)
        for _ in range(reps):
            hp.step(n, quantize=False)
        L.fl_event_record(e1, None)
        import ctypes
        ms = ctypes.c_float()
        hip.check(L.fl_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        L.fl_event_destroy(e0)
        L.fl_event_destroy(e1)
        return ms.value / reps   # ms per pass over all matmul launches

    wk = hp.work(N)
    pre_ms = kernel_only(N, max(2, args.steps // 2))
    n_launch = wk["n_matmuls"]
    tops = wk["flops"] / (pre_ms * 1e-3) / 1e12
    roofline = {
        "kernel": "gemm_q4_mfma_kernel<Q4_%d>" % (qtype - 2), "bound": "mfma",
        "achieved": tops, "peak": PEAK_I8_TOPS, "unit": "TOP/s", "frac": tops / PEAK_I8_TOPS,
        "traffic": None, "launches_per_step": n_launch,
        "avg_launch_us": pre_ms * 1e3 / n_launch,
        "algorithmic_flops_per_launch": wk["flops"] / n_launch,
    }
    wk1 = hp.work(1)
    dec_ms = kernel_only(1, 8)
    gbs = wk1["bytes"] / (dec_ms * 1e-3) / 1e9
    roofline_decode = {
        "kernel": "gemv_q4_kernel<Q4_%d,1>" % (qtype - 2), "bound": "hbm",
        "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
        "traffic": None, "launches_per_step": n_launch, "avg_launch_us": dec_ms * 1e3 / n_launch,
        "algorithmic_bytes_per_launch": wk1["bytes"] / n_launch,
    }

    out = {
        "metric": "tokens/sec (prefill n_batch=512 + decode) LLaMA-7B Q4_0",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8", "data": "synthetic",
        "config": {
            "workload": (f"LLaMA-{args.model} {args.qtype.upper()} n_batch={N} prefill; step = hot path of one "
                         f"eval = {n_launch} mul_mat_q_f32 (Q8_0 INIT + Q4xQ8 COMPUTE), activations resident in HBM"),
            "n_batch": N, "global_batch_tokens": N * world,
            "parallelism": f"dp{world} (one model replica and one batch per GPU, no data-path collective)",
        },
        "prefill_tokens_per_s": value,
        "decode_tokens_per_s": world / (decode_ms * 1e-3), "decode_ms_per_token": decode_ms,
        "roofline": roofline, "roofline_decode": roofline_decode,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(hp.cfg, N, qtype)
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "unavailable",
                                   "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
