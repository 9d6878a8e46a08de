"""Exact (reference-order) vs fast mode: per-shape matmul timings and whole-eval prefill / decode timings on the 7B shape.
Development aid: python scripts/exact_perf.py [--model 7B] [--shapes] [--eval]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastllama_amd import hip, ops
from harness import synth
from harness.flmodel import FlModel

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7B")
ap.add_argument("--shapes", action="store_true")
ap.add_argument("--eval", action="store_true")
ap.add_argument("--qtype", type=int, default=2)
ap.add_argument("--layers", type=int, default=0)
args = ap.parse_args()
L = hip.load()
hip.require_device(0)
if args.shapes or not args.eval:
    for (M, K) in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32000, 4096)]:
        W = ops.QTensor(args.qtype, synth.synth_q4(M, K, args.qtype, 3), M, K)
        for N in (1, 4, 8, 64, 512):
            x = torch.randn(N, K, device="cuda")
            a = ops.QAct(N, K).quantize(x)
            y = torch.empty(N, M, device="cuda")
            res = []
            for which in (None, 3):
                ops.mul_mat_q(W, a, which=which, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    ops.mul_mat_q(W, a, which=which, out=y)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / reps * 1e3)
            wb = M * K // 32 * (20 if args.qtype == 2 else 24)
            print(f"M={M:6d} K={K:6d} N={N:4d}  fast {res[0]:9.1f} us  exact {res[1]:9.1f} us  ratio {res[1]/res[0]:5.2f}  "
                  f"exact: {wb/res[1]/1e6:7.2f} TB/s(W) {2.0*M*K*N/res[1]/1e6:9.1f} GOP/s", flush=True)
        W.free()
if args.eval:
    cfg = dict(synth.MODELS[args.model])
    if args.layers:
        cfg["n_layer"] = args.layers
    m = FlModel(cfg, args.qtype, synth.synth_model_tensors(cfg, args.qtype, seed=1234), n_ctx=1024, max_batch=512)
    toks = np.random.default_rng(7).integers(3, 259, 512).astype(np.int32)
    for exact in (0, 1):
        m.set_exact(bool(exact))
        m.eval_nocopy(toks, 0)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 5 if exact else 20
        for _ in range(reps):
            m.eval_nocopy(toks, 0)
        tp = (time.time() - t0) / reps
        one = toks[:1]
        for i in range(8):
            m.eval_nocopy(one, 512 + i)
        t0 = time.time()
        nd = 64
        for i in range(nd):
            m.eval_nocopy(one, 128 + i)
        td = (time.time() - t0) / nd
        print(f"{args.model} L={cfg['n_layer']} exact={exact}: prefill512 {tp*1e3:8.2f} ms = {512/tp:9.0f} tok/s   decode {td*1e3:7.3f} ms = {1/td:7.1f} tok/s", flush=True)
    m.free()
