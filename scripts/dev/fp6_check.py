"""fp6 block-scaled form of the prefill GEMM against the i8 form: bit identity and time per launch (the fp6 time includes the
QA16 -> QA16F6 conversion kernel).  Development aid: python scripts/dev/fp6_check.py [--eval]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fastllama_amd import hip, ops
from harness import synth

ap = argparse.ArgumentParser()
ap.add_argument("--eval", action="store_true")
ap.add_argument("--noshapes", action="store_true")
args = ap.parse_args()
L = hip.load()
hip.require_device(0)


def tm(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if not args.noshapes:
    for qt in (2, 3):
        for (M, K, N) in [(12288, 4096, 512), (4096, 4096, 512), (22016, 4096, 512), (4096, 11008, 512), (32000, 4096, 512), (4096, 4096, 40), (200, 1408, 17)]:
            W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 3), M, K)
            x = torch.randn(N, K, device="cuda")
            a = ops.QAct(N, K).quantize(x)
            out = []
            for i8, f6 in ((101, 201), (106, 206), (116, 216)):
                L.fl_debug_set(0, i8)
                y0 = ops.mul_mat_q(W, a).clone()
                t0 = tm(lambda: ops.mul_mat_q(W, a))
                L.fl_debug_set(0, f6)
                y1 = ops.mul_mat_q(W, a).clone()
                t1 = tm(lambda: ops.mul_mat_q(W, a))
                nd = int((y0.view(torch.int32) != y1.view(torch.int32)).sum())
                out.append(f"{i8}:{t0:7.1f} {f6}:{t1:7.1f} diff {nd}")
            L.fl_debug_set(0, -1)
            print(f"q{qt} M={M:6d} K={K:6d} N={N:4d} | " + " | ".join(out), flush=True)
            W.free()
if args.eval:
    from harness.flmodel import FlModel
    cfg = dict(synth.MODELS["7B"])
    m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2, seed=1234), n_ctx=1024, max_batch=512)
    toks = np.random.default_rng(7).integers(3, 259, 512).astype(np.int32)
    res = {}
    for fp6 in (1, 0, 1, 0):
        L.fl_debug_set(3, fp6)
        lg = m.eval(toks, all_logits=True)
        res.setdefault(fp6, lg)
        assert np.array_equal(res[fp6].view(np.int32), lg.view(np.int32))
        m.eval_nocopy(toks, 0); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(20):
            m.eval_nocopy(toks, 0)
        torch.cuda.synchronize()
        tp = (time.time() - t0) / 20
        print(f"fp6={fp6}: prefill {tp*1e3:.2f} ms  {512/tp:.0f} tok/s", flush=True)
    print("logit bits differing fp6 vs i8:", int((res[0].view(np.int32) != res[1].view(np.int32)).sum()))
