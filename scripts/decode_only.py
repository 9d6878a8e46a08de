"""Decode-only run for profiling: python scripts/decode_only.py [steps] [graph 0/1] [gemv waves, 0 = auto] [n_past] [model]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
graph = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # fl_model_set_graph mode bits (1 = hipGraph replay, 64 = f32 w1|w3 product, ...)
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pasts = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [128]
name = sys.argv[5] if len(sys.argv) > 5 else "7B"
cfg = dict(synth.MODELS[name])
n_ctx = int(os.environ.get("FL_NCTX", 0)) or max(1024, (max(pasts) + steps + 8 + 511) // 512 * 512)
QT = int(os.environ.get("FL_QTYPE", "2"))
m = FlModel(cfg, QT, synth.synth_model_tensors(cfg, QT), n_ctx=n_ctx, max_batch=512)
hip.load().fl_model_set_graph(m.h, graph)
if waves:
    hip.load().fl_debug_set(1, waves)                 # force GEMV waves per row group
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
m.eval_nocopy(toks, 0)                                # positions beyond 512 hold zeros: fine for timing
t1 = toks[:1].copy()
for past in pasts:
    for i in range(3):
        m.eval_nocopy(t1, past + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.eval_nocopy(t1, past + 3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name} n_ctx={n_ctx} n_past={past}: decode {dt*1e3:.3f} ms/token  {1/dt:.1f} tok/s (graph={graph})")

L = hip.load()
if hasattr(L, "fl_debug_da_timing"):
    import ctypes as C
    buf = (C.c_longlong * 8)()
    L.fl_debug_da_timing.argtypes = [C.c_void_p]
    L.fl_debug_da_timing(buf)
    t = np.array(buf[:7]); d = (t[1:] - t[:-1]) * 10.0 / 1e3
    print("decode_attention wg0 (us): n_past+prefetch issue %.2f  rope+stores %.2f  (sync) scores %.2f  max/exp %.2f  softmax->probs %.2f  kqv %.2f | total %.2f" % (d[0], d[1], d[2], d[3], d[4], d[5], (t[6]-t[0])*10/1e3))
