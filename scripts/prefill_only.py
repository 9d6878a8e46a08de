"""Prefill-only run for profiling: python scripts/prefill_only.py [steps] [n_batch] [n_past,n_past,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = dict(synth.MODELS["7B"])
pasts = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
n_ctx = max(1024, (max(pasts) + nb + 511) // 512 * 512)
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=n_ctx, max_batch=512)
toks = np.random.default_rng(0).integers(3, 259, nb).astype(np.int32)
for past in pasts:
    for _ in range(2):
        m.eval_nocopy(toks, past)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.eval_nocopy(toks, past)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"prefill n_past={past}: {dt*1e3:.3f} ms/eval  {nb/dt:.1f} tok/s")
