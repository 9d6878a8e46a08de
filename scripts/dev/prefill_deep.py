"""Reference-order prefill of n_batch tokens behind n_past cached positions (7B): python scripts/dev/prefill_deep.py [model] [n_pasts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
name = sys.argv[1] if len(sys.argv) > 1 else "7B"
pasts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 512, 1024, 1536]
cfg = dict(synth.MODELS[name])
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=2048, max_batch=512)
m.prepare(1)
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
for p in range(0, 2048, 512):
    m.eval_nocopy(toks, p)                                   # fill the cache with real rows
for past in pasts:
    for _ in range(2):
        m.eval_nocopy(toks, past)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m.eval_nocopy(toks, past)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name} n_batch 512 at n_past {past}: {dt * 1e3:.2f} ms/eval  {512 / dt:.0f} tok/s", flush=True)
