"""Reference-order prefill of 512 tokens at n_past 0 / 512 / 1536 with the V.P form switched between evals (fl_debug_set(8, .): 4 or 8 waves per workgroup behind a deep context): python scripts/dev/prefill_pv_ab.py [forms] [n_pasts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
forms = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 4]
pasts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 512, 1536]
cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=2048, max_batch=512)
m.prepare(1)
L = hip.load()
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
for p in range(0, 2048, 512):
    m.eval_nocopy(toks, p)
for rep in range(2):
    for past in pasts:
        for f in forms:
            L.fl_debug_set(8, f)
            for _ in range(2):
                m.eval_nocopy(toks, past)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                m.eval_nocopy(toks, past)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"form {f} n_past {past}: {dt * 1e3:.2f} ms/eval  {512 / dt:.0f} tok/s", flush=True)
