"""A/B of a GEMM dispatch choice inside one process: python scripts/dev/ab_cfg.py <force value A> <force value B> [steps]
(fl_debug_set(0, v): -1 = the default choice, -3 = never the mixed-tile launch, -2 = the round-1 kernel, >= 0 one configuration)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
va, vb = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
qt = int(sys.argv[4]) if len(sys.argv) > 4 else 2
L = hip.load()
cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, qt, synth.synth_model_tensors(cfg, qt), n_ctx=1024, max_batch=512)
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
for rep in range(3):
    for v in (va, vb):
        L.fl_debug_set(0, v)
        for _ in range(2): m.eval_nocopy(toks, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): m.eval_nocopy(toks, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"force={v:3d}: {dt*1e3:7.3f} ms/eval  {512/dt:8.0f} tok/s", flush=True)
L.fl_debug_set(0, -1)
