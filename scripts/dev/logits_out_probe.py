"""decode step time with the last token's logits copied to the host per step vs left in HBM, both modes: python scripts/dev/logits_out_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=2048, max_batch=512)
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
t1 = toks[:1].copy()
lg = np.empty(cfg["n_vocab"], np.float32)
lgp = torch.empty(cfg["n_vocab"], dtype=torch.float32).pin_memory().numpy()
for mode in (True, False, True, False):
    m.set_exact(mode)
    m.eval_nocopy(toks, 0)
    for name, fn in (("logits left in HBM", lambda i: m.eval_nocopy(t1, 128 + i)), ("logits -> pageable host buffer", lambda i: m.eval_last_logits(t1, 128 + i, lg)),
                     ("logits -> pinned host buffer", lambda i: m.eval_last_logits(t1, 128 + i, lgp))):
        for i in range(3): fn(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(64): fn(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 64
        print(f"{'exact' if mode else 'fast'}: {name}: {dt * 1e3:.3f} ms/token  {1 / dt:.1f} tok/s", flush=True)
