"""The headline eval with its host hand-over included: token ids in, ALL 512 x 32000 logits out (64 MB over PCIe) / last-token logits out,
against the device-resident eval bench.py times.  python scripts/dev/pcie_inclusive.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from harness import synth
from harness.flmodel import FlModel
cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2, seed=1234), n_ctx=1024, max_batch=512)
toks = np.random.default_rng(7).integers(3, 259, 512).astype(np.int32)
def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3
a = timed(lambda: m.eval_nocopy(toks, 0))
b = timed(lambda: m.eval(toks, n_past=0, all_logits=False))
c = timed(lambda: m.eval(toks, n_past=0, all_logits=True))
print(f"7B Q4_0 n_batch 512, default mode: device-resident {a:.2f} ms = {512 / a * 1e3:.0f} tok/s | + last-token logits to the host {b:.2f} ms = {512 / b * 1e3:.0f} tok/s | "
      f"+ all 512 x 32000 logits to the host {c:.2f} ms = {512 / c * 1e3:.0f} tok/s")
