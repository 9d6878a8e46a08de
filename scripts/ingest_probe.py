"""A long prompt through fl_model_ingest (two chunks in flight) against fl_model_eval chunk by chunk: python scripts/ingest_probe.py [tokens] [chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=max(2048, total), max_batch=chunk)
toks = np.random.default_rng(0).integers(3, 259, total).astype(np.int32)

def seq():
    for i in range(0, total, chunk):
        m.eval_nocopy(toks[i:i + chunk], i)

def pipe():
    m.ingest(toks, chunk, want_logits=False)

for name, f in (("chunk by chunk", seq), ("pipelined", pipe), ("chunk by chunk", seq), ("pipelined", pipe)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{total} tokens in chunks of {chunk}, {name:15s}: {dt*1e3:8.2f} ms  {total/dt:8.0f} tok/s", flush=True)
