"""Does overlapping two half-size prefills (two streams) beat one full-size prefill?  Two models, two host threads.
python scripts/two_stream_probe.py [steps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = dict(synth.MODELS["7B"])

ms = [FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=1024, max_batch=512) for _ in range(2)]
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)

def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0

def single(n):
    def f():
        for _ in range(steps): ms[0].eval_nocopy(toks[:n], 0)
    return f

def dual(n):
    def f():
        bar = threading.Barrier(2)
        def w(m):
            bar.wait()
            for _ in range(steps): m.eval_nocopy(toks[:n], 0)
        th = [threading.Thread(target=w, args=(m,)) for m in ms]
        [t.start() for t in th]; [t.join() for t in th]
    return f

t512 = timed(single(512)); print(f"one stream,  512 tokens: {t512/steps*1e3:7.2f} ms/step  {512*steps/t512:8.0f} tok/s", flush=True)
t256 = timed(single(256)); print(f"one stream,  256 tokens: {t256/steps*1e3:7.2f} ms/step  {256*steps/t256:8.0f} tok/s", flush=True)
d256 = timed(dual(256));   print(f"two streams, 256 tokens each: {d256/steps*1e3:7.2f} ms/step-pair  {512*steps/d256:8.0f} tok/s", flush=True)
d512 = timed(dual(512));   print(f"two streams, 512 tokens each: {d512/steps*1e3:7.2f} ms/step-pair  {1024*steps/d512:8.0f} tok/s", flush=True)
