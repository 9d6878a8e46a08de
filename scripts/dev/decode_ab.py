"""Decode A/B inside one process: python scripts/dev/decode_ab.py [steps] [n_past] [model] [qtype] [reps] -- tok/s of the reference-order decode, hipGraph
replay, alternating between: head and round5 (without the one-wave-per-row-group form: round 5's kernels).  AB=head selects the variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
past = int(sys.argv[2]) if len(sys.argv) > 2 else 128
name = sys.argv[3] if len(sys.argv) > 3 else "7B"
QT = int(sys.argv[4]) if len(sys.argv) > 4 else 2
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
cfg = dict(synth.MODELS[name])
n_ctx = max(1024, (past + steps + 8 + 511) // 512 * 512)
m = FlModel(cfg, QT, synth.synth_model_tensors(cfg, QT), n_ctx=n_ctx, max_batch=512)
L = hip.load()
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
m.eval_nocopy(toks, 0)
t1 = toks[:1].copy()
def run(tag):
    for i in range(3):
        m.eval_nocopy(t1, past + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.eval_nocopy(t1, past + 3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"[{tag}] {name} qtype={QT} n_past={past}: {dt*1e3:.3f} ms/token  {1/dt:.1f} tok/s", flush=True)
# variants: name -> (fl_debug_set(6, .) value, fl_model_set_graph mode)
VARS = {"head": (-1, 1), "no_kv_prefetch": (-1, 1 | 512), "round5": (1 << 30, 1), "stream_all": (1, 1), "stream300": (300, 1), "no_helpers": (-1, 1, 0)}
names = os.environ.get("AB", "head,round5").split(",")
for r in range(reps):
    for tag in names:
        v, mode = VARS[tag][:2]
        L.fl_debug_set(6, v)
        L.fl_debug_set(9, VARS[tag][2] if len(VARS[tag]) > 2 else 1)
        L.fl_model_set_graph(m.h, mode ^ 2)   # (the fuse bit flips twice: the captured graph is dropped and the next eval captures the selected kernels)
        L.fl_model_set_graph(m.h, mode)
        run(tag)
