"""Stress of the two-workgroup w1|w3 pairing (gemv1_q4_exact_llc.hip, PAIR = 2): T decode steps of LLaMA-7B (hipGraph replay), logits of
every step written to argv[1].  Run once as is and once with FL_EXACT_PAIR=1 (the one-workgroup form; 2 = two workgroups per pair), then compare the files bit for bit:
python scripts/dev/pair_stress.py out.npy [T] [qtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harness import synth
from harness.flmodel import FlModel
out, T, qt = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 384, int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, qt, synth.synth_model_tensors(cfg, qt, seed=1234), n_ctx=1024, max_batch=64)
toks = np.random.default_rng(7).integers(3, 259, 64).astype(np.int32)
m.eval(toks, n_past=0)
res = np.empty((T, cfg["n_vocab"]), np.float32)
tok = 5
for i in range(T):
    lg = m.eval([tok], n_past=64 + i)
    res[i] = lg[-1]
    tok = int(np.argmax(lg[-1])) % 30000 + 3
np.save(out, res)
print("wrote", out, res.shape, float(np.abs(res).max()))
