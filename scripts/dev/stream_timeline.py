"""Per-workgroup timeline of the one-wave-per-row-group kernel (gemv1_q4_exact_stream.hip) inside a decode token, hipGraph replay.
Needs a -DLLC_TIMING build:  ALL_FLAGS=-DLLC_TIMING TAG=tl OUT=gpurun_variants/libtl.so bash scripts/dev/fastbuild.sh
Run:  FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/stream_timeline.py [model=7B] [n_past=128]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel
name = sys.argv[1] if len(sys.argv) > 1 else "7B"
past = int(sys.argv[2]) if len(sys.argv) > 2 else 128
qtype = int(os.environ.get("FL_QTYPE", "2"))
L = hip.load()
lib = C.CDLL(hip.LIB_PATH)
lib.fl_debug_stream_timeline.argtypes = [C.c_void_p, C.c_int, C.c_int]
cfg = dict(synth.MODELS[name])
m = FlModel(cfg, qtype, synth.synth_model_tensors(cfg, qtype), n_ctx=1024, max_batch=512)
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
m.eval_nocopy(toks[:max(past, 1)], 0)
t1 = toks[:1].copy()
for i in range(4):
    m.eval_nocopy(t1, past + i)
torch.cuda.synchronize()
lib.fl_debug_stream_timeline(None, 0, 1)
m.eval_nocopy(t1, past + 4)
torch.cuda.synchronize()
cap = 1 << 16
buf = np.zeros((cap, 12), np.int64)
n = lib.fl_debug_stream_timeline(buf.ctypes.data_as(C.c_void_p), cap, 0)
rec = buf[:n]
rec = rec[np.argsort(rec[:, 0], kind="stable")]
kid = rec[:, 9] >> 32
cuts = np.flatnonzero(np.diff(kid) != 0) + 1
runs = np.split(np.arange(len(rec)), cuts)
rows = {}
for r in runs:
    t = rec[r][:, :9].astype(np.float64) * 0.01      # us
    k = int(kid[r[0]])
    t0 = t[:, 0].min()
    d = dict(wgs=len(r), dur=t[:, 8].max() - t0, ramp=t[:, 0].max() - t0, issued=np.median(t[:, 1] - t[:, 0]), pro=np.median(t[:, 2] - t[:, 0]),
             pro_max=(t[:, 2] - t0).max(), loop_med=np.median(t[:, 3:7].max(axis=1) - t[:, 2]), loop_min=(t[:, 3:7].min(axis=1) - t[:, 2]).min(),
             loop_max=(t[:, 3:7].max(axis=1) - t[:, 2]).max(), last_loop_end=(t[:, 3:7].max(axis=1)).max() - t0,
             wave_spread=np.median(t[:, 3:7].max(axis=1) - t[:, 3:7].min(axis=1)), epi=np.median(t[:, 7] - t[:, 3:7].max(axis=1)), life=np.median(t[:, 8] - t[:, 0]), life_max=(t[:, 8] - t[:, 0]).max())
    rows.setdefault((k, len(r)), []).append(d)
print(f"{name} qtype {qtype} n_past {past}: one decode token, hipGraph replay; medians over the launches of a kind (us)")
print("| kernel id (PRO*100+EPI*10+U16) | workgroups | launches | duration | dispatch ramp | entry -> loads issued | entry -> prologue done (median / last, from launch start) | prologue -> wave loop end (min / median / max) | last loop end from launch start | spread of a workgroup's waves | loop end -> stored | workgroup lifetime median / max |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for (k, w), ds in sorted(rows.items()):
    med = lambda key: float(np.median([d[key] for d in ds]))
    print(f"| {k} | {w} | {len(ds)} | {med('dur'):.2f} | {med('ramp'):.2f} | {med('issued'):.2f} | {med('pro'):.2f} / {med('pro_max'):.2f} | {med('loop_min'):.2f} / {med('loop_med'):.2f} / {med('loop_max'):.2f} | {med('last_loop_end'):.2f} | {med('wave_spread'):.2f} | {med('epi'):.2f} | {med('life'):.2f} / {med('life_max'):.2f} |")
# per-workgroup records of ONE launch of each kind (the launch in the middle of the token), for offline study: gpurun_out/stream_wg_<id>.csv
os.makedirs("gpurun_out", exist_ok=True)
seen = {}
for r in runs:
    k = int(kid[r[0]]); seen.setdefault((k, len(r)), []).append(r)
for (k, w), rs in seen.items():
    r = rs[len(rs) // 2]
    t = rec[r][:, :9].astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    wg = rec[r][:, 9] & 0xffffffff
    with open(f"gpurun_out/stream_wg_{k}_{w}.csv", "w") as f:
        f.write("wg,entry,issued,prologue_done,loop_end_w0,loop_end_w1,loop_end_w2,loop_end_w3,stored,end\n")
        for i in np.argsort(wg):
            f.write(f"{int(wg[i])}," + ",".join(f"{v - t0:.2f}" for v in t[i]) + "\n")
