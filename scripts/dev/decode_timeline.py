"""In-graph, per-launch timeline of a reference-order decode token: every workgroup of every launch of ONE token stamps
the 100 MHz wall clock (entry, loads issued, prologue done, first quad summed, last byte summed, chains handed through, end) into a ring;
the launches of a layer line up on one time axis, so the table shows per launch: first wave start -> last workgroup start (dispatch ramp),
first data, last data, end, and the gap to the next launch's first wave -- inside the hipGraph replay and as plain stream launches.

Needs a -DLLC_TIMING build:  ALL_FLAGS=-DLLC_TIMING TAG=tl OUT=gpurun_variants/libtl.so bash scripts/dev/fastbuild.sh
Run:  FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py [model=7B] [n_past=128] > gpurun_out/decode_timeline.md"""
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
for f in ("fl_debug_llc_timeline", "fl_debug_da_timeline", "fl_debug_stream_timeline"):
    getattr(lib, f).argtypes = [C.c_void_p, C.c_int, C.c_int]
cfg = dict(synth.MODELS[name])
m = FlModel(cfg, qtype, synth.synth_model_tensors(cfg, qtype), n_ctx=1024, max_batch=512)
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
m.eval_nocopy(toks[:max(past, 1)], 0)
t1 = toks[:1].copy()
KNAME = {1101: "wq|wk|wv + rms_norm prologue (round 6: one wave per row group)", 1121: "w1|w3 woven + rms_norm -> the Q8_0 operand of w2 (round 6: one wave per row group)",
         104: "wq|wk|wv + rms_norm prologue", 900: "attention (one launch)", 4: "wo (+ residual)", 8: "w2 on the Q8_0 operand w1|w3 wrote (+ residual), 8 slices",
         124: "w1|w3 woven + rms_norm, pair exchange", 114: "w1|w3 woven + rms_norm, one workgroup per pair", 308: "w2 + Q8_0 prologue (+ residual)",
         304: "w2 + Q8_0 prologue (+ residual), 4 slices"}


def fetch(fn, cap, w=8):
    buf = np.zeros((cap, w), np.int64)
    n = fn(buf.ctypes.data_as(C.c_void_p), cap, 0)
    assert n >= 0
    out = np.zeros((n, 12), np.int64)
    out[:, :w] = buf[:n]
    return out


def one_token(graph):
    L.fl_model_set_graph(m.h, graph)
    for i in range(4):
        m.eval_nocopy(t1, past + i)
    torch.cuda.synchronize()
    lib.fl_debug_llc_timeline(None, 0, 1); lib.fl_debug_da_timeline(None, 0, 1); lib.fl_debug_stream_timeline(None, 0, 1)
    m.eval_nocopy(t1, past + 4)
    torch.cuda.synchronize()
    # the one-wave-per-row-group kernel's ring (gemv1_q4_exact_stream.hip: entry, loads issued, prologue done, loop end of waves 0..3, stored, end, id) in the
    # llc record's terms: last byte summed = the last wave's loop end, "chains" = loop end -> stored, first quad summed = prologue done; ids + 1000
    sraw = fetch(lib.fl_debug_stream_timeline, 1 << 16, 12)
    srec = np.zeros((len(sraw), 12), np.int64)
    if len(sraw):
        srec[:, 0:3] = sraw[:, 0:3]
        srec[:, 3] = sraw[:, 3:7].max(axis=1); srec[:, 4] = sraw[:, 7]; srec[:, 5] = sraw[:, 8]; srec[:, 6] = sraw[:, 2]
        srec[:, 7] = ((((sraw[:, 9] >> 32) | 1) + 1000) << 32) | (sraw[:, 9] & 0xffffffff)      # (| 1: the same row whether the launch runs 8 or 16 quads in flight)
    rec = np.concatenate([fetch(lib.fl_debug_llc_timeline, 1 << 17, 12), fetch(lib.fl_debug_da_timeline, 1 << 14, 8), srec])
    rec = rec[np.argsort(rec[:, 0], kind="stable")]
    kid = rec[:, 7] >> 32
    # launches run one after the other on the stream: a launch = a maximal run of one kernel id on the time axis (the lm-head, also 104, follows a 308)
    cuts = np.flatnonzero(np.diff(kid) != 0) + 1
    runs = np.split(np.arange(len(rec)), cuts)
    out = []
    for r in runs:
        t = rec[r][:, :7].astype(np.float64) * 0.01          # us
        k = int(kid[r[0]])
        att = k == 900
        first, last_start = t[:, 0].min(), t[:, 0].max()
        end = (t[:, 6] if att else t[:, 5]).max()
        d = dict(kid=k, wgs=len(r), first=first, ramp=last_start - first, end=end, dur=end - first)
        if att:
            d.update(first_data=np.median(t[:, 2] - t[:, 0]), phases=[float(np.median(t[:, i + 1] - t[:, i])) for i in range(6)])
        else:
            d.update(issue=float(np.median(t[:, 1] - t[:, 0])), prologue=float(np.median(t[:, 2] - t[:, 0])), first_data=float(np.median(t[:, 6] - t[:, 0])),
                     last_data_med=float(np.median(t[:, 3] - t[:, 0])), last_data=float(t[:, 3].max() - first), chains=float(np.median(t[:, 4] - t[:, 3])),
                     tail=float(end - t[:, 3].max()), life_med=float(np.median(t[:, 5] - t[:, 0])), life_max=float((t[:, 5] - t[:, 0]).max()))
            tp = rec[r][:, 8:11].astype(np.float64) * 0.01
            if tp[:, 0].max() > 0:                            # rms_norm prologue: x squared | first barrier | scale + norm weights there (from entry)
                d["pro"] = [float(np.median(tp[:, j] - t[:, 0])) for j in range(3)]
        out.append(d)
    for a, b in zip(out, out[1:]):
        a["gap"] = b["first"] - a["end"]
    return out


for graph, what in ((1, "hipGraph replay"), (0, "plain stream launches")):
    runs = one_token(graph)
    tok_us = runs[-1]["end"] - runs[0]["first"]
    print(f"\n## {name} Q4_{qtype - 2}, n_past {past + 4}, reference-order decode, {what}: {len(runs)} launches, {tok_us:.1f} us from the first wave of the token to its last store\n")
    print("| launch (median over layers 2..) | workgroups | duration: first wave -> last end | dispatch ramp (first -> last workgroup start) | entry -> loads issued | entry -> prologue done | entry -> first quad summed | entry -> last byte summed (median / last workgroup, from launch start) | chains | last byte -> launch end | workgroup lifetime median / max | gap to the next launch's first wave |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    seq = [1101, 104, 900, 4, 1121, 124, 114, 8, 308, 304]
    for k in seq:
        rs = [r for r in runs[5:-1] if r["kid"] == k and "gap" in r]
        if not rs:
            continue
        med = lambda key: float(np.median([r[key] for r in rs]))
        if k == 900:
            ph = np.median(np.array([r["phases"] for r in rs]), axis=0)
            print(f"| {KNAME[k]} | {rs[0]['wgs']} | {med('dur'):.2f} | {med('ramp'):.2f} | requests issued {ph[0]:.2f} | rope + K/V stores {ph[1]:.2f} | scores + max {ph[2]:.2f} | soft_max {ph[3]:.2f} | P.V {ph[4]:.2f} | Q8_0 store {ph[5]:.2f} | - | {med('gap'):.2f} |")
        else:
            if "pro" in rs[0]:
                pm = np.median(np.array([r["pro"] for r in rs]), axis=0)
                print(f"|   ... its rms_norm prologue, from entry: x arrived and squared {pm[0]:.2f}, first barrier passed {pm[1]:.2f}, scale known + norm weights arrived {pm[2]:.2f} | | | | | | | | | | | |")
            print(f"| {KNAME.get(k, k)} | {rs[0]['wgs']} | {med('dur'):.2f} | {med('ramp'):.2f} | {med('issue'):.2f} | {med('prologue'):.2f} | {med('first_data'):.2f} | {med('last_data_med'):.2f} / {med('last_data'):.2f} | {med('chains'):.2f} | {med('tail'):.2f} | {med('life_med'):.2f} / {med('life_max'):.2f} | {med('gap'):.2f} |")
    lay = [r for r in runs if r["kid"] in (104, 1101)]
    if len(lay) > 3:
        per_layer = np.diff([r["first"] for r in lay[:-1]])
        print(f"\nlayer period (first wave of wq|wk|wv to the next layer's): median {np.median(per_layer):.2f} us, min {per_layer.min():.2f}, max {per_layer.max():.2f}; "
              f"sum of the five launches' durations {sum(float(np.median([r['dur'] for r in runs[5:-1] if r['kid'] == k])) for k in set(r['kid'] for r in runs[5:-1])):.2f} us, "
              f"sum of gaps {sum(float(np.median([r['gap'] for r in runs[5:-1] if r['kid'] == k and 'gap' in r])) for k in set(r['kid'] for r in runs[5:-1])):.2f} us")
    lm = runs[-1]
    print(f"lm-head: {lm['wgs']} workgroups, {lm['dur']:.2f} us")
# throughput of the same build (stamps cost a little): graph replay
L.fl_model_set_graph(m.h, 1)
import time
for i in range(3):
    m.eval_nocopy(t1, past + i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(64):
    m.eval_nocopy(t1, past + 3 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 64
print(f"\n(this build, with its stamps: {dt * 1e3:.3f} ms/token = {1 / dt:.1f} tok/s)")
