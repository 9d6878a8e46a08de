"""The slow decode pass behind a burst of prefill evals (profiles/r05_bench_notes.md: a whole 64-step pass now and then ~40 % slow): where does it
come from?  python scripts/dev/decode_transient.py [bursts] -- per burst: B prefill evals (B = 0, 5, 20, 40), then four decode passes of 48 steps
back to back, each with the clocks / power / temperature the driver reports (sysfs) right before and after, in both modes; then the same with a
0.5 s idle gap between the burst and the first pass, and with graph replay off."""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
from harness import synth
from harness.flmodel import FlModel

def sysfs():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            try:
                cur = [l.split(":")[1].strip().rstrip("*").strip() for l in open(os.path.join(card, f)) if l.strip().endswith("*")]
                out[f[7:]] = cur[0] if cur else "?"
            except Exception:
                pass
        for hw in glob.glob(os.path.join(card, "hwmon/hwmon*")):
            for f, k, sc in (("power1_average", "W", 1e6), ("power1_input", "W", 1e6), ("temp1_input", "C", 1e3), ("temp2_input", "Cj", 1e3), ("freq1_input", "sclk_hw", 1e6)):
                try:
                    out[k] = round(int(open(os.path.join(hw, f)).read()) / sc, 1)
                except Exception:
                    pass
        break
    return out

cfg = dict(synth.MODELS["7B"])
m = FlModel(cfg, 2, synth.synth_model_tensors(cfg, 2), n_ctx=2048, max_batch=512)
L = hip.load()
toks = np.random.default_rng(0).integers(3, 259, 512).astype(np.int32)
t1 = toks[:1].copy()
lg = np.empty(cfg["n_vocab"], dtype=np.float32)
m.eval_nocopy(toks, 0)
for i in range(4):
    m.eval_nocopy(t1, 128 + i)
torch.cuda.synchronize()
print("sysfs fields:", sysfs(), flush=True)

def dpass(n=48, host=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if host:
            m.eval_last_logits(t1, 128 + i, lg)
        else:
            m.eval_nocopy(t1, 128 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def burst(B):
    for i in range(B):
        m.eval_nocopy(toks, 0)
    torch.cuda.synchronize()

for exact in (True, False):
    m.set_exact(exact)
    burst(2); dpass(8)
    for gap in (0.0, 0.5):
        for B in (0, 5, 20, 40):
            t0 = time.perf_counter(); burst(B); tb = time.perf_counter() - t0
            if gap:
                time.sleep(gap)
            s0 = sysfs()
            ps = []
            for k in range(4):
                ps.append(dpass(48, host=(k % 2 == 1)))
            s1 = sysfs()
            print(f"mode={'exact' if exact else 'fast'} burst={B:2d} ({tb*1e3:7.1f} ms) gap={gap:.1f}s  passes ms/token: " + " ".join(f"{p:.3f}" for p in ps) + f"   before {s0}  after {s1}", flush=True)
# per-step times of one pass right behind a long burst: is it the whole pass, or a ramp inside it?
m.set_exact(True)
burst(40)
ts = []
for i in range(96):
    t0 = time.perf_counter(); m.eval_nocopy(t1, 128 + i); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms behind a 40-eval burst (synchronised per step):", " ".join(f"{t:.2f}" for t in ts), flush=True)
