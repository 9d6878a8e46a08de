"""Per-workgroup clocks of one GEMM launch (needs a -DG32_TIMING build): python scripts/dev/g32_timeline.py [M] [K] [cfg]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip, ops
from harness import synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 106
N = 512
L = hip.load(); hip.require_device(0)
L.fl_debug_set(0, cfg)
W = ops.QTensor(2, synth.synth_q4(M, K, 2, 1), M, K)
a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
y = torch.empty(N, M, device="cuda")
for _ in range(5):
    ops.mul_mat_q(W, a, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.mul_mat_q(W, a, out=y); e1.record(); torch.cuda.synchronize()
nwg = (M // 128) * (N // (64 if cfg in (100, 106, 102, 103, 108, 116) else 32))
n = min(nwg, 4096)
buf = (C.c_longlong * (n * 8))()
lib = C.CDLL(hip.LIB_PATH)
lib.fl_debug_g32_timing.argtypes = [C.c_void_p, C.c_int]
assert lib.fl_debug_g32_timing(buf, n) == 0
raw = np.array(buf[:]).reshape(n, 8)
t = raw[:, :5].astype(np.float64) * 10e-3     # us (100 MHz clock)
t0 = t[:, 0].min()
t -= t0
print(f"M={M} K={K} cfg={cfg}: {nwg} workgroups, launch {e0.elapsed_time(e1)*1e3:.1f} us by events")
def q(v): return f"min {v.min():6.2f}  p10 {np.percentile(v,10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}"
print("start (after the first workgroup's start) ", q(t[:, 0]))
print("prologue issue        (stamp1 - stamp0)   ", q(t[:, 1] - t[:, 0]))
print("first data + barrier  (stamp2 - stamp1)   ", q(t[:, 2] - t[:, 1]))
print("main loop             (stamp3 - stamp2)   ", q(t[:, 3] - t[:, 2]))
print("drain + epilogue      (stamp4 - stamp3)   ", q(t[:, 4] - t[:, 3]))
print("end (after the first workgroup's start)   ", q(t[:, 4]))
hw = raw[:, 5] & 0xFFFF
import collections
print("HW_ID.WAVE_ID of wave 0 / wave 1:", dict(collections.Counter((raw[:, 5] & 15).tolist())), dict(collections.Counter((raw[:, 6] & 15).tolist())),
      " SIMD_ID of wave 0 / 1:", dict(collections.Counter(((raw[:, 5] >> 4) & 3).tolist())), dict(collections.Counter(((raw[:, 6] >> 4) & 3).tolist())))
