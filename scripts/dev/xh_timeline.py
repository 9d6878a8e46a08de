"""Per-workgroup clocks of one exact-GEMM launch (needs a -DXH_TIMING build): python scripts/dev/xh_timeline.py [M] [K]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip, ops
from harness import synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = 512
L = hip.load(); hip.require_device(0)
W = ops.QTensor(2, synth.synth_q4(M, K, 2, 1), M, K)
a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"), layout=16)
y = torch.empty(N, M, device="cuda")
for _ in range(5):
    ops.mul_mat_q(W, a, which=5, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.mul_mat_q(W, a, which=5, out=y); e1.record(); torch.cuda.synchronize()
nwg = ((M + 63) // 64) * ((N + 63) // 64)
n = min(nwg, 8192)
buf = (C.c_longlong * (n * 8))()
lib = C.CDLL(hip.LIB_PATH)
lib.fl_debug_xh_timing.argtypes = [C.c_void_p, C.c_int]
assert lib.fl_debug_xh_timing(buf, n) == 0
raw = np.array(buf[:]).reshape(n, 8)
t = raw[:, :6].astype(np.float64) * 10e-3     # us (100 MHz clock)
t -= t[:, 0].min()
print(f"M={M} K={K}: {nwg} workgroups, launch (+ conversion) {e0.elapsed_time(e1)*1e3:.1f} us by events")
def q(v): return f"min {v.min():6.2f}  p10 {np.percentile(v,10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}"
print("start (after the first workgroup's start)  ", q(t[:, 0]))
print("entry -> first K-step landed + barrier      ", q(t[:, 1] - t[:, 0]))
print("K-step 0 (4 blocks)                         ", q(t[:, 2] - t[:, 1]))
print("K-step 1                                    ", q(t[:, 3] - t[:, 2]))
print("K-steps 2..                                 ", q(t[:, 4] - t[:, 3]), " per step", f"{np.median(t[:, 4] - t[:, 3]) / max(1, (K // 128) - 2):.3f}")
print("drain + tree + stores                       ", q(t[:, 5] - t[:, 4]))
print("end (after the first workgroup's start)     ", q(t[:, 5]))
order = np.argsort(t[:, 0])
print("workgroup lifetimes by start order (start, end):", [(round(float(t[i, 0]), 1), round(float(t[i, 5]), 1)) for i in order[:: max(1, n // 12)]])
