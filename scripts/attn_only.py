"""Time the fused prefill attention kernel alone: python scripts/attn_only.py [N] [n_past]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastllama_amd import hip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H, D, n_ctx = 32, 128, 1024
E = H * D
L = hip.load(); hip.require_device(0)
qkv = torch.randn(N, 3 * E, device="cuda"); kc = torch.randn(n_ctx, E, device="cuda"); vc = torch.randn(E, n_ctx, device="cuda")
e = np.empty(1 << 16, np.uint16); L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
ed = torch.from_numpy(e.view(np.int16)).cuda(); ao = torch.empty(N, E, device="cuda")
def run():
    hip.check(L.fl_debug_prefill_attention(qkv.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kc.data_ptr(), vc.data_ptr(), ed.data_ptr(), 0.0883883, ao.data_ptr(), E, None, None))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"prefill_attention N={N} n_past={P0}: {e0.elapsed_time(e1)/20*1e3:.1f} us")

if hasattr(L, "fl_debug_pa_timing"):
    buf = (C.c_longlong * 512)()
    L.fl_debug_pa_timing.argtypes = [C.c_void_p]
    L.fl_debug_pa_timing(buf)
    t = np.array(buf[:]).reshape(64, 8)
    for wg in range(min(8, (N + 63) // 64)):
        d = (t[wg, 1:7] - t[wg, 0:6]) * 10.0 / 1e3   # wall_clock64: 100 MHz
        print(f"  wg{wg}: table {d[0]:.2f}  scores(wave0) {d[1]:.2f}  wait {d[2]:.2f}  softmax(wave0) {d[3]:.2f}  wait {d[4]:.2f}  pv(wave0) {d[5]:.2f} us")
