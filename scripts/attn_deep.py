"""Time the three forms of prefill attention (LDS-resident, key-tiled, three kernels) over n_past: python scripts/attn_deep.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastllama_amd import hip
H, D, n_ctx, N = 32, 128, 2048, 512
E = H * D
L = hip.load(); hip.require_device(0)
qkv = torch.randn(N, 3 * E, device="cuda"); kc = torch.randn(n_ctx, E, device="cuda"); vc = torch.randn(E, n_ctx, device="cuda")
e = np.empty(1 << 16, np.uint16); L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
ed = torch.from_numpy(e.view(np.int16)).cuda(); ao = torch.empty(N, E, device="cuda")
att = torch.empty(H, N, n_ctx, device="cuda")
scale = 0.08838834764831845

def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for P0 in ([int(a) for a in sys.argv[1:]] or [0, 256, 512, 1024, 1536]):
    P = P0 + N
    def one():
        hip.check(L.fl_debug_prefill_attention(qkv.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kc.data_ptr(), vc.data_ptr(), ed.data_ptr(), scale, ao.data_ptr(), E, None, None))
    def three():
        hip.check(L.fl_debug_gemm_f32_abt(qkv.data_ptr(), 3 * E, D, kc.data_ptr(), E, D, att.data_ptr(), n_ctx, N * n_ctx, N, P, D, H, scale, 1, P0, None))
        hip.check(L.fl_debug_softmax_rows(att.data_ptr(), n_ctx, N * n_ctx, N, P, P0, H, ed.data_ptr(), None))
        hip.check(L.fl_debug_gemm_f32_abt(att.data_ptr(), n_ctx, N * n_ctx, vc.data_ptr(), n_ctx, D * n_ctx, ao.data_ptr(), E, D, N, D, P, H, 1.0, 2, P0, None))
    hip.check(L.fl_debug_prefill_attention_scratch(None, 0, 0))
    t_lds = timeit(one) if P0 == 0 else float("nan")
    hip.check(L.fl_debug_prefill_attention_scratch(att.data_ptr(), n_ctx, N * n_ctx))
    t_deep = timeit(one)
    hip.check(L.fl_debug_prefill_attention_scratch(None, 0, 0))
    t3 = timeit(three)
    flop = 2.0 * 2 * H * D * (N * P0 + N * (N + 32) / 2)
    print(f"N={N} n_past={P0:5d}: LDS-resident {t_lds:7.1f} us   key-tiled {t_deep:7.1f} us ({flop / t_deep / 1e6:5.1f} TFLOP/s f32)   three kernels {t3:7.1f} us", flush=True)

if os.environ.get("FL_PD_TIMING"):      # a -DPA_TIMING build: per-phase clocks of head 0's workgroups in the LAST key-tiled launch
    lib = C.CDLL(hip.LIB_PATH)
    if hasattr(lib, "fl_debug_pd_timing"):
        P0 = int(os.environ["FL_PD_TIMING"])
        hip.check(L.fl_debug_prefill_attention_scratch(att.data_ptr(), n_ctx, N * n_ctx))
        hip.check(L.fl_debug_prefill_attention(qkv.data_ptr(), 3 * E, D, H, N, P0, n_ctx, E, kc.data_ptr(), vc.data_ptr(), ed.data_ptr(), scale, ao.data_ptr(), E, None, None))
        torch.cuda.synchronize()
        buf = (C.c_longlong * 512)()
        lib.fl_debug_pd_timing.argtypes = [C.c_void_p]
        lib.fl_debug_pd_timing(buf)
        t = np.array(buf[:]).reshape(64, 8)[: (N + 31) // 32, :6].astype(np.float64) * 10e-3
        t -= t[:, 0].min()
        print(f"key-tiled launch at n_past {P0}: per-phase clocks of head 0 (us; workgroup y = heaviest block first)")
        for y in range(t.shape[0]):
            d = t[y, 1:] - t[y, :-1]
            print(f"  y={y:2d}: start {t[y,0]:6.2f}  table+Q {d[0]:5.2f}  scores(wave0) {d[1]:6.2f}  wait {d[2]:5.2f}  sums {d[3]:6.2f}  kqv {d[4]:6.2f}  | end {t[y,5]:7.2f}")
        hip.check(L.fl_debug_prefill_attention_scratch(None, 0, 0))
