"""Per-kernel timing of the single-token matmuls, fast vs exact, at 7B shapes (development aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
qt = int(os.environ.get("QT", "2"))
s = np.empty(1 << 16, np.uint16); L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p))
sd = torch.from_numpy(s.view(np.int16)).cuda()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
E, F = 4096, 11008
cases = []
Wqkv = ops.QTensor(qt, synth.synth_q4(3 * E, E, qt, 1), 3 * E, E)
Wo = ops.QTensor(qt, synth.synth_q4(E, E, qt, 2), E, E)
W13 = ops.QTensor(qt, synth.synth_q4(2 * F, E, qt, 3), 2 * F, E)
W2 = ops.QTensor(qt, synth.synth_q4(E, F, qt, 4), E, F)
Wlm = ops.QTensor(qt, synth.synth_q4(32000, E, qt, 5), 32000, E)
x = torch.randn(1, E, device="cuda"); nw = torch.ones(E, device="cuda"); xf = torch.randn(1, F, device="cuda")
y = torch.empty(32000, device="cuda"); act = torch.empty(F, device="cuda")
a = ops.QAct(1, E).quantize(x, layout=1)
null = lambda t: None
for name, fn in [
    ("wqkv norm-gemv 12288x4096", lambda: L.fl_debug_gemv_norm(Wqkv.handle, x.data_ptr(), nw.data_ptr(), None, y.data_ptr(), None)),
    ("lm-head norm-gemv 32000x4096", lambda: L.fl_debug_gemv_norm(Wlm.handle, x.data_ptr(), nw.data_ptr(), None, y.data_ptr(), None)),
    ("wo gemv 4096x4096", None),
    ("w13 pair norm-gemv 22016x4096", lambda: L.fl_debug_gemv_norm_silu(W13.handle, x.data_ptr(), nw.data_ptr(), sd.data_ptr(), act.data_ptr(), None)),
    ("w2 quant-gemv 4096x11008", lambda: L.fl_debug_gemv_quant(W2.handle, xf.data_ptr(), y.data_ptr(), None, None)),
]:
    res = []
    for exact in (0, 1):
        L.fl_debug_set(2, exact)
        if fn is None:
            f = (lambda: L.fl_debug_mul_mat_q(Wo.handle, a.handle, y.data_ptr(), E, 3 if exact else 2, None))
        else:
            f = fn
        res.append(timeit(f))
    L.fl_debug_set(2, 0)
    print(f"{name:34s} fast {res[0]:7.2f} us   exact {res[1]:7.2f} us   x{res[1]/res[0]:.2f}", flush=True)
