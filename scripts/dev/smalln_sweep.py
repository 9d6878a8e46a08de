"""Which tile configuration is fastest when the batch is small (a short prompt)?  N = 16 ... 128, LLaMA-7B shapes, all configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
def t(f, n=20):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cfgs = list(range(14)) + [101, 105, 104]
for M, K in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
    W = ops.QTensor(2, synth.synth_q4(M, K, 2, 1), M, K)
    for N in (16, 32, 64, 128):
        a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
        y = torch.empty(N, M, device="cuda")
        res = []
        for c in cfgs:
            L.fl_debug_set(0, c)
            try:
                res.append((t(lambda: ops.mul_mat_q(W, a, out=y)), c))
            except Exception:
                pass
        res.sort()
        base = [r for r in res if r[1] == 101][0][0]
        print(f"{M}x{K} N={N:3d}: cfg101 {base:6.1f} us | best " + "  ".join(f"cfg{c} {u:.1f}" for u, c in res[:4]), flush=True)
    W.free()
L.fl_debug_set(0, -1)
