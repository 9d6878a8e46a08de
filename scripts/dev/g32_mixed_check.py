"""cfg 116 (128x64 tiles for whole rounds of workgroups + 128x32 tiles for the rest, one launch) against cfg 106: bits and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
N = 512
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for qt in (2, 3):
    for M, K in ((22016, 4096), (12288, 4096), (11008, 4096), (4096, 11008), (32000, 4096), (13824 * 2, 5120), (22016 // 2, 4096), (4000, 4096)):
        W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
        a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
        ys = {}
        line = f"Q4_{qt-2} {M}x{K}:"
        for cfg in (106, 116, 101):
            L.fl_debug_set(0, cfg)
            y = torch.empty(N, M, device="cuda")
            us = t(lambda: ops.mul_mat_q(W, a, out=y))
            ys[cfg] = y
            line += f"  cfg{cfg} {us:7.1f} us"
        ok = torch.equal(ys[106], ys[116]) and torch.equal(ys[106], ys[101])
        print(line, " identical" if ok else "  MISMATCH", flush=True)
        W.free()
L.fl_debug_set(0, 0)
