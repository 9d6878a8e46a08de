"""Decode GEMV with the norm prologue, isolated, for each wave count: python scripts/gemv_probe.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
for (M, K) in [(12288, 4096), (4096, 4096), (32000, 4096)]:
    Ws = [ops.QTensor(2, synth.synth_q4(M, K, 2, i), M, K) for i in range(12)]
    x = torch.randn(1, K, device="cuda"); nw = torch.ones(K, device="cuda"); y = torch.empty(M, device="cuda")
    g = torch.cuda.CUDAGraph()
    res = []
    for force in (0, 4, 8, 16):
        L.fl_debug_set(1, force)
        def run(i): hip.check(L.fl_debug_gemv_norm(Ws[i % len(Ws)].handle, x.data_ptr(), nw.data_ptr(), None, y.data_ptr(), None))
        for i in range(12): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(240): run(i)
        e1.record(); torch.cuda.synchronize()
        res.append((force, e0.elapsed_time(e1) / 240 * 1e3))
    L.fl_debug_set(1, 0)
    print(f"gemv_norm {M}x{K}: " + "  ".join(f"nw={f or 'auto'}: {t:.2f}us" for f, t in res))
    for W in Ws: W.free()
