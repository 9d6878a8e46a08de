"""time chosen configs on the 7B shapes: python scripts/dev/g32_sweep2.py cfg,cfg,... [qt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
cfgs = [int(v) for v in sys.argv[1].split(",")]
qt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = 512
for (M, K) in [(4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008), (32000, 4096)]:
    W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
    a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
    y = torch.empty(N, M, device="cuda")
    res = []
    for cfg in cfgs:
        L.fl_debug_set(0, cfg)
        for _ in range(3):
            ops.mul_mat_q(W, a, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.mul_mat_q(W, a, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        res.append((cfg, ms * 1e3, 2.0 * M * K * N / ms / 1e9))
    print(f"q4_{qt-2} M={M:6d} K={K:6d}: " + "  ".join(f"c{c}:{us:7.1f}us/{t:5.0f}T" for c, us, t in res), flush=True)
    W.free()
