"""Time every GEMM tile configuration on the LLaMA prefill shapes: python scripts/sweep_cfg.py [N] [qtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastllama_amd import hip, ops
from harness import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
qt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = hip.load(); hip.require_device(0)
shapes = [(4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008), (32000, 4096)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[3:]]
for (M, K) in shapes:
    W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
    x = torch.randn(N, K, device="cuda")
    a = ops.QAct(N, K).quantize(x)
    y = torch.empty(N, (M + 3) // 4 * 4, device="cuda")[:, :M]
    res = []
    for cfg in ([int(v) for v in os.environ['FL_SWEEP_CFGS'].split(',')] if 'FL_SWEEP_CFGS' in os.environ else [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]):
        L.fl_debug_set(0, cfg)
        for _ in range(3):
            ops.mul_mat_q(W, a, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.mul_mat_q(W, a, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append((cfg, ms * 1e3, 2.0 * M * K * N / ms / 1e9))
    print(f"M={M:6d} K={K:6d} N={N}: " + "  ".join(f"c{c}:{us:7.1f}us/{t:5.0f}T" for c, us, t in res), flush=True)
    W.free()
