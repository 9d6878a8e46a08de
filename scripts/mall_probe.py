"""Is a GEMV faster when its weights were just read (Infinity Cache / L2 warm)?  python scripts/mall_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
for (M, K) in [(22016, 4096), (44032, 8192), (12288, 4096)]:
    Ws = [ops.QTensor(2, synth.synth_q4(M, K, 2, i), M, K) for i in range(10)]      # 10 tensors: well over the 256 MB of cache
    x = torch.randn(1, K, device="cuda"); a = ops.QAct(1, K).quantize(x)
    y = torch.empty(1, M, device="cuda")
    def timed(seq, reps=50):
        for W in seq[:3]: ops.mul_mat_q(W, a, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): ops.mul_mat_q(seq[i % len(seq)], a, out=y)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    cold, warm = timed(Ws), timed(Ws[:1])
    mb = M * K / 32 * 20 / 1e6
    print(f"GEMV {M}x{K} ({mb:.1f} MB): cold (10 tensors round-robin) {cold:.1f} us = {mb / cold:.2f} TB/s   warm (same tensor) {warm:.1f} us = {mb / warm:.2f} TB/s   (python launch gaps included)")
    for W in Ws: W.free()
