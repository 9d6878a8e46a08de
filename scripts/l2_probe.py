"""Is the reference-order decode GEMV faster when its weights sit in the XCDs' L2 (same tensor, same grid, launched back to back)?
python scripts/l2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
for (M, K) in [(4096, 4096), (4096, 11008), (12288, 4096)]:
    Ws = [ops.QTensor(2, synth.synth_q4(M, K, 2, i), M, K) for i in range(24)]      # 24 tensors: beyond L2 (32 MB) and, but for wo, the 256 MB of MALL
    x = torch.randn(1, K, device="cuda"); a = ops.QAct(1, K).quantize(x)
    y = torch.empty(1, M, device="cuda")
    for which, name in ((3, "reference order"), (None, "fast")):
        L.fl_set_op_mode(0 if which is None else -1)
        def timed(seq, reps=96):
            g = torch.cuda.CUDAGraph()
            for W in seq: ops.mul_mat_q(W, a, which=which, out=y)       # (derived copies are built on first use: not inside the capture)
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    for i in range(reps): ops.mul_mat_q(seq[i % len(seq)], a, which=which, out=y)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        cold, warm = timed(Ws), timed(Ws[:1])
        print(f"GEMV {M}x{K} {name}: round-robin over 24 tensors {cold:.2f} us   same tensor {warm:.2f} us", flush=True)
    L.fl_set_op_mode(-1)
    for W in Ws: W.free()
