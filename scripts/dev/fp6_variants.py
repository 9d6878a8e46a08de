"""time the i8 and fp6 forms of the prefill GEMM of the library FASTLLAMA_HIP_LIB names (variant builds: timing only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
L.fl_debug_set(3, 1)
def tm(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, K, N) in [(12288, 4096, 512), (22016, 4096, 512), (4096, 11008, 512)]:
    W = ops.QTensor(2, synth.synth_q4(M, K, 2, 3), M, K)
    a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
    out = []
    for cfg in (101, 201, 106, 206, 116, 216):
        L.fl_debug_set(0, cfg)
        out.append(f"{cfg}:{tm(lambda: ops.mul_mat_q(W, a)):7.1f}")
    print(os.environ.get("FASTLLAMA_HIP_LIB", "default")[-20:], M, K, " ".join(out), flush=True)
    W.free()
