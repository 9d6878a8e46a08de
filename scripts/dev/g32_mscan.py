"""kernel time against M (number of workgroups) at fixed K, N: where does a second round of workgroups start?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 106
K, N = 4096, 512
L.fl_debug_set(0, cfg)
for M in (4096, 6144, 8192, 9216, 10240, 11264, 12288, 13312, 14336, 16384, 20480, 24576):
    W = ops.QTensor(2, synth.synth_q4(M, K, 2, 1), M, K)
    a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
    y = torch.empty(N, M, device="cuda")
    for _ in range(3):
        ops.mul_mat_q(W, a, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.mul_mat_q(W, a, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    wgs = (M // 128) * (N // (64 if cfg in (100, 106, 102, 103, 108) else 32))
    print(f"cfg={cfg} M={M:6d} WGs={wgs:5d}: {us:8.1f} us  {2.0 * M * K * N / us / 1e6:6.0f} TOP/s", flush=True)
    W.free()
