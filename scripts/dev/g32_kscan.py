"""kernel time against K at fixed M x N: the intercept is the per-launch fixed cost.  python scripts/dev/g32_kscan.py [M] [cfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 106
N = 512
L.fl_debug_set(0, cfg)
for K in (256, 512, 1024, 2048, 4096, 8192, 16384):
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
    print(f"M={M} cfg={cfg} K={K:6d}: {us:8.1f} us  {us / (K // 32):7.3f} us/block  {2.0 * M * K * N / us / 1e6:6.0f} TOP/s", flush=True)
    W.free()
