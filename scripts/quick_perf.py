"""Quick per-shape timing of fl_mul_mat_q (COMPUTE) and fl_quantize_q8 (INIT) on LLaMA-7B shapes.
Development aid (not the bench contract): python scripts/quick_perf.py [N ...]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastllama_amd import hip, ops

Ns = [int(a) for a in sys.argv[1:]] or [1, 512]
ABL = int(os.environ.get("FL_ABL", "0"))
L = hip.load()
hip.require_device(0)
L.fl_debug_set(0, int(os.environ.get('FL_CFG', '-1')))
shapes = [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096)]
for qt in (2, 3):
    for (M, K) in shapes:
        bs = 20 if qt == 2 else 24
        blocks = torch.randint(0, 255, (M, K // 32 * bs), dtype=torch.uint8, device="cuda")
        # plausible scales: overwrite the f32 fields with small positive numbers
        v = blocks.view(M, K // 32, bs)
        sc = (torch.rand(M, K // 32, device="cuda") * 0.01 + 0.001).view(torch.uint8) if False else None
        f = (torch.rand(M, K // 32, 1, device="cuda") * 0.01 + 0.001)
        v[:, :, 0:4] = f.view(torch.uint8).view(M, K // 32, 4)
        if qt == 3:
            v[:, :, 4:8] = (-f * 7).view(torch.uint8).view(M, K // 32, 4)
        W = ops.QTensor(qt, blocks, M, K)
        for N in Ns:
            x = torch.randn(N, K, device="cuda")
            a = ops.QAct(N, K)
            y = torch.empty(N, M, device="cuda")
            a.quantize(x)
            ops.mul_mat_q(W, a, out=y)
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            reps = 20
            e0.record()
            for _ in range(reps):
                a.quantize(x)
            e1.record()
            for _ in range(reps):
                ops.mul_mat_q(W, a, out=y)
            e2.record()
            torch.cuda.synchronize()
            tq, tm = e0.elapsed_time(e1) / reps, e1.elapsed_time(e2) / reps
            wbytes = M * K // 32 * bs
            flops = 2.0 * M * K * N
            print(f"q4_{qt-2} M={M:6d} K={K:6d} N={N:4d}  quant {tq*1e3:8.1f} us   matmul {tm*1e3:9.1f} us  "
                  f"{wbytes/tm/1e6:8.1f} GB/s(W)  {flops/tm/1e9:9.1f} GOP/s", flush=True)
        W.free()
