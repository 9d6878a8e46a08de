"""Run one (cfg, qtype, M, K, N) GEMM a few times (PMC / rocprof target): python scripts/dev/g32_one.py cfg qt M K N [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
cfg, qt, M, K, N = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
x = torch.randn(N, K, device="cuda")
a = ops.QAct(N, K).quantize(x)
y = torch.empty(N, M, device="cuda")
L.fl_debug_set(0, cfg)
for _ in range(reps):
    ops.mul_mat_q(W, a, out=y)
torch.cuda.synchronize()
