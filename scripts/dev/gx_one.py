"""Run the exact-mode GEMM (which = 3) on one shape a few times (PMC / rocprof target): python scripts/dev/gx_one.py qt M K N [reps] [which]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
qt, M, K, N = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
which = int(sys.argv[6]) if len(sys.argv) > 6 else 3
W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"), layout=16 if N >= 2 else None)
y = torch.empty(N, M, device="cuda")
for _ in range(reps):
    ops.mul_mat_q(W, a, which=which, out=y)
torch.cuda.synchronize()
