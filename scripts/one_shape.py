"""Run one mul_mat_q shape a few times (for rocprofv3 PMC runs): python scripts/one_shape.py M K N [qtype] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastllama_amd import hip, ops
from harness import synth
M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
qt = int(sys.argv[4]) if len(sys.argv) > 4 else 2
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
L = hip.load(); hip.require_device(0)
L.fl_debug_set(0, int(os.environ.get("FL_CFG", "-1")))
W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
x = torch.randn(N, K, device="cuda")
a = ops.QAct(N, K).quantize(x)
y = torch.empty(N, (M + 3) // 4 * 4, device="cuda")[:, :M]
for _ in range(reps):
    ops.mul_mat_q(W, a, out=y)
torch.cuda.synchronize()
print("done", float(y.abs().max()))
