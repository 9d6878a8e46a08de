"""Structured-input diagnosis of one gemm32 configuration: python scripts/dev/g32_diag.py cfg qtype M K N"""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip, ops
L = hip.load(); hip.require_device(0)
cfg, qt, M, K, N = (int(v) for v in sys.argv[1:6])
KB = K // 32
bs = 20 if qt == 2 else 24
blk = np.zeros((M, KB, bs), dtype=np.uint8)
dw = (1.0 + (np.arange(M) % 7)[:, None] * 0.25 + np.zeros((1, KB))).astype(np.float32)      # d_w depends on the row
blk[:, :, 0:4] = dw.view(np.uint8).reshape(M, KB, 4)
if qt == 3:
    blk[:, :, 4:8] = np.zeros((M, KB), dtype=np.float32).view(np.uint8).reshape(M, KB, 4)       # m = 0
    blk[:, :, 8:] = 0x11                                                                             # nibble 1
else:
    blk[:, :, 4:] = 0x99                                                                             # nibble 9 -> w = +1
W = ops.QTensor(qt, blk.reshape(M, -1), M, K)
x = np.zeros((N, K), dtype=np.float32)
for b in range(KB):
    x[:, b * 32:(b + 1) * 32] = (b % 5 + 1) * (1 + (np.arange(N) % 3))[:, None]                     # d_x depends on block and column
xt = torch.from_numpy(x).cuda()
a = ops.QAct(N, K).quantize(xt)
ldy = (M + 3) // 4 * 4
outs = {}
for c in (12, cfg):
    L.fl_debug_set(0, c)
    y = torch.full((N, ldy), -1.0, device="cuda")[:, :M]
    ops.mul_mat_q(W, a, out=y)
    torch.cuda.synchronize()
    outs[c] = y.cpu().numpy()
ref, got = outs[12], outs[cfg]
want = (x.reshape(N, KB, 32).sum(-1)[:, None, :] * dw[None, :, :]).sum(-1)
print("old kernel vs closed form:", np.abs(ref - want).max())
bad = got != ref
print("mismatch count", bad.sum(), "of", bad.size)
if bad.any():
    rows = np.nonzero(bad.any(axis=0))[0]
    cols = np.nonzero(bad.any(axis=1))[0]
    print("bad rows (m):", rows[:40], "... total", len(rows))
    print("bad cols (n):", cols[:40], "... total", len(cols))
    for n in cols[:3]:
        for m in rows[:6]:
            print(f"  n={n} m={m}: got {got[n, m]:.4f} want {ref[n, m]:.4f} ratio {got[n, m] / ref[n, m]:.4f}")
