"""Per-workgroup clocks of the reference-order V.P launch at LLaMA-7B, N = 512 (needs a -DXA_TIMING build of exact_kernels.hip, e.g.
X_SRC=exact_kernels.hip X_FLAGS=-DXA_TIMING TAG=xa OUT=gpurun_variants/libxa.so bash scripts/dev/fastbuild.sh and FASTLLAMA_HIP_LIB=...)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
L = hip.load(); hip.require_device(0)
D, H, N, n_past, n_ctx = 128, 32, 512, 0, 512
E = D * H
rng = np.random.default_rng(1)
qkv = torch.from_numpy(rng.standard_normal((N, 3 * E)).astype(np.float32)).cuda()
kc = torch.from_numpy(rng.standard_normal((n_ctx, E)).astype(np.float32)).cuda()
vc = torch.from_numpy(rng.standard_normal((E, n_ctx)).astype(np.float32)).cuda()
e = np.empty(1 << 16, np.uint16); L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
ed = torch.from_numpy(e.view(np.int16)).cuda()
att = torch.zeros((H, N, n_ctx), device="cuda"); ao = torch.zeros((N, E), device="cuda")
for _ in range(3):
    hip.check(L.fl_debug_attn_exact(qkv.data_ptr(), 3 * E, D, H, N, n_past, n_ctx, E, kc.data_ptr(), vc.data_ptr(), ed.data_ptr(), 0.0884, att.data_ptr(), ao.data_ptr(), 1, None))
torch.cuda.synchronize()
lib = C.CDLL(hip.LIB_PATH)
buf = (C.c_longlong * (1024 * 16))()
lib.fl_debug_xa_timing.argtypes = [C.c_void_p]
assert lib.fl_debug_xa_timing(buf) == 0
raw = np.array(buf[:]).reshape(1024, 16)[:512, :8].astype(np.float64) * 10e-3      # us; workgroup id = blockIdx.y * 32 + head
t0 = raw[:, 0].min()
names = ["entry -> P and V(0) in LDS", "MFMAs of feature block 0", "t exchange + final sums (wave 0)", "Q8_0 of the tile", "next V block to LDS", "MFMAs of feature block 1"]
for qb_y in (0, 1, 7, 15):          # blockIdx.y: 0 = heaviest query block (qb 15)
    r = raw[qb_y * 32:(qb_y + 1) * 32]
    d = [np.median(r[:, k + 1] - r[:, k]) for k in range(6)]
    print(f"query block {15 - qb_y:2d}: start {np.median(r[:, 0]) - t0:6.2f} end {np.median(r[:, 7]) - t0:6.2f} us | " + " | ".join(f"{n} {v:.2f}" for n, v in zip(names, d)))
print("launch: last end", raw[:, 7].max() - t0)
