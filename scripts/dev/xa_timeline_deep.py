"""Per-workgroup clocks of the reference-order V.P launch behind a deep context (LLaMA-7B heads, N = 512 at n_past 1536; needs a -DXA_TIMING build:
X_SRC=exact_kernels.hip X_FLAGS=-DXA_TIMING TAG=xa OUT=gpurun_variants/libxa.so bash scripts/dev/fastbuild.sh and FASTLLAMA_HIP_LIB=...)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
L = hip.load(); hip.require_device(0)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D, H, N, n_past, n_ctx = 128, 32, 512, 1536, 2048
E = D * H
rng = np.random.default_rng(1)
qkv = torch.from_numpy(rng.standard_normal((N, 3 * E)).astype(np.float32)).cuda()
kc = torch.from_numpy(rng.standard_normal((n_ctx, E)).astype(np.float32)).cuda()
vc = torch.from_numpy(rng.standard_normal((E, n_ctx)).astype(np.float32)).cuda()
e = np.empty(1 << 16, np.uint16); L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
ed = torch.from_numpy(e.view(np.int16)).cuda()
att = torch.zeros((H, N, n_ctx), device="cuda"); ao = torch.zeros((N, E), device="cuda")
for _ in range(3):
    hip.check(L.fl_debug_attn_exact(qkv.data_ptr(), 3 * E, D, H, N, n_past, n_ctx, E, kc.data_ptr(), vc.data_ptr(), ed.data_ptr(), 0.0884, att.data_ptr(), ao.data_ptr(), which, None))
torch.cuda.synchronize()
lib = C.CDLL(hip.LIB_PATH)
buf = (C.c_longlong * (1024 * 16))()
lib.fl_debug_xa_timing.argtypes = [C.c_void_p]
assert lib.fl_debug_xa_timing(buf) == 0
raw = np.array(buf[:]).reshape(1024, 16)[:512].astype(np.float64) * 10e-3      # us; workgroup id = blockIdx.y * 32 + head
t0 = raw[:, 0].min()
for qb_y in (0, 1, 7, 8, 15):          # blockIdx.y: 0 = heaviest query block (qb 15)
    r = raw[qb_y * 32:(qb_y + 1) * 32]
    med = lambda a, b: np.median(r[:, a] - r[:, b])
    pieces = " ".join(f"[wait {med(8 + 2 * c, 9 + 2 * c - 2 if c else 1):.2f} store {med(9 + 2 * c, 8 + 2 * c):.2f}]" for c in range(4))
    print(f"query block {15 - qb_y:2d}: start {np.median(r[:, 0]) - t0:7.2f} end {np.median(r[:, 7]) - t0:7.2f} us | entry->loop {med(1, 0):.2f} | fb0 chains done {med(2, 1):.2f} | "
          f"exchange+sums {med(3, 2):.2f} | q8 {med(4, 3):.2f} | fb1 chains {med(6, 5):.2f} | pieces of fb0 (barrier-to-barrier; 'wait' includes the previous piece's chains): {pieces}")
print("launch: last end", raw[:, 7].max() - t0)
if hasattr(lib, "fl_debug_xa_cycles"):
    lib.fl_debug_xa_cycles.argtypes = [C.c_void_p]
    assert lib.fl_debug_xa_cycles(buf) == 0
    cyc = np.array(buf[:]).reshape(1024, 16)[:512].astype(np.float64)
    mhz = (cyc[:, 7] - cyc[:, 0]) / (raw[:, 7] - raw[:, 0])
    print(f"shader clock over the workgroups' lifetimes: median {np.median(mhz):.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
