"""Time of the reference-order deep-context attention (K.Q, soft_max, V.P through the test hook; LLaMA-7B heads, N = 512 at n_past 1536) for one library:
FASTLLAMA_HIP_LIB=... python scripts/dev/attn_deep_ab.py [which] -- run alternately for the libraries to compare, in ONE gpurun call."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip
L = hip.load(); hip.require_device(0)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2: L.fl_debug_set(8, int(sys.argv[2]))          # waves per workgroup of the deep V.P kernel (8 / 4)
D, H, N, n_past, n_ctx = 128, 32, 512, 1536, 2048
E = D * H
rng = np.random.default_rng(1)
qkv = torch.from_numpy(rng.standard_normal((N, 3 * E)).astype(np.float32)).cuda()
kc = torch.from_numpy(rng.standard_normal((n_ctx, E)).astype(np.float32)).cuda()
vc = torch.from_numpy(rng.standard_normal((E, n_ctx)).astype(np.float32)).cuda()
e = np.empty(1 << 16, np.uint16); L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
ed = torch.from_numpy(e.view(np.int16)).cuda()
att = torch.zeros((H, N, n_ctx), device="cuda"); ao = torch.zeros((N, E), device="cuda")
def run():
    hip.check(L.fl_debug_attn_exact(qkv.data_ptr(), 3 * E, D, H, N, n_past, n_ctx, E, kc.data_ptr(), vc.data_ptr(), ed.data_ptr(), 0.0884, att.data_ptr(), ao.data_ptr(), which, None))
for _ in range(20): run()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(40): run()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 40 * 1e3)
print(f"{os.environ.get('FASTLLAMA_HIP_LIB', 'default')} {' '.join(sys.argv[1:])}: K.Q + soft_max + V.P {np.median(ts):.1f} us (min {min(ts):.1f}, max {max(ts):.1f})", flush=True)
