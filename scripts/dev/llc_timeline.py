"""Per-workgroup clocks of the reference-order decode GEMV launches at LLaMA-7B shapes (needs a -DLLC_TIMING build of gemv1_q4_exact_llc.hip:
X_SRC=gemv1_q4_exact_llc.hip X_FLAGS=-DLLC_TIMING TAG=llct OUT=gpurun_variants/libllct.so bash scripts/dev/fastbuild.sh; FASTLLAMA_HIP_LIB=...)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
L.fl_debug_set(2, 1)
lib = C.CDLL(hip.LIB_PATH); lib.fl_debug_llc_timing.argtypes = [C.c_void_p]
s = np.empty(1 << 16, np.uint16); L.fl_debug_tables(None, s.ctypes.data_as(C.c_void_p)); sd = torch.from_numpy(s.view(np.int16)).cuda()
names = ["entry -> weight loads + prologue loads issued", "prologue (Q8_0 of x in LDS, lane view)", "order-free part of wave 0 (lane sums, scales)", "chains: all slices, hand-offs", "reduce + store"]
def report(tag, n_wg):
    torch.cuda.synchronize()
    buf = (C.c_longlong * (2048 * 8))(); assert lib.fl_debug_llc_timing(buf) == 0
    n = min(n_wg, 2048)
    t = np.array(buf[:]).reshape(2048, 8)[:n, :6].astype(np.float64) * 10e-3
    t0 = t[:, 0].min()
    d = [np.median(t[:, k + 1] - t[:, k]) for k in range(5)]
    print(f"{tag}: {n_wg} workgroups; start median {np.median(t[:, 0]) - t0:.2f} max {t[:, 0].max() - t0:.2f}; end max {t[:, 5].max() - t0:.2f} us | " + " | ".join(f"{nm} {v:.2f}" for nm, v in zip(names, d)))
E, F = 4096, 11008
x = torch.randn(1, E, device="cuda"); nw = torch.ones(E, device="cuda")
# wq|wk|wv with the rms_norm prologue
W = ops.QTensor(2, synth.synth_q4(3 * E, E, 2, 1), 3 * E, E); y = torch.empty(3 * E, device="cuda"); yn = torch.empty(1, E, device="cuda")
for _ in range(3): hip.check(L.fl_debug_gemv_norm(W.handle, x.data_ptr(), nw.data_ptr(), yn.data_ptr(), y.data_ptr(), None))
report("wq|wk|wv 12288x4096 (norm prologue)", 3 * E // 16)
# wo, plain
Wo = ops.QTensor(2, synth.synth_q4(E, E, 2, 2), E, E); a = ops.QAct(1, E).quantize(x); yo = torch.empty(1, E, device="cuda")
for _ in range(3): ops.mul_mat_q(Wo, a, which=3, out=yo)
report("wo 4096x4096 (Q8_0 operand)", E // 16)
# w1|w3 woven, two workgroups per feature pair
W13 = ops.QTensor(2, synth.synth_q4(2 * F, E, 2, 3), 2 * F, E); act = torch.empty(F, device="cuda")
for _ in range(3): hip.check(L.fl_debug_gemv_norm_silu(W13.handle, x.data_ptr(), nw.data_ptr(), sd.data_ptr(), act.data_ptr(), None))
report("w1|w3 22016x4096 (norm prologue, automatic form)", 2 * F // 16)
for form, n in ((2, 2 * F // 16), (1, F // 16)):
    L.fl_debug_set(5, form)
    for _ in range(3): hip.check(L.fl_debug_gemv_norm_silu(W13.handle, x.data_ptr(), nw.data_ptr(), sd.data_ptr(), act.data_ptr(), None))
    report(f"w1|w3 22016x4096 (norm prologue, form {form})", n)
L.fl_debug_set(5, 0)
# w2 with the Q8_0 prologue
W2 = ops.QTensor(2, synth.synth_q4(E, F, 2, 4), E, F); y2 = torch.empty(E, device="cuda"); xf = torch.randn(F, device="cuda")
for _ in range(3): hip.check(L.fl_debug_gemv_quant(W2.handle, xf.data_ptr(), y2.data_ptr(), None, None))
report("w2 4096x11008 (Q8_0 prologue)", E // 16)
L.fl_debug_set(2, 0)
