"""gemm_q4_mfma32 (cfg ids 100..105) against gemm_q4_mfma (cfg 12): bit equality of plain / residual outputs on ragged and
LLaMA shapes, then timing.   python scripts/dev/g32_check.py [check|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
CFGS = [100, 101, 102, 103, 104, 105, 106, 108]
bad = 0
if mode in ("check", "all"):
    shapes = [(32, 128, 32), (48, 192, 17), (200, 1408, 9), (130, 256, 70), (4096, 4096, 512), (1000, 4096, 100), (4096, 11008, 64),
              (264, 320, 33), (256, 352, 33), (40, 96, 20), (33, 32, 16), (4096, 5504, 48)]
    for qt in (2, 3):
        for (M, K, N) in shapes:
            W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
            x = torch.randn(N, K, device="cuda") * 3
            a = ops.QAct(N, K).quantize(x)
            ldy = (M + 3) // 4 * 4
            L.fl_debug_set(0, 12)
            ref = torch.zeros(N, ldy, device="cuda")[:, :M]
            ops.mul_mat_q(W, a, out=ref)
            for cfg in CFGS:
                L.fl_debug_set(0, cfg)
                y = torch.full((N, ldy), 7.0, device="cuda")[:, :M]
                ops.mul_mat_q(W, a, out=y)
                torch.cuda.synchronize()
                ok = torch.equal(y, ref)
                if not ok:
                    bad += 1
                    d = (y - ref).abs()
                    nbad = int((y != ref).sum())
                    idx = torch.nonzero(y != ref)[:4].tolist()
                    print(f"MISMATCH q{qt} M={M} K={K} N={N} cfg={cfg}: {nbad} elements differ, max |d| {float(d.max()):.3g} of {float(ref.abs().max()):.3g}; first {idx}")
            W.free()
    print("bit check:", "all equal" if bad == 0 else f"{bad} mismatching (shape, cfg) pairs")
if mode in ("time", "all"):
    N = 512
    for qt in (2, 3):
        for (M, K) in [(4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008), (32000, 4096)]:
            W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 1), M, K)
            x = torch.randn(N, K, device="cuda")
            a = ops.QAct(N, K).quantize(x)
            y = torch.empty(N, M, device="cuda")
            res = []
            for cfg in [-1] + CFGS:
                L.fl_debug_set(0, cfg)
                for _ in range(3):
                    ops.mul_mat_q(W, a, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.mul_mat_q(W, a, out=y)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                res.append((cfg, ms * 1e3, 2.0 * M * K * N / ms / 1e9))
            print(f"q4_{qt-2} M={M:6d} K={K:6d} N={N}: " + "  ".join(f"c{c}:{us:7.1f}us/{t:5.0f}T" for c, us, t in res), flush=True)
            W.free()
sys.exit(1 if bad else 0)
