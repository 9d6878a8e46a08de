"""Round-4 exact GEMM (H16 form, which=5) against round 3's (which=6): bit identity over shapes / types / ragged N, then timings.
Development aid: python scripts/dev/xh_check.py [--perf-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fastllama_amd import hip, ops
from harness import synth

L = hip.load()
hip.require_device(0)
ok = True
if "--perf-only" not in sys.argv:
    for qt in (2, 3):
        for (M, K) in [(64, 64), (96, 128), (4096 + 16, 4096), (1000, 160)]:
            W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 3), M, K)
            for N in (9, 31, 33, 64, 100, 512):
                x = torch.randn(N, K, device="cuda") * 3
                a = ops.QAct(N, K).quantize(x, layout=16)
                y5 = torch.full((N, M + 3), -7.0, device="cuda")[:, :M]
                y6 = torch.full((N, M + 3), -7.0, device="cuda")[:, :M]
                y5 = torch.zeros(N, (M + 3) // 4 * 4, device="cuda")
                y6 = torch.zeros(N, (M + 3) // 4 * 4, device="cuda")
                ops.mul_mat_q(W, a, which=5, out=y5[:, :M])
                ops.mul_mat_q(W, a, which=6, out=y6[:, :M])
                torch.cuda.synchronize()
                same = torch.equal(y5.view(torch.int32), y6.view(torch.int32))
                if not same:
                    ok = False
                    d = (y5 != y6).nonzero()
                    print(f"DIFF qt={qt} M={M} K={K} N={N}: {d.shape[0]} of {N*M} differ; first {d[:4].tolist()}  y5 {y5[tuple(d[0])].item()} y6 {y6[tuple(d[0])].item()}", flush=True)
                a.free()
            W.free()
    print("bit identity:", "OK" if ok else "FAILED", flush=True)
quick = bool(os.environ.get("XH_QUICK"))
for qt in ((2,) if quick else (2, 3)):
    for (M, K) in ([(12288, 4096), (4096, 11008)] if quick else [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32000, 4096)]):
        W = ops.QTensor(qt, synth.synth_q4(M, K, qt, 3), M, K)
        for N in (512,):
            x = torch.randn(N, K, device="cuda")
            a = ops.QAct(N, K).quantize(x, layout=16)
            y = torch.empty(N, M, device="cuda")
            res = []
            for which in ((5, 5, 5) if quick else (1, 6, 5)):
                ops.mul_mat_q(W, a, which=which, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    ops.mul_mat_q(W, a, which=which, out=y)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / reps * 1e3)
            tb = M * K // 32 * N / 1024
            print(f"qt={qt} M={M:6d} K={K:6d} N={N:4d}  fast {res[0]:8.1f} us  exact-r3 {res[1]:8.1f} us  exact-h16(+conv) {res[2]:8.1f} us  "
                  f"r3/h16 {res[1]/res[2]:5.2f}  h16: {2.0*M*K*N/res[2]/1e6:8.1f} TOP/s  {res[2]*1e-6*2.4e9*1024/tb:6.0f} cyc/tile-block", flush=True)
            a.free()
        W.free()
sys.exit(0 if ok else 1)
