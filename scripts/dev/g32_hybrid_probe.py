"""Would finishing a launch's last partial round with half-size (128x32) tiles pay?  22016 x 4096 x 512 as ONE launch of 128x64
tiles (1376 workgroups = 5.4 per CU) against 20480 rows of 128x64 tiles + 1536 rows of 128x32 tiles on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastllama_amd import hip, ops
from harness import synth
L = hip.load(); hip.require_device(0)
K, N = 4096, 512
def mk(M):
    return ops.QTensor(2, synth.synth_q4(M, K, 2, 1), M, K)
W, Wa, Wb = mk(22016), mk(20480), mk(1536)
a = ops.QAct(N, K).quantize(torch.randn(N, K, device="cuda"))
y, ya, yb = (torch.empty(N, m, device="cuda") for m in (22016, 20480, 1536))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def one():
    L.fl_debug_set(0, 106); hip.check(L.fl_mul_mat_q(W.handle, a.handle, y.data_ptr(), 22016, None))
def part_a():
    L.fl_debug_set(0, 106); hip.check(L.fl_mul_mat_q(Wa.handle, a.handle, ya.data_ptr(), 20480, None))
def part_b():
    L.fl_debug_set(0, 101); hip.check(L.fl_mul_mat_q(Wb.handle, a.handle, yb.data_ptr(), 1536, None))
def both():
    ev = torch.cuda.Event(); ev.record()
    s1.wait_event(ev); s2.wait_event(ev)
    L.fl_debug_set(0, 106); hip.check(L.fl_mul_mat_q(Wa.handle, a.handle, ya.data_ptr(), 20480, C.c_void_p(s1.cuda_stream)))
    L.fl_debug_set(0, 101); hip.check(L.fl_mul_mat_q(Wb.handle, a.handle, yb.data_ptr(), 1536, C.c_void_p(s2.cuda_stream)))
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
import ctypes as C
print(f"one launch 22016 rows, 128x64 tiles : {t(one):7.1f} us")
print(f"20480 rows, 128x64 tiles            : {t(part_a):7.1f} us")
print(f"1536 rows, 128x32 tiles             : {t(part_b):7.1f} us")
print(f"both, two streams                   : {t(both):7.1f} us")
L.fl_debug_set(0, 0)
