import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from oracle import llama_eval as le
from fastllama_amd import hip, ops
L = hip.load(); hip.require_device(0)
port = oracle.Port()
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (D, H, n_past) in [(32, 4, 9), (96, 2, 515), (32, 4, 40), (32,4,8), (32,4,7),(32,4,3)]:
    n_ctx, E, P = 1024, H * D, n_past + 1
    rng = np.random.default_rng(D + H + n_past)
    qkv = rng.standard_normal((1, 3 * E)).astype(np.float32)
    kc = np.zeros((n_ctx, E), np.float32); vc = np.zeros((E, n_ctx), np.float32)
    kc[:n_past] = rng.standard_normal((n_past, E)); vc[:, :n_past] = rng.standard_normal((E, n_past)); vc[:, n_past:] = 7.0
    e = np.empty(1 << 16, np.uint16); L.fl_debug_tables(e.ctypes.data_as(C.c_void_p), None)
    rt = np.empty((n_ctx, D // 2, 2), np.float32); L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), n_ctx, D)
    ed, rd, qd = dev(e.view(np.int16)), dev(rt), dev(qkv)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    outs = {}
    L.fl_debug_set(2, 1)
    for split in (0, 1):
        kd, vd = dev(kc), dev(vc)
        a = ops.QAct(1, E)
        hip.check(L.fl_quantize_q8_layout(a.handle, qd.data_ptr(), 3 * E, 1, E, 1, None)); a.N, a.K = 1, E
        if split:
            sc = torch.full((H, n_ctx), float("nan"), device="cuda")
            hip.check(L.fl_debug_decode_attention_split(qd.data_ptr(), E, D, H, n_past, n_ctx, rd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ed.data_ptr(), scale, sc.data_ptr(), a.handle, None, None))
        else:
            hip.check(L.fl_debug_decode_attention(qd.data_ptr(), E, D, H, n_past, n_ctx, rd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ed.data_ptr(), scale, a.handle, None))
        torch.cuda.synchronize()
        outs[split] = a.export().cpu().numpy()[0].reshape(-1, 40).copy()
    L.fl_debug_set(2, 0)
    diff = np.where((outs[0] != outs[1]).any(axis=1))[0]
    print(D, H, n_past, "blocks differing fused vs split:", diff.tolist(), "of", outs[0].shape[0])
    for b in diff[:3]:
        d0 = outs[0][b, :4].copy().view(np.float32)[0]; d1 = outs[1][b, :4].copy().view(np.float32)[0]
        q0 = outs[0][b, 8:].view(np.int8); q1 = outs[1][b, 8:].view(np.int8)
        print("  block", b, "d", d0, d1, "quants differing at", np.where(q0 != q1)[0].tolist())
