"""The LLaMA hot path as a flat list of quantized matmuls (harness; drives the C-ABI directly).

One ``step(N)`` = one pass over the 7*n_layer + 1 ``mul_mat_q_f32`` calls of one ``Model::eval``
(/root/reference/lib/llama.cpp:328,329,334,401,422,426,435,465) with N activation columns:
INIT (quantize_row_q8_0 of the input rows, shared by wq/wk/wv and by w3/w1 -- the reference quantizes
per matmul, lib/ggml.c:8105-8119, the values are identical) + COMPUTE for every weight.
Inputs are fixed synthetic activations resident in HBM; every layer owns its own weights so a pass
streams the whole model exactly once (no cross-step cache reuse beyond what a real eval would see).
"""
from __future__ import annotations

import ctypes as C

import torch

from fastllama_amd import hip, ops
from . import synth


class HotPath:
    def __init__(self, model: str = "7B", qtype: int = synth.Q4_0, max_N: int = 512, seed: int = 1234,
                 device: int = 0, n_layer: int | None = None):
        self.cfg = dict(synth.MODELS[model])
        if n_layer is not None:
            self.cfg["n_layer"] = n_layer
        self.model, self.qtype, self.max_N = model, qtype, max_N
        self.lib = hip.load()
        hip.require_device(device)
        E, F, V = self.cfg["n_embd"], self.cfg["n_ff"], self.cfg["n_vocab"]
        self.layers = []
        t = 0
        for il in range(self.cfg["n_layer"]):
            ws = {}
            for name, M, K in synth.layer_matmuls(self.cfg):
                blocks = synth.synth_q4(M, K, qtype, seed + t)
                ws[name] = ops.QTensor(qtype, blocks, M, K)
                t += 1
                del blocks
            self.layers.append(ws)
        blocks = synth.synth_q4(V, E, qtype, seed + t)
        self.head = ops.QTensor(qtype, blocks, V, E)
        del blocks
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        self.xE = torch.randn((max_N, E), generator=g, device="cuda")
        self.xE2 = torch.randn((max_N, E), generator=g, device="cuda")
        self.xF = torch.randn((max_N, F), generator=g, device="cuda")
        self.qE = ops.QAct(max_N, E)
        self.qF = ops.QAct(max_N, F)
        ld = lambda m: (m + 3) // 4 * 4
        self.yE = [torch.empty((max_N, ld(E)), device="cuda") for _ in range(3)]
        self.yF = [torch.empty((max_N, ld(F)), device="cuda") for _ in range(2)]
        self.yV = torch.empty((max_N, ld(V)), device="cuda")
        torch.cuda.synchronize()

    # -- raw C-ABI calls (minimal Python overhead) --
    def _quant(self, qa, x, N):
        hip.check(self.lib.fl_quantize_q8(qa.handle, C.c_void_p(x.data_ptr()), x.stride(0), N, x.shape[1], None))
        qa.N = N

    def _mm(self, W, qa, y):
        hip.check(self.lib.fl_mul_mat_q(W.handle, qa.handle, C.c_void_p(y.data_ptr()), y.stride(0), None))

    def step(self, N: int, quantize: bool = True, matmul: bool = True):
        """One pass of the hot path on the null stream (torch's default stream)."""
        q, mm = self._quant, self._mm
        for ws in self.layers:
            if quantize:
                q(self.qE, self.xE, N)
            if matmul:
                mm(ws["wq"], self.qE, self.yE[0]); mm(ws["wk"], self.qE, self.yE[1]); mm(ws["wv"], self.qE, self.yE[2])
            if quantize:
                q(self.qE, self.xE2, N)
            if matmul:
                mm(ws["wo"], self.qE, self.yE[0])
            if quantize:
                q(self.qE, self.xE, N)
            if matmul:
                mm(ws["w3"], self.qE, self.yF[0]); mm(ws["w1"], self.qE, self.yF[1])
            if quantize:
                q(self.qF, self.xF, N)
            if matmul:
                mm(ws["w2"], self.qF, self.yE[1])
        if quantize:
            q(self.qE, self.xE2, N)
        if matmul:
            mm(self.head, self.qE, self.yV)

    def prepare(self, N: int):
        """Make sure both activation workspaces hold N quantized columns (for matmul-only timing)."""
        self._quant(self.qE, self.xE, N)
        self._quant(self.qF, self.xF, N)

    def work(self, N: int):
        return synth.algorithmic_work(self.cfg, N, self.qtype)

    def weight_bytes_device(self) -> int:
        return sum(w.device_bytes for ws in self.layers for w in ws.values()) + self.head.device_bytes
