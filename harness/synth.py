"""Synthetic LLaMA-shaped Q4_0 / Q4_1 weights, generated ON THE GPU with torch (harness only).

Follows the weight recipe of SURVEY.md section 8(d): f32 ~ N(0, 0.02^2) per tensor, quantized per
32-element block with the rule of the reference's quantize_row_q4_{0,1}_reference
(/root/reference/lib/ggml.c:630-664, :917-956: d = amax/7, q = roundf(x/d)+8  |  d = (max-min)/15,
q = roundf((x-min)/d)), packed into the reference's AoS block bytes (element 2j -> low nibble).
The CPU oracle's quantizer (oracle.Port.quantize_q4) is the bit-exact one used by the parity tests;
this torch version exists so that 4-40 GB of bench weights take seconds instead of minutes.
"""
from __future__ import annotations

import torch

Q4_0, Q4_1 = 2, 3
BLOCK_BYTES = {Q4_0: 20, Q4_1: 24}

# n_embd, n_head, n_layer, n_ff (n_ff from lib/llama.cpp:129 with n_mult = 256), n_vocab
MODELS = {
    "7B": dict(n_embd=4096, n_head=32, n_layer=32, n_ff=11008, n_vocab=32000),
    "13B": dict(n_embd=5120, n_head=40, n_layer=40, n_ff=13824, n_vocab=32000),
    "30B": dict(n_embd=6656, n_head=52, n_layer=60, n_ff=17920, n_vocab=32000),
    "65B": dict(n_embd=8192, n_head=64, n_layer=80, n_ff=22016, n_vocab=32000),
    # tiny shape for smoke tests (same structure, every K a multiple of 64)
    "tiny": dict(n_embd=256, n_head=4, n_layer=2, n_ff=704, n_vocab=512),
    # ... and one that splits eight ways by rows (8 heads, n_ff / 8 a multiple of 32): the 8-rank rehearsal of bench.py's tensor-parallel leg
    "tiny8": dict(n_embd=512, n_head=8, n_layer=2, n_ff=1024, n_vocab=512),
}


def layer_matmuls(cfg):
    """(name, M, K) of the 7 quantized matmuls of one layer, in eval order (lib/llama.cpp:328-435)."""
    E, F = cfg["n_embd"], cfg["n_ff"]
    return [("wq", E, E), ("wk", E, E), ("wv", E, E), ("wo", E, E), ("w3", F, E), ("w1", F, E), ("w2", E, F)]


def algorithmic_work(cfg, N, qtype=Q4_0):
    """bytes = M*K/32*B_blk + 4*N*K + 4*N*M, flops = 2*M*K*N, summed over the
    7*n_layer + 1 quantized matmuls of one eval."""
    bb = BLOCK_BYTES[qtype]
    mats = [(M, K) for _, M, K in layer_matmuls(cfg)] * cfg["n_layer"] + [(cfg["n_vocab"], cfg["n_embd"])]
    wbytes = sum(M * K // 32 * bb for M, K in mats)
    abytes = sum(4 * N * K + 4 * N * M for M, K in mats)
    flops = sum(2 * M * K * N for M, K in mats)
    return dict(weight_bytes=wbytes, act_bytes=abytes, bytes=wbytes + abytes, flops=flops, n_matmuls=len(mats))


@torch.no_grad()
def quantize_q4_torch(w: torch.Tensor, qtype: int) -> torch.Tensor:
    """f32 [M, K] (device) -> uint8 [M, K/32*block_bytes] AoS blocks (device), through the library's
    quantize_row_q_reference -- the arithmetic of the reference's ggml_quantize_q4_0/1 (lib/ggml.c:12122-12166),
    bit for bit (tests/test_kernels_gpu.py::test_quantize_row_q4_slots_match_reference): a synthetic model written
    from these blocks is the file the CPU reference would have quantized itself."""
    from fastllama_amd import ops
    M, K = w.shape
    return ops.quantize_row_q(qtype, w.contiguous().view(-1), reference=True).view(M, -1)


@torch.no_grad()
def synth_q4(M: int, K: int, qtype: int, seed: int, device="cuda", scale: float = 0.02) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((M, K // 32 * BLOCK_BYTES[qtype]), dtype=torch.uint8, device=device)
    rows = max(16, min(M, (1 << 26) // K))          # generate in <=256 MiB f32 slabs
    for r0 in range(0, M, rows):
        r1 = min(M, r0 + rows)
        w = torch.randn((r1 - r0, K), generator=g, device=device, dtype=torch.float32) * scale
        out[r0:r1] = quantize_q4_torch(w, qtype)
    return out


def synth_model_tensors(cfg, qtype, seed=1234, device="cuda", scale=0.02):
    """Generator of (name, (gtype, shape, device tensor)) for a whole LLaMA-shaped model, in GGJT tensor order.
    2-D weights ~ N(0, 0.02^2) quantized on the GPU (SURVEY.md 8d recipe); norm vectors 1 + 0.1 N(0,1).
    scale: a float, or a callable K -> sigma (e.g. lambda K: 0.5 / K ** 0.5 for a non-expansive network)."""
    from . import ggjt
    E, L, V = cfg["n_embd"], cfg["n_layer"], cfg["n_vocab"]
    F = cfg["n_ff"]
    g = torch.Generator(device=device)
    for t, (name, shape) in enumerate(ggjt.tensor_specs(V, E, L, F)):
        if len(shape) == 1:
            g.manual_seed(seed + t)
            v = 1.0 + 0.1 * torch.randn(shape[0], generator=g, device=device, dtype=torch.float32)
            yield name, (0, shape, v)
        else:
            K, M = shape
            yield name, (qtype, shape, synth_q4(M, K, qtype, seed + t, device=device, scale=scale(K) if callable(scale) else scale))
