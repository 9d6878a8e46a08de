"""Synthetic GGJT-v1 llama model files (harness).  File format as the reference reads it
(/root/reference/include/file_loader.hpp:94-250, writer side :301-373):

  u32 magic 'ggjt' 0x67676a74, u32 version 1
  7 x u32 hparams {n_vocab, n_embd, n_mult, n_head, n_layer, n_rot, ftype}   (ftype 2 = Q4_0, 3 = Q4_1)
  n_vocab x {u32 len, bytes, f32 score}
  tensors: {u32 n_dims, u32 name_len, u32 type, u32 ne[n_dims], name, pad to 32 B, data}

The vocabulary is byte level: id 0 <unk>, 1 <s>, 2 </s>, ids 3..258 the single bytes 0..255 (the
tokenizer's byte fallback is id = byte + 3, include/tokenizer.hpp:130-133), all scores 0, so a text of
c ASCII characters tokenizes to exactly [1] + [b + 3 for b in text].
"""
from __future__ import annotations

import struct

import numpy as np

Q4_0, Q4_1 = 2, 3
BLOCK_BYTES = {Q4_0: 20, Q4_1: 24}


def n_ff_of(n_embd: int, n_mult: int) -> int:
    return ((2 * (4 * n_embd) // 3 + n_mult - 1) // n_mult) * n_mult       # lib/llama.cpp:129


def tensor_specs(n_vocab, n_embd, n_layer, n_ff):
    """(name, [ne0 (row length K), ne1 (rows M)] or [ne0]) in file order; 2-D ones are quantized."""
    E, F, V = n_embd, n_ff, n_vocab
    out = [("tok_embeddings.weight", [E, V]), ("norm.weight", [E]), ("output.weight", [E, V])]
    for i in range(n_layer):
        p = f"layers.{i}."
        out += [(p + "attention_norm.weight", [E]),
                (p + "attention.wq.weight", [E, E]), (p + "attention.wk.weight", [E, E]),
                (p + "attention.wv.weight", [E, E]), (p + "attention.wo.weight", [E, E]),
                (p + "ffn_norm.weight", [E]),
                (p + "feed_forward.w1.weight", [E, F]), (p + "feed_forward.w2.weight", [F, E]),
                (p + "feed_forward.w3.weight", [E, F])]
    return out


def vocab_tokens(n_vocab: int):
    toks = []
    for i in range(n_vocab):
        if i == 0:
            t = b"<unk>"
        elif i == 1:
            t = b"<s>"
        elif i == 2:
            t = b"</s>"
        elif i < 259:
            t = bytes([i - 3])
        else:
            t = b"<%d>" % i
        toks.append(t)
    return toks


def synth_tensors(cfg: dict, qtype: int, quantize, seed: int = 1234, scale: float | None = None):
    """Random tensors for cfg = {n_vocab, n_embd, n_mult, n_head, n_layer}.  `quantize(qtype, f32[M,K]) -> uint8[M, bytes]`
    is the oracle's (bit-exact reference rule) weight quantizer.  2-D weights ~ N(0, scale^2) (default 1/sqrt(K), so
    that activations keep O(1) magnitude through the layers), norm vectors 1 + 0.1 N(0,1)."""
    E, L = cfg["n_embd"], cfg["n_layer"]
    F = n_ff_of(E, cfg["n_mult"])
    out = {}
    for t, (name, shape) in enumerate(tensor_specs(cfg["n_vocab"], E, L, F)):
        rng = np.random.default_rng(seed + t)
        if len(shape) == 1:
            out[name] = (0, shape, (1.0 + 0.1 * rng.standard_normal(shape[0])).astype(np.float32))
        else:
            K, M = shape
            sc = scale if scale is not None else 1.0 / np.sqrt(K)
            w = (rng.standard_normal((M, K)) * sc).astype(np.float32)
            out[name] = (qtype, shape, quantize(qtype, w))
    return out


def write_ggjt(path: str, cfg: dict, qtype: int, tensors: dict, container: str = "ggjt"):
    """container: "ggjt" (v1: scores in the vocabulary, tensor data aligned to 32 bytes), "ggmf" (v1: scores, no alignment) or
    "ggml" (the unversioned original: no version word, no scores, no alignment) -- the three layouts the reference's loader
    accepts (include/file_loader.hpp:94-250: read_magic_number :94-150, read_vocab :183-205, tensor alignment :236-240)."""
    E, H, L = cfg["n_embd"], cfg["n_head"], cfg["n_layer"]
    magic = {"ggjt": 0x67676A74, "ggmf": 0x67676D66, "ggml": 0x67676D6C}[container]
    with open(path, "wb") as f:
        f.write(struct.pack("<I", magic))
        if container != "ggml":
            f.write(struct.pack("<I", 1))
        f.write(struct.pack("<7I", cfg["n_vocab"], E, cfg["n_mult"], H, L, E // H, qtype))
        for t in vocab_tokens(cfg["n_vocab"]):
            f.write(struct.pack("<I", len(t)) + t + (struct.pack("<f", 0.0) if container != "ggml" else b""))
        for name, (gtype, shape, data) in tensors.items():
            nb = name.encode()
            f.write(struct.pack("<III", len(shape), len(nb), gtype))
            f.write(struct.pack("<%dI" % len(shape), *shape))
            f.write(nb)
            if container == "ggjt":
                f.write(b"\0" * ((-f.tell()) & 31))
            f.write(np.ascontiguousarray(data).tobytes())


def write_ggjt_stream(path: str, cfg: dict, qtype: int, tensor_iter):
    """write_ggjt for models that do not fit in host memory twice: `tensor_iter` yields (name, (gtype, shape, data)) one at a
    time (data: numpy array or torch tensor on any device), e.g. harness.synth.synth_model_tensors."""
    E, H, L = cfg["n_embd"], cfg["n_head"], cfg["n_layer"]
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0x67676A74, 1))
        f.write(struct.pack("<7I", cfg["n_vocab"], E, cfg["n_mult"], H, L, E // H, qtype))
        for t in vocab_tokens(cfg["n_vocab"]):
            f.write(struct.pack("<I", len(t)) + t + struct.pack("<f", 0.0))
        for name, (gtype, shape, data) in tensor_iter:
            nb = name.encode()
            f.write(struct.pack("<III", len(shape), len(nb), gtype))
            f.write(struct.pack("<%dI" % len(shape), *shape))
            f.write(nb)
            f.write(b"\0" * ((-f.tell()) & 31))
            arr = data.cpu().numpy() if hasattr(data, "cpu") else np.ascontiguousarray(data)
            f.write(memoryview(np.ascontiguousarray(arr)).cast("B"))


def write_ggjt_parts(path: str, cfg: dict, qtype: int, tensors: dict, n_parts: int):
    """The same model as `n_parts` files <path>, <path>.1, ... the way the original multi-part LLaMA checkpoints were
    converted: tok_embeddings / wo / w2 split along ne0 (columns), every other matrix along ne1 (rows), vectors whole in
    every part (the split table of the reference's loader, include/tensor/utils.hpp:93-112)."""
    for part in range(n_parts):
        sub = {}
        for name, (gtype, shape, data) in tensors.items():
            if len(shape) == 1:
                sub[name] = (gtype, shape, data)
                continue
            K, M = shape
            arr = np.ascontiguousarray(data).reshape(M, -1)
            if name.startswith("tok_embeddings.") or ".attention.wo.weight" in name or ".feed_forward.w2.weight" in name:
                w = arr.shape[1] // n_parts
                sub[name] = (gtype, (K // n_parts, M), arr[:, part * w:(part + 1) * w])
            else:
                rows = M // n_parts
                sub[name] = (gtype, (K, rows), arr[part * rows:(part + 1) * rows])
        write_ggjt(path if part == 0 else f"{path}.{part}", cfg, qtype, sub)


def write_lora(path: str, adapters: dict, r: int, alpha: int, cached: bool = False):
    """'ggla' v1 adapter file as scripts/convert-lora-to-ggml.py writes it (header :52-55, tensors :57-72).
    adapters: base tensor name -> (A [ne0(base), r] f32 ALREADY scaled by alpha/r, B [ne1(base), r] f32), or, with
    cached=True, base -> BA [ne1(base), ne0(base)] f32 (already scaled)."""
    def tensor(f, name, arr):
        nb = name.encode()
        f.write(struct.pack("<III", arr.ndim, len(nb), 0))
        f.write(struct.pack("<%dI" % arr.ndim, *arr.shape[::-1]))
        f.write(nb)
        f.write(b"\0" * ((-f.tell()) & 31))
        f.write(np.ascontiguousarray(arr, dtype=np.float32).tobytes())
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0x67676C61, 1))
        f.write(struct.pack("<?II", cached, r, alpha))
        for base, t in adapters.items():
            if cached:
                tensor(f, base + ".lora", t)
            else:
                tensor(f, base + ".loraA", t[0])
                tensor(f, base + ".loraB", t[1])


def text_tokens(text: str, bos: bool = True):
    return ([1] if bos else []) + [b + 3 for b in text.encode()]


TINY = dict(n_vocab=320, n_embd=128, n_mult=64, n_head=4, n_layer=2)       # n_ff = 384
SMALL = dict(n_vocab=512, n_embd=256, n_mult=64, n_head=8, n_layer=3)      # n_ff = 704
