"""Build an fl_model (include/fastllama_hip.h, "the model") straight from in-memory tensors (harness)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from fastllama_amd import hip
from . import ggjt


class FlModel:
    def __init__(self, cfg: dict, qtype: int, tensors: dict, n_ctx: int, max_batch: int, tp_rank=0, tp_size=1,
                 device: int = 0):
        L = self.L = hip.load()
        hip.require_device(device)
        self.cfg, self.qtype = cfg, qtype
        E = cfg["n_embd"]
        self.V = cfg["n_vocab"]
        self.E = E
        p = hip.ModelParams(cfg["n_vocab"], E, cfg["n_head"], cfg["n_layer"],
                            cfg["n_ff"] if "n_ff" in cfg else ggjt.n_ff_of(E, cfg["n_mult"]), n_ctx, qtype,
                            max_batch, tp_rank, tp_size)
        h = L.fl_model_create(C.byref(p))
        if not h:
            raise hip.FastLlamaHipError("fl_model_create: " + L.fl_last_error().decode())
        self.h = C.c_void_p(h)
        for name, (gtype, shape, data) in (tensors.items() if hasattr(tensors, "items") else tensors):
            if hasattr(data, "data_ptr"):                       # torch tensor (host or device memory)
                ptr = C.c_void_p(data.data_ptr())
            else:
                data = np.ascontiguousarray(data)
                ptr = data.ctypes.data_as(C.c_void_p)
            hip.check(L.fl_model_set_tensor(self.h, name.encode(), gtype, ptr, shape[0],
                                            shape[1] if len(shape) > 1 else 1), "fl_model_set_tensor " + name)
            del data
        hip.check(L.fl_model_finalize(self.h), "fl_model_finalize")

    def set_exact(self, on: bool):
        """reference-order matmuls (logits bit-identical to the reference) / the fast kernels"""
        hip.check(self.L.fl_model_set_exact(self.h, 1 if on else 0), "fl_model_set_exact")

    def prepare(self, flags: int = 3) -> int:
        """build the derived weight copies of the reference-order kernels now (bit 0: WH16, prefill; bit 1: QWD, decode) instead of
        inside the first eval that needs them; returns which are resident (fl_model_prepared)"""
        hip.check(self.L.fl_model_prepare(self.h, flags), "fl_model_prepare")
        return self.L.fl_model_prepared(self.h)

    def eval_last_logits(self, toks_i32: np.ndarray, n_past: int, out: np.ndarray):
        """eval that hands the LAST token's logits back to the host (what llama_eval() callers get): the bench's timed call."""
        hip.check(self.L.fl_model_eval(self.h, toks_i32.ctypes.data_as(C.c_void_p), toks_i32.size, n_past,
                                       out.ctypes.data_as(C.c_void_p), 0, None), "fl_model_eval")

    def set_comm(self, comm):
        hip.check(self.L.fl_model_set_comm(self.h, comm), "fl_model_set_comm")

    def eval_nocopy(self, toks_i32: np.ndarray, n_past: int):
        """eval without copying logits back (they stay in HBM: fl_model_logits_dev) -- the bench's timed call."""
        hip.check(self.L.fl_model_eval(self.h, toks_i32.ctypes.data_as(C.c_void_p), toks_i32.size, n_past, None, 0, None),
                  "fl_model_eval")

    def profile(self, enable: int):
        ms, n = C.c_double(), C.c_long()
        hip.check(self.L.fl_model_profile(self.h, enable, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def eval(self, tokens, n_past=0, all_logits=False, embeddings=False):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        N = toks.size
        lg = np.empty((N if all_logits else 1, self.V), dtype=np.float32)
        emb = np.empty(self.E, dtype=np.float32) if embeddings else None
        hip.check(self.L.fl_model_eval(self.h, toks.ctypes.data_as(C.c_void_p), N, n_past, lg.ctypes.data_as(C.c_void_p),
                                       1 if all_logits else 0, emb.ctypes.data_as(C.c_void_p) if embeddings else None),
                  "fl_model_eval")
        return (lg, emb) if embeddings else lg

    def ingest(self, tokens, chunk, n_past=0, want_logits=True):
        """The consecutive evals of a long prompt, `chunk` tokens at a time, pipelined on two streams (fl_model_ingest).
        Returns the last token's logits."""
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        lens = np.array([min(chunk, toks.size - i) for i in range(0, toks.size, chunk)], dtype=np.int32)
        lg = np.empty((1, self.V), dtype=np.float32) if want_logits else None
        hip.check(self.L.fl_model_ingest(self.h, toks.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), lens.size, n_past,
                                         lg.ctypes.data_as(C.c_void_p) if want_logits else None), "fl_model_ingest")
        return lg

    def free(self):
        if getattr(self, "h", None):
            self.L.fl_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
