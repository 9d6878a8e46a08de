"""ctypes driver for a shared library that exports the reference's `llama_*` C-ABI
(/root/reference/interfaces/c/fastllama.h:46-218).  Used with oracle/_ref/pyfastllama.so (the reference)
and with fastllama_amd/libfastllama_hip.so (ours): same structs, same calls -- that is the drop-in claim."""
from __future__ import annotations

import ctypes as C
import os

LOG_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_int, C.c_char_p, C.c_int)
RESET_FN = C.CFUNCTYPE(None)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint8, C.c_size_t, C.c_size_t)
STREAM_FN = C.CFUNCTYPE(None, C.POINTER(C.c_char), C.c_int)


class Logger(C.Structure):                                   # struct llama_logger, fastllama.h:30-36
    _fields_ = [("log", LOG_FN), ("log_err", LOG_FN), ("log_warn", LOG_FN), ("reset", RESET_FN),
                ("progress", PROGRESS_FN)]


class ArrayViewF(C.Structure):                               # struct llama_array_view_f, fastllama.h:39-42
    _fields_ = [("data", C.POINTER(C.c_float)), ("size", C.c_size_t)]


class ContextArgs(C.Structure):                              # struct llama_model_context_args, fastllama.h:46-61
    _fields_ = [("embedding_eval_enabled", C.c_bool), ("should_get_all_logits", C.c_bool), ("use_mmap", C.c_bool),
                ("use_mlock", C.c_bool), ("load_parallel", C.c_bool), ("seed", C.c_int), ("n_keep", C.c_int),
                ("n_ctx", C.c_int), ("n_threads", C.c_int), ("n_batch", C.c_int),
                ("n_load_parallel_blocks", C.c_uint32), ("last_n_tokens", C.c_size_t),
                ("allocate_extra_mem", C.c_size_t), ("logger", Logger)]


class LlamaLib:
    def __init__(self, path: str):
        self.lib = L = C.CDLL(path)
        L.llama_create_default_context_args.restype = ContextArgs
        L.llama_create_context.restype = C.c_void_p
        L.llama_create_context.argtypes = [ContextArgs]
        for n in ("llama_load_model", "llama_ingest", "llama_ingest_system_prompt", "llama_save_state",
                  "llama_load_state", "llama_attach_lora"):
            getattr(L, n).restype = C.c_bool
            getattr(L, n).argtypes = [C.c_void_p, C.c_char_p]
        L.llama_set_stop_words.restype = C.c_bool
        L.llama_set_stop_words.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t]
        L.llama_generate.restype = C.c_bool
        L.llama_generate.argtypes = [C.c_void_p, STREAM_FN, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float]
        L.llama_perplexity.restype = C.c_float
        L.llama_perplexity.argtypes = [C.c_void_p, C.c_char_p]
        for n in ("llama_get_embeddings", "llama_get_logits"):
            getattr(L, n).restype = ArrayViewF
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("llama_detach_lora", "llama_reset_model"):
            getattr(L, n).restype = C.c_bool
            getattr(L, n).argtypes = [C.c_void_p]
        L.llama_free_context.restype = None
        L.llama_free_context.argtypes = [C.c_void_p]
        L.llama_handle_signal.restype = None
        L.llama_handle_signal.argtypes = [C.c_int]


class Session:
    """One llama_model_context (the reference's fastllama.Model, interfaces/python/fastllama.py:194-479)."""

    def __init__(self, lib: LlamaLib, path: str, n_ctx=512, n_batch=16, n_threads=4, all_logits=False,
                 embeddings=False, seed=0, n_keep=200, last_n_tokens=64, quiet=True, use_mmap=False, extra_mem=0):
        self.L = lib.lib
        args = self.L.llama_create_default_context_args()
        args.embedding_eval_enabled = embeddings
        args.should_get_all_logits = all_logits
        args.use_mmap = use_mmap
        args.use_mlock = False
        args.load_parallel = False
        args.seed, args.n_keep, args.n_ctx, args.n_threads, args.n_batch = seed, n_keep, n_ctx, n_threads, n_batch
        args.last_n_tokens = last_n_tokens
        args.allocate_extra_mem = extra_mem      # the reference's own knob: bytes added to the eval context's memory pool (lib/llama.cpp:173)
        self.log = []
        self._cbs = (LOG_FN(lambda f, fl, m, ml: self.log.append(("I", m[:ml]))),
                     LOG_FN(lambda f, fl, m, ml: self.log.append(("E", m[:ml]))),
                     LOG_FN(lambda f, fl, m, ml: self.log.append(("W", m[:ml]))),
                     RESET_FN(lambda: None), PROGRESS_FN(lambda t, d, tot: None))
        if quiet:
            args.logger = Logger(*self._cbs)
        # the reference prints an ASCII-art banner straight to stdout from llama_create_context: mute fd 1
        import sys
        sys.stdout.flush()
        saved, devnull = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            self.ctx = self.L.llama_create_context(args)
            ok = bool(self.ctx) and bool(self.L.llama_load_model(self.ctx, os.fsencode(path)))
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        if not self.ctx:
            raise RuntimeError("llama_create_context returned NULL")
        if not ok:
            raise RuntimeError("llama_load_model failed: %r" % (self.log[-3:],))

    def perplexity(self, text: str) -> float:
        return float(self.L.llama_perplexity(self.ctx, text.encode()))

    def logits(self):
        import numpy as np
        v = self.L.llama_get_logits(self.ctx)
        return np.ctypeslib.as_array(v.data, shape=(v.size,)).copy() if v.size else np.zeros(0, np.float32)

    def embeddings(self):
        import numpy as np
        v = self.L.llama_get_embeddings(self.ctx)
        return np.ctypeslib.as_array(v.data, shape=(v.size,)).copy() if v.size else np.zeros(0, np.float32)

    def ingest(self, text: str, system=False) -> bool:
        fn = self.L.llama_ingest_system_prompt if system else self.L.llama_ingest
        return bool(fn(self.ctx, text.encode()))

    def generate(self, n_tokens: int, top_k=40.0, top_p=0.95, temp=0.0, repeat_penalty=1.0, stop_words=()):
        out = []
        cb = STREAM_FN(lambda p, n: out.append(C.string_at(p, n)))
        arr = (C.c_char_p * max(1, len(stop_words)))(*[w.encode() for w in stop_words])
        self.L.llama_set_stop_words(self.ctx, arr, len(stop_words))
        ok = self.L.llama_generate(self.ctx, cb, n_tokens, top_k, top_p, temp, repeat_penalty)
        return bool(ok), b"".join(out)

    def save_state(self, path): return bool(self.L.llama_save_state(self.ctx, os.fsencode(path)))
    def load_state(self, path): return bool(self.L.llama_load_state(self.ctx, os.fsencode(path)))
    def reset(self): return bool(self.L.llama_reset_model(self.ctx))
    def attach_lora(self, path): return bool(self.L.llama_attach_lora(self.ctx, os.fsencode(path)))
    def detach_lora(self): return bool(self.L.llama_detach_lora(self.ctx))

    def close(self):
        if self.ctx:
            self.L.llama_free_context(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
