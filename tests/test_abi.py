"""CPU: the C-ABI shared library loads and exports every symbol the headers declare.
No compute is attempted here (there is no GPU in this container and no CPU fallback in the product).
"""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fastllama_amd", "libfastllama_hip.so")


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    names = set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt))
    return {n for n in names if not n.endswith("_t")}


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call([os.path.join(ROOT, "build.sh")])
    return ctypes.CDLL(LIB)


def test_kernel_abi_symbols_exported(lib):
    names = _declared("fastllama_hip.h", "fl_")
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_primary_llama_abi_symbols_exported(lib):
    """include/fastllama.h = the reference's interfaces/c/fastllama.h: all 17 llama_* symbols are exported."""
    names = _declared("fastllama.h", "llama_")
    names = {n for n in names if n not in ("llama_model_context", "llama_model_context_args", "llama_logger",
                                           "llama_array_view_f")}
    assert len(names) == 17, sorted(names)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_llama_context_args_layout_matches_reference_python_binding(lib):
    """sizeof/offsets of llama_model_context_args as ctypes lays it out (fastllama.py:132-151 does the same)."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    from harness import llama_capi
    assert C.sizeof(llama_capi.Logger) == 5 * C.sizeof(C.c_void_p)
    a = llama_capi.ContextArgs
    assert a.seed.offset == 8 and a.n_load_parallel_blocks.offset == 28 and a.last_n_tokens.offset == 32
    assert a.logger.offset == 48 and C.sizeof(a) == 88
    lib.llama_create_default_context_args.restype = a
    d = lib.llama_create_default_context_args()
    assert (d.n_ctx, d.n_batch, d.n_keep, d.n_threads, d.last_n_tokens) == (512, 16, 64, 1, 64)   # bridge.hpp:21-36


def test_python_binding_covers_header():
    from fastllama_amd import hip
    names = _declared("fastllama_hip.h", "fl_")
    assert names == set(hip._PROTOS), names ^ set(hip._PROTOS)
    hip.load()


def test_no_device_means_loud_failure(lib):
    """Without a GPU the product refuses to compute (FL_ENODEV) -- it never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fastllama_amd import hip
    L = hip.load()
    assert L.fl_device_count() == 0
    assert L.fl_init(0) == hip.FL_ENODEV
    assert b"no CPU fallback" in L.fl_last_error()
    assert not L.fl_malloc(64)
    assert not L.fl_qact_create(4, 64)
    with pytest.raises(hip.FastLlamaHipError):
        hip.require_device(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under fastllama_amd/ may reference it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "fastllama_amd")):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"^\s*(import|from)\s+oracle\b|liboracle|orc_|oracle/_ref", src, flags=re.M):
                    bad.append(fn)
    assert not bad, bad
