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


def test_malformed_model_files_do_not_crash_the_loader(tmp_path):
    """The file reader runs before any device work: truncated / corrupted GGJT files must come back as `false` with a
    logged reason, never as a read outside the mapping or a giant allocation (no GPU needed; on the GPU box the same
    cases run in tests/test_llama_api_gpu.py next to an intact load)."""
    import oracle
    from harness import ggjt, llama_capi
    cfg = ggjt.TINY
    tensors = ggjt.synth_tensors(cfg, ggjt.Q4_0, oracle.Port().quantize_q4, seed=77)
    path = str(tmp_path / "t.bin")
    ggjt.write_ggjt(path, cfg, ggjt.Q4_0, tensors)
    blob = open(path, "rb").read()
    cases = {"cut%d" % n: blob[:n] for n in (0, 3, 8, 20, 36, 40, 1000, len(blob) // 2, len(blob) - 1)}
    off = 8 + 7 * 4
    for _ in range(cfg["n_vocab"]):
        off += 4 + int.from_bytes(blob[off:off + 4], "little") + 4
    for name, (a, b, v) in {"zero_dim": (12, 16, 0), "name_len": (4, 8, 0x7FFFFFFF), "n_dims": (0, 4, 9), "type": (8, 12, 77)}.items():
        rec = bytearray(blob)
        rec[off + a:off + b] = v.to_bytes(4, "little")
        cases[name] = bytes(rec)
    for name, at in {"n_vocab": 8, "n_embd": 12, "n_layer": 24}.items():
        rec = bytearray(blob)
        rec[at:at + 4] = (0x7FFFFFFF).to_bytes(4, "little")
        cases[name] = bytes(rec)
    L = llama_capi.LlamaLib(LIB).lib
    for name, data in cases.items():
        f = tmp_path / (name + ".bin")
        f.write_bytes(data)
        msgs = []
        cbs = [llama_capi.LOG_FN(lambda f_, fl, m, ml: msgs.append(m[:ml])) for _ in range(3)]
        cbs += [llama_capi.RESET_FN(lambda: None), llama_capi.PROGRESS_FN(lambda t, d, tot: None)]
        args = L.llama_create_default_context_args()
        args.n_ctx, args.n_batch = 32, 8
        args.logger = llama_capi.Logger(*cbs)
        ctx = L.llama_create_context(args)
        assert ctx
        assert not L.llama_load_model(ctx, os.fsencode(str(f))), name
        assert msgs, name
        L.llama_free_context(ctx)
