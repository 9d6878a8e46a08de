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
HOOKS = os.path.join(ROOT, "fastllama_amd", "libfastllama_hip_hooks.so")


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    names = set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt))
    return {n for n in names if not n.endswith("_t")}


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB) or not os.path.exists(HOOKS):
        subprocess.check_call([os.path.join(ROOT, "build.sh")])
    return ctypes.CDLL(LIB, mode=ctypes.RTLD_GLOBAL)


def test_kernel_abi_symbols_exported(lib):
    names = _declared("fastllama_hip.h", "fl_")
    assert len(names) >= 30 and not any(n.startswith("fl_debug_") for n in names)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_test_hooks_live_in_their_own_library(lib):
    """The fl_debug_* hooks (include/fastllama_hip_test.h) are exported by libfastllama_hip_hooks.so, which is linked against the
    product library; the product library exports none of them."""
    names = _declared("fastllama_hip_test.h", "fl_debug_")
    assert len(names) >= 20
    hooks = ctypes.CDLL(HOOKS)
    missing = [n for n in sorted(names) if not hasattr(hooks, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    assert "fl_debug_" not in out
    needed = subprocess.run(["readelf", "-d", HOOKS], capture_output=True, text=True, check=True).stdout
    assert "libfastllama_hip.so" in needed


def test_product_library_exports_its_c_api_and_nothing_else(lib):
    """-fvisibility=hidden + fastllama_amd/csrc/exports.map: the dynamic symbol table holds the 17 llama_* symbols, the fl_* API and
    fl_internal_table (the hook library's one way in) -- no C++ launcher, no kernel handle."""
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TDBVWRi"]
    stray = [s for s in syms if not (s.startswith("llama_") or s.startswith("fl_"))]
    assert not stray, stray[:10]
    assert sum(s.startswith("llama_") for s in syms) == 17
    assert "fl_internal_table" in syms and "fl_model_prepare" in syms and "fl_set_warn_handler" in syms
    und = subprocess.run(["nm", "-D", "--undefined-only", HOOKS], capture_output=True, text=True, check=True).stdout
    assert "_ZN2fl" not in und, "the hook library must reach the product library through fl_internal_table only"


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
    names = _declared("fastllama_hip.h", "fl_") | _declared("fastllama_hip_test.h", "fl_debug_")
    assert names == set(hip._PROTOS), names ^ set(hip._PROTOS)
    L = hip.load()
    L.fl_debug_set                        # (the hook library loads and declares its prototypes on first use)


def test_operator_mode_switch_checks_its_argument():
    """fl_set_op_mode(1 | 0 | -1): reference order / fast kernels / the library default for fl_mul_mat_q*; anything else is refused.
    The default itself follows FL_FAST / FL_EXACT (fl_default_exact)."""
    from fastllama_amd import hip
    L = hip.load()
    assert L.fl_set_op_mode(2) == hip.FL_EINVAL and L.fl_set_op_mode(-2) == hip.FL_EINVAL
    assert b"fl_set_op_mode" in L.fl_last_error()
    for mode in (1, 0, -1):
        assert L.fl_set_op_mode(mode) == hip.FL_OK
    e, f = os.environ.get("FL_EXACT"), os.environ.get("FL_FAST")          # (FL_EXACT wins over FL_FAST; neither: reference order)
    want = (int(e) != 0) if e else (int(f) == 0) if f else 1
    assert L.fl_default_exact() == int(want)


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


def test_unmodified_reference_python_binding_drives_this_library(tmp_path):
    """The reference's own interfaces/python/fastllama.py (imported from /root/reference, unmodified) is pointed at
    libfastllama_hip.so through its `library_path=` parameter: default args, context creation, the logger struct and
    llama_load_model all go through the real ctypes declarations.  Without a GPU the load must fail the way the
    reference reports failures (RuntimeError from the constructor) with this library's reason in the reference's
    Logger; on a GPU box the constructor succeeds."""
    ref_py = "/root/reference/interfaces/python"
    if not os.path.isfile(os.path.join(ref_py, "fastllama.py")):
        pytest.skip("reference tree not present")
    import importlib.util
    import signal
    import torch
    import oracle
    from harness import ggjt
    spec = importlib.util.spec_from_file_location("ref_fastllama", os.path.join(ref_py, "fastllama.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cfg = ggjt.TINY
    path = str(tmp_path / "t.bin")
    ggjt.write_ggjt(path, cfg, ggjt.Q4_0, ggjt.synth_tensors(cfg, ggjt.Q4_0, oracle.Port().quantize_q4, seed=5))

    class Capture(ref.Logger):
        def __init__(self):
            self.lines = []

        def log_info(self, func_name, message):
            self.lines.append(("I", func_name, message))

        def log_err(self, func_name, message):
            self.lines.append(("E", func_name, message))

        def log_warn(self, func_name, message):
            self.lines.append(("W", func_name, message))

        def progress(self, tag, done_size, total_size):
            pass

    log = Capture()
    old = signal.getsignal(signal.SIGINT)
    try:
        if torch.cuda.is_available():
            m = ref.Model(path, num_threads=1, n_ctx=64, n_batch=8, logger=log, library_path=LIB)
            assert m.ingest("hello")
        else:
            with pytest.raises(RuntimeError, match="Unable to load model"):
                ref.Model(path, num_threads=1, n_ctx=64, n_batch=8, logger=log, library_path=LIB)
            assert any(k == "E" and "no CPU fallback" in msg for k, _, msg in log.lines), log.lines
    finally:
        signal.signal(signal.SIGINT, old)


@pytest.mark.parametrize("example", ["perplexity.c", "example.c", "example-alpaca.c"])
def test_reference_c_examples_compile_and_link_against_this_library(tmp_path, example):
    """The reference's C programs (examples/c/*.c, compiled from where they lie, unmodified) build against
    include/fastllama.h and link against libfastllama_hip.so: the header is source-compatible and every symbol they
    use is exported.  Run without a GPU, the program ends at the reference's own `if (!llama_load_model(...)) return 1`."""
    src = os.path.join("/root/reference/examples/c", example)
    if not os.path.isfile(src):
        pytest.skip("reference tree not present")
    exe = str(tmp_path / "prog")
    cc = subprocess.run(["gcc", "-std=gnu11", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", os.path.dirname(LIB),
                         "-l:libfastllama_hip.so", "-Wl,-rpath," + os.path.dirname(LIB), "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
        assert run.returncode == 1, (run.returncode, run.stdout, run.stderr)     # model file absent / no device: load fails


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path only: with no device it exits with a message instead of timing anything else."""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU path" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses) must start two ranks itself: here, without a
    GPU, both ranks come up under torch.distributed.run (RANK / WORLD_SIZE set) and each refuses loudly -- no silent 1-rank run."""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (tests/test_bench_gpu.py covers the launch there)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    out = r.stderr + r.stdout
    assert r.returncode != 0
    # one refusal per rank -- or, when the launcher takes the second rank down the moment the first has failed, one refusal and the launcher's own
    # report of a second local rank
    assert out.count("no CPU path") >= 2 or (out.count("no CPU path") == 1 and "local_rank: 1" in out), out[-2000:]
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


def test_mixed_tile_split_covers_every_row_group_once():
    """Host logic of the GEMM launch with two tile shapes (gemm_q4_mfma32.hip, cfg 116): row groups [0, mg_split) go to whole
    rounds of 256 workgroups of 128x64 tiles, the rest to 128x32 tiles.  Whatever the shape: the two regions partition the row
    groups, region A is a multiple of 128-row tiles and of the 8 XCDs (region B's tile ids are remapped per XCD), and its
    workgroup count is what its rows need -- at most the whole rounds that fit."""
    import ctypes as C
    from fastllama_amd import hip
    L = hip.load()
    for M in (16, 128, 4000, 4096, 11008, 12288, 22016, 27648, 32000, 44032, 100000):
        for N in (9, 16, 48, 64, 100, 128, 256, 500, 512, 1024):
            mgt, ngt = (M + 15) // 16, (N + 15) // 16
            na, split, nb = C.c_int(), C.c_int(), C.c_int()
            assert L.fl_debug_gemm_mixed_split(mgt, ngt, C.byref(na), C.byref(split), C.byref(nb)) == 0
            na, split, nb = na.value, split.value, nb.value
            tn_a, tn_b = (ngt + 3) // 4, (ngt + 1) // 2
            assert 0 <= split <= mgt and (split % 8 == 0 or split == mgt)
            assert na == ((split + 7) // 8) * tn_a if split < mgt else na >= 0
            assert na % 8 == 0 and na <= ((mgt + 7) // 8) * tn_a
            assert nb == ((mgt - split + 7) // 8) * tn_b
            if na:                                               # whole rounds, up to the XCD rounding
                assert na <= (((mgt + 7) // 8) * tn_a) // 256 * 256
    # the case it exists for: LLaMA-7B w1|w3 at n_batch 512 -- 1376 tiles of 128x64 = five whole rounds + 96
    na, split, nb = C.c_int(), C.c_int(), C.c_int()
    L.fl_debug_gemm_mixed_split(22016 // 16, 512 // 16, C.byref(na), C.byref(split), C.byref(nb))
    assert (na.value, split.value * 16, nb.value) == (1280, 20480, 192)
