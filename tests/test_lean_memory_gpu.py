"""GPU: the lean memory mode (fl_model_prepare bit 2).

The reference keeps every weight resident once (lib/llama.cpp:105-270).  The reference-order kernels here read derived copies (WH16 for
N >= 9, QWD for N = 1); lean mode holds exactly the copies a session names and -- with both resident -- frees the QW16 nibble planes that
nothing reads then.  Every logit stays the same bit; the planes come back from QWD (a permutation of nibbles) before anything that reads
them: the fast mode, a LoRA merge, a tensor download."""
import ctypes as C

import numpy as np
import pytest

import oracle
from harness import ggjt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU; there is no CPU fallback to test"
    return torch


def _mem(L, m):
    b = (C.c_size_t * 5)()
    assert L.fl_model_memory(m.h, b) == 0
    return [int(v) for v in b]


@pytest.mark.parametrize("qt", [oracle.Q4_0, oracle.Q4_1])
def test_lean_mode_drops_and_restores_the_nibble_planes_without_changing_a_bit(torch, qt):
    from fastllama_amd import hip
    from harness.flmodel import FlModel
    L = hip.load()
    port = oracle.Port()
    cfg = ggjt.SMALL
    tensors = ggjt.synth_tensors(cfg, qt, port.quantize_q4, seed=11)
    toks = np.random.default_rng(3).integers(3, 259, 48).astype(np.int32)

    def evals(m):
        out = [m.eval(toks[:40], all_logits=True)]                 # N = 40: WH16 (or, without it, the nibble-operand kernel)
        out.append(m.eval(toks[40:45], n_past=40, all_logits=True))  # N = 5: the small-batch kernel / the H16 form while the planes are gone
        out.append(m.eval(toks[45:46], n_past=45))                 # N = 1: QWD
        out.append(m.eval(toks[46:47], n_past=46))
        return [o.copy() for o in out]

    m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=40)
    want = evals(m)
    full = _mem(L, m)
    assert full[0] > 0 and full[2] > 0 and full[3] > 0             # planes, WH16 and QWD all resident after a default session
    m.free()

    # decode-only session: QWD alone, never WH16; the planes stay (the prefill's nibble-operand kernel reads them)
    m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=40)
    assert m.prepare(2 | 4) == 2
    got = evals(m)
    mem = _mem(L, m)
    assert mem[2] == 0 and mem[3] == full[3] and mem[0] == full[0]
    assert L.fl_model_prepared(m.h) == 2
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    m.free()

    # both copies, lean: the nibble planes of the matmul tensors go (the token embeddings keep theirs)
    m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=40)
    assert m.prepare(1 | 2 | 4) == 3
    mem = _mem(L, m)
    assert 0 < mem[0] < full[0] // 4 and mem[1] == full[1] and mem[2] == full[2] and mem[3] == full[3]
    assert L.fl_model_device_bytes(m.h) == sum(mem)
    got = evals(m)
    assert _mem(L, m)[0] == mem[0]                                  # no eval of the reference-order path brought them back
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # a download reads the planes: they come back from QWD, bit for bit the blocks that were uploaded
    name = "layers.1.feed_forward.w2.weight"
    gtype, shape, blocks = tensors[name]
    buf = np.empty_like(blocks)
    hip.check(L.fl_model_tensor_download(m.h, name.encode(), buf.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(buf, blocks)
    assert _mem(L, m)[0] == full[0]
    # ... and the fast mode finds them (its results: whatever the fast mode of a default session gives)
    assert m.prepare(1 | 2 | 4) == 3 and _mem(L, m)[0] == mem[0]   # dropped again
    m.set_exact(False)
    assert _mem(L, m)[0] == full[0]
    fast = m.eval(toks[:40], all_logits=True).copy()
    m.free()
    m = FlModel(cfg, qt, tensors, n_ctx=64, max_batch=40)
    m.set_exact(False)
    assert np.array_equal(m.eval(toks[:40], all_logits=True).view(np.uint32), fast.view(np.uint32))
    m.free()
