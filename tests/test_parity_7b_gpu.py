"""GPU: parity of the device eval with the reference on the configuration the headline number is quoted on
(BASELINE.json configs 2/3: LLaMA-7B, 32 layers, n_batch = 512, Q4_0 and Q4_1), in BOTH modes of the library, and a measured
account of where the fast mode's deviation comes from.

1. test_llama7b_full_model_nbatch512_vs_reference: the same synthetic GGJT file through the reference's own C-ABI
   (oracle/_ref/pyfastllama.so, CPU) and through fl_model (GPU); all 512 x 32000 logits compared.
     * exact mode (fl_model_set_exact, reference-order kernels): every logit BIT-IDENTICAL to the reference -- north_star's
       "within 1e-3" with nothing to spare on either side;
     * fast mode (MFMA kernels, own f32 order): the measured deviation, asserted with margin.
   The numbers go to gpurun_out/parity_7b_<type>_<weights>.json (committed under profiles/).
2. test_teacher_forced_layers_flip_accounting: 7B-width layers fed the ORACLE's layer input (teacher forcing).  Every op
   of a layer is bit-exact given equal inputs except the summation order inside the fast matmuls (1e-7); the reference path
   re-quantizes to int8 before every matmul and rounds to fp16 to index its exp / silu tables, so that 1e-7 either
   vanishes or flips one rounding.  The test counts the flips in the Q8_0 operands of a layer (5-14 per 262 144 quants, each
   one quantum), and checks that the exact mode has none: its layer output equals the oracle's bit for bit.

At this width the reference is bit-identical to itself across batch splits (its f32 dots have no remainder loops; measured
in test 1); on toy widths it is not (tests/test_llama_eval_oracle.py::test_reference_logits_depend_on_batch_split) -- and the
exact mode reproduces that too, because it follows the same per-eval order.
"""
import ctypes as C
import json
import os
import time

import numpy as np
import pytest

import oracle
from oracle import llama_eval as le
from harness import ggjt, llama_capi

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


@pytest.fixture(scope="module")
def reflib():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not shipped")
    return llama_capi.LlamaLib(os.path.join(oracle.REF_DIR, "pyfastllama.so"))


def _metrics(got, want):
    scale = np.max(np.abs(want), axis=1)                           # SURVEY 8c metric: max_i |g - c| / max_i |c| per logits vector
    per_pos = np.max(np.abs(got - want), axis=1) / scale
    return per_pos, float(np.linalg.norm(got - want) / np.linalg.norm(want)), float(np.mean(np.argmax(got, axis=1) == np.argmax(want, axis=1)))


def _ppl(logits, toks, row0=0):
    """exp(mean -log softmax(logits[j])[toks[j+1]]) over the second half of the rows (the reference's window, bridge.cpp:397-407)"""
    n = logits.shape[0]
    nll, cnt = 0.0, 0
    for j in range(n >> 1, n - 1):
        l = logits[j]
        nll -= (l[toks[row0 + j + 1]] - l.max()) - np.log(np.exp(l - l.max()).sum())
        cnt += 1
    return float(np.exp(nll / cnt))


WEIGHTS = {
    # SURVEY.md 8(d) recipe.  sigma = 0.02 makes every matmul expansive (gain sqrt(K) * 0.02 = 1.3 .. 2.1): the 32-layer net
    # amplifies ANY perturbation -- including the reference's own, between two batch splits -- by orders of magnitude.
    "recipe_sigma0.02": 0.02,
    # the same architecture with non-expansive matmuls (gain 0.5): perturbations decay, so what remains is what an
    # implementation itself contributes (summation order + the rounding flips it causes)
    "nonexpansive_sigma0.5_over_sqrtK": lambda K: 0.5 / K ** 0.5,
}


@pytest.mark.parametrize("wname", list(WEIGHTS))
@pytest.mark.parametrize("qtype,tag", [(ggjt.Q4_0, "q4_0"), (ggjt.Q4_1, "q4_1")])
def test_llama7b_full_model_nbatch512_vs_reference(tmp_path_factory, reflib, qtype, tag, wname):
    import torch
    from harness import synth
    from harness.flmodel import FlModel
    cfg = dict(n_vocab=32000, n_embd=4096, n_mult=256, n_head=32, n_layer=32)
    scfg = dict(synth.MODELS["7B"])
    assert ggjt.n_ff_of(cfg["n_embd"], cfg["n_mult"]) == scfg["n_ff"]
    gen = lambda: synth.synth_model_tensors(scfg, qtype, seed=1234, scale=WEIGHTS[wname])
    path = str(tmp_path_factory.mktemp("m7b") / f"llama7b_{tag}.bin")
    ggjt.write_ggjt_stream(path, cfg, qtype, gen())
    rng = np.random.default_rng(7)
    text = bytes(rng.integers(33, 127, size=510).astype(np.uint8)).decode()
    toks = [1] + [b + 3 for b in (" " + text).encode()]           # llama_ingest: BOS + the inserted space + the text (bridge.cpp:193)
    assert len(toks) == 512                                        # = one n_batch = 512 eval, BASELINE.json config 2 / 3
    nthr = min(32, os.cpu_count() or 8)
    # ---- the reference (CPU) through its own C-ABI: one 512-token batch; and the same prompt as two 256-token batches ----
    t0 = time.time()
    # (llama_ingest leaves the last batch pending; the first step of llama_generate evaluates it: lib/bridge.cpp:186-238)
    ref = llama_capi.Session(reflib, path, n_ctx=1024, n_batch=512, n_threads=nthr, all_logits=True)
    assert ref.ingest(text) and ref.generate(1, temp=0.0)[0]
    t_ref = time.time() - t0
    want = ref.logits().reshape(512, cfg["n_vocab"]).astype(np.float64)
    ref.close()
    ref = llama_capi.Session(reflib, path, n_ctx=1024, n_batch=256, n_threads=nthr, all_logits=True)
    assert ref.ingest(text) and ref.generate(1, temp=0.0)[0]
    want_split = ref.logits().reshape(256, cfg["n_vocab"]).astype(np.float64)     # rows 256..511 (the second batch, n_past = 256)
    ref.close()
    os.remove(path)
    # ---- this library (GPU), same tensors, same tokens ----
    m = FlModel(scfg, qtype, gen(), n_ctx=512, max_batch=512)
    m.set_exact(False)
    got = m.eval(toks, n_past=0, all_logits=True).astype(np.float64)
    m.set_exact(True)
    t0 = time.time()
    got_exact = m.eval(toks, n_past=0, all_logits=True)
    t_exact = time.time() - t0
    m.free()
    torch.cuda.empty_cache()
    want32 = want.astype(np.float32)
    n_bits_differ = int((got_exact.view(np.uint32) != want32.view(np.uint32)).sum())
    xp, x_l2, x_greedy = _metrics(got_exact.astype(np.float64), want)
    per_pos, rel_l2, greedy = _metrics(got, want)
    pp2, rel_l2_half, greedy_half = _metrics(got[256:], want[256:])
    sp, self_l2, self_greedy = _metrics(want_split, want[256:])   # the reference against itself
    ppl_ref, ppl = _ppl(want, toks), _ppl(got, toks)
    rec = dict(config=f"LLaMA-7B {tag.upper()}, 32 layers, n_batch 512, synthetic weights '{wname}' quantized by "
                      "quantize_row_q_reference, random printable-ASCII prompt of 512 tokens",
               reference="oracle/_ref/pyfastllama.so (the reference compiled in place): llama_ingest + llama_generate(1) + llama_get_logits",
               metric="per position: max_i |g_i - c_i| / max_i |c_i| over the 32000 logits (SURVEY.md 8c)",
               exact_mode_vs_reference=dict(logits_with_different_bits=n_bits_differ, of=int(want32.size), max=float(xp.max()), rel_l2=x_l2,
                                            frac_positions_within_1e3=float(np.mean(xp <= 1e-3)), greedy_token_agreement=x_greedy,
                                            seconds_eval512_incl_logits_copy=round(t_exact, 3)),
               gpu_vs_reference=dict(position0=float(per_pos[0]), max=float(per_pos.max()), median=float(np.median(per_pos)),
                                     rel_l2=rel_l2, frac_positions_within_1e3=float(np.mean(per_pos <= 1e-3)),
                                     frac_positions_within_1e2=float(np.mean(per_pos <= 1e-2)), greedy_token_agreement=greedy,
                                     perplexity_reference=ppl_ref, perplexity_gpu=ppl),
               positions_256_511=dict(
                   gpu_vs_reference=dict(max=float(pp2.max()), median=float(np.median(pp2)), rel_l2=rel_l2_half, greedy=greedy_half),
                   reference_2x256_vs_reference_1x512=dict(max=float(sp.max()), median=float(np.median(sp)), rel_l2=self_l2,
                                                           greedy=self_greedy)),
               reference_seconds_load_plus_eval512=round(t_ref, 1), reference_threads=nthr)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"parity_7b_{tag}_{wname.split('_')[0]}.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    # exact mode: the reference's logits, bit for bit (north_star: within 1e-3 at every position)
    assert n_bits_differ == 0 and xp.max() == 0.0, (n_bits_differ, float(xp.max()))
    # fast mode -- what a user observes agrees in both regimes; the logits bounds are the measured ones with margin (profiles/r02_parity_7b.json):
    # ~10 one-quantum flips per position on the way through 32 layers (test_teacher_forced_layers_flip_accounting counts
    # them per stage), each worth ~1e-3 of max|x|, decaying in the non-expansive net and amplified in the expansive one.
    # (At this width the reference is bit-identical to itself across batch splits -- no remainder loops in its f32 dots --
    #  so its self-deviation, printed above, is no yardstick here; test_deviation_floor_of_a_reordered_cpu_implementation is.)
    assert abs(ppl - ppl_ref) / ppl_ref < (2e-3 if wname.startswith("nonexpansive") else 1e-2), (ppl, ppl_ref)
    # (round 4: the bounds sit at the measured order -- max 1.8e-2 / 2.0e-2, rel-L2 1.1e-2 / 1.2e-2, greedy 99.2 % / 98.4 % non-expansive;
    #  max 9.8e-2 / 1.0e-1, rel-L2 7.0e-2 / 7.2e-2, greedy 83 % / 86 % with the recipe weights, profiles/r03_parity_7b.json --
    #  they characterise the opt-in fast mode; the contract is the exact-mode assertion above)
    if wname.startswith("nonexpansive"):
        assert per_pos.max() <= 2.5e-2 and rel_l2 <= 1.5e-2 and greedy >= 0.97, (per_pos.max(), rel_l2, greedy)
    else:
        assert per_pos.max() <= 0.12 and rel_l2 <= 0.09 and greedy >= 0.78, (per_pos.max(), rel_l2, greedy)


def test_oracle_reference_and_exact_gpu_agree_at_7b_width(tmp_path_factory, reflib):
    """Three implementations of the same arithmetic on a 7B-WIDTH model (4 layers to bound the numpy oracle's run time, real
    vocabulary, 96 tokens in one batch): the reference (CPU, its own C-ABI), the pinned numpy oracle (oracle/llama_eval.py -- the
    teacher of test_teacher_forced_layers_flip_accounting) and this library's exact mode (GPU) return the same bits; the fast
    mode's distance from them is recorded.  (Round 2 measured here how far a re-ordered CPU implementation lands from the
    reference -- the oracle then summed its attention dots in numpy's order and sat 6e-2 away on the 32-layer model; with the
    reference's own order in those dots it is at 0.)"""
    import torch
    from harness import synth
    from harness.flmodel import FlModel
    port = oracle.Port()
    qtype, N = ggjt.Q4_0, 96
    cfg = dict(n_vocab=32000, n_embd=4096, n_mult=256, n_head=32, n_layer=4)
    scfg = dict(synth.MODELS["7B"], n_layer=4)
    tensors = {name: (g, shape, data.cpu().numpy()) for name, (g, shape, data) in synth.synth_model_tensors(scfg, qtype, seed=1234)}
    torch.cuda.empty_cache()
    path = str(tmp_path_factory.mktemp("m7b") / "llama7b_4l.bin")
    ggjt.write_ggjt(path, cfg, qtype, tensors)
    rng = np.random.default_rng(11)
    text = bytes(rng.integers(33, 127, size=N - 2).astype(np.uint8)).decode()
    toks = [1] + [b + 3 for b in (" " + text).encode()]
    ref = llama_capi.Session(reflib, path, n_ctx=128, n_batch=N, n_threads=min(32, os.cpu_count() or 8), all_logits=True)
    assert ref.ingest(text) and ref.generate(1, temp=0.0)[0]
    want = ref.logits().reshape(N, cfg["n_vocab"]).copy()
    ref.close()
    os.remove(path)
    t0 = time.time()
    orc, _ = le.eval_tokens(le.Weights(cfg, qtype, tensors), le.KV(cfg["n_layer"], N, cfg["n_embd"]), toks, 0, port)
    t_orc = time.time() - t0
    m = FlModel(cfg, qtype, tensors, n_ctx=128, max_batch=N)
    m.set_exact(True)
    got_x = m.eval(toks, n_past=0, all_logits=True)
    m.set_exact(False)
    got = m.eval(toks, n_past=0, all_logits=True).astype(np.float64)
    m.free()
    pp, l2, gr = _metrics(got, want.astype(np.float64))
    rec = dict(config="LLaMA-7B width, 4 layers, Q4_0, 96 tokens in one batch, SURVEY 8d recipe weights (sigma 0.02)",
               oracle_vs_reference_bits_differing=int((orc.view(np.uint32) != want.view(np.uint32)).sum()),
               exact_gpu_vs_reference_bits_differing=int((got_x.view(np.uint32) != want.view(np.uint32)).sum()),
               fast_gpu_vs_reference=dict(max=float(pp.max()), median=float(np.median(pp)), position0=float(pp[0]), rel_l2=l2, greedy=gr),
               oracle_seconds=round(t_orc, 1))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_7b_width_three_way.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    assert rec["oracle_vs_reference_bits_differing"] == 0 and rec["exact_gpu_vs_reference_bits_differing"] == 0, rec
    assert l2 <= 5e-2, rec          # fast mode: the measured order (2e-2 after 4 layers at this width), with margin


def _q8_rows(port, x):
    return np.stack([port.quantize_row_q8_0(r) for r in x])


def _flips(got_blocks, want_blocks):
    """block_q8_0 rows (d f32, s f32, 32 int8): (#quants that differ, max |dq|, #blocks whose scale d differs, #quants)"""
    g = got_blocks.reshape(got_blocks.shape[0], -1, 40)
    w = want_blocks.reshape(want_blocks.shape[0], -1, 40)
    gq, wq = g[:, :, 8:].view(np.int8).astype(np.int32), w[:, :, 8:].view(np.int8).astype(np.int32)
    dq = np.abs(gq - wq)
    dd = (g[:, :, :4].copy().view(np.uint32) != w[:, :, :4].copy().view(np.uint32)).sum()
    return int((dq != 0).sum()), int(dq.max()), int(dd), gq.size


def _relmax(got, want):
    return float(np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want)))


@pytest.mark.parametrize("qtype,tag", [(ggjt.Q4_0, "q4_0"), (ggjt.Q4_1, "q4_1")])
def test_teacher_forced_layers_flip_accounting(qtype, tag):
    """Every stage of a 7B-width layer on the GPU, each fed the ORACLE's input of that stage:
         norm+Q8_0 (bit-exact) -> wq|wk|wv+rope (1e-5) -> attention -> Q8_0 [flips] -> wo+residual (1e-5) -> norm+Q8_0 (bit-exact)
         -> w1|w3 + silu*mul -> Q8_0 [flips] -> w2+residual (1e-5)
       so the only places where the GPU and the oracle can part ways by more than summation round-off are the two stages
       that round an order-dependent f32 value to something discrete; there the differing quants are COUNTED, and each is one
       quantum.  The last number per layer is the whole layer run end to end from the oracle's layer input: the cascade
       those flips cause inside a single layer."""
    import torch
    from fastllama_amd import hip, ops
    from harness.flmodel import FlModel
    port = oracle.Port()
    L = hip.load()
    cfg = dict(n_vocab=512, n_embd=4096, n_mult=256, n_head=32, n_layer=3)     # 7B-width layers, small vocabulary
    E, H, N, n_ctx = cfg["n_embd"], cfg["n_head"], 64, 128
    D, F = E // H, ggjt.n_ff_of(E, cfg["n_mult"])
    bs = 20 if qtype == ggjt.Q4_0 else 24
    tensors = ggjt.synth_tensors(cfg, qtype, port.quantize_q4, seed=99, scale=0.02)
    w = le.Weights(cfg, qtype, tensors)
    toks = ggjt.text_tokens("".join(chr(32 + (7 * i) % 90) for i in range(N - 1)))
    m = FlModel(cfg, qtype, tensors, n_ctx=n_ctx, max_batch=N)
    kv = le.KV(cfg["n_layer"], n_ctx, E)
    tabs = np.empty((2, 1 << 16), np.uint16)
    L.fl_debug_tables(tabs[0].ctypes.data_as(C.c_void_p), tabs[1].ctypes.data_as(C.c_void_p))
    exp_d, silu_d = (torch.from_numpy(t.view(np.int16)).cuda() for t in tabs)
    rt = np.empty((n_ctx, D // 2, 2), np.float32)
    L.fl_debug_rope_table(rt.ctypes.data_as(C.c_void_p), n_ctx, D)
    rope_d = torch.from_numpy(rt).cuda()
    keep = []                                         # device inputs must outlive the (asynchronous) kernels that read them

    def dev(a):
        keep.append(torch.from_numpy(np.ascontiguousarray(a)).cuda())
        return keep[-1]

    def qact(x_f32):                                  # QA16 workspace holding Q8_0(x) -- bit-exact quantizer (test_kernels_gpu)
        keep.append(ops.QAct(x_f32.shape[0], x_f32.shape[1]).quantize(dev(x_f32), layout=16))
        return keep[-1]

    x = np.stack([port.dequantize_row(qtype, w.q("tok_embeddings.weight")[t], E) for t in toks])
    layers = []
    for il in range(cfg["n_layer"]):
        p = f"layers.{il}."
        x_next, mid = le.layer_forward(w, kv, il, x, 0, port)                  # the oracle's trajectory (teacher)
        rec = dict(layer=il)
        # S1: attention_norm -> Q8_0, bit-exact
        cur = le.rms_norm_mul(x, w.f(p + "attention_norm.weight"))
        a1 = ops.QAct(N, E)
        hip.check(L.fl_quantize_q8_layout(a1.handle, dev(cur).data_ptr(), E, N, E, 16, None))           # bookkeeping (N, layout)
        hip.check(L.fl_debug_rmsnorm_quant(dev(x).data_ptr(), E, dev(w.f(p + "attention_norm.weight")).data_ptr(), N, E, None, 0, a1.handle,
                                           16, None))
        a1.N, a1.K = N, E
        assert np.array_equal(a1.export().cpu().numpy(), _q8_rows(port, cur))
        aref = a1
        # S2: wq|wk|wv + rope + KV stores on the oracle's Q8_0 input: agree to summation round-off
        wqkv = np.concatenate([w.q(p + f"attention.w{c}.weight") for c in "qkv"])
        Wqkv = ops.QTensor(qtype, wqkv, 3 * E, E)
        y = torch.zeros((N, 3 * E), device="cuda")
        kc, vc = torch.zeros((n_ctx, E), device="cuda"), torch.zeros((E, n_ctx), device="cuda")
        hip.check(L.fl_debug_gemm_qkv(Wqkv.handle, aref.handle, y.data_ptr(), 3 * E, rope_d.data_ptr(), kc.data_ptr(), vc.data_ptr(), E, D, 0,
                                      n_ctx, None))
        q_or = le.rope(port.mul_mat_q(qtype, w.q(p + "attention.wq.weight"), cur), 0, H)
        rec["qkv_rope_err"] = max(_relmax(y[:, :E].cpu().numpy(), q_or), _relmax(kc[:N].cpu().numpy(), kv.k[il, :N]),
                                  _relmax(vc[:, :N].cpu().numpy().T, kv.v[il, :N]))
        assert rec["qkv_rope_err"] <= 1e-5
        # S3: attention on the ORACLE's roped q / K / V -> Q8_0 of the result: the first discrete stage
        qkv_or = np.zeros((N, 3 * E), np.float32)
        qkv_or[:, :E] = q_or
        kc_or, vc_or = np.zeros((n_ctx, E), np.float32), np.zeros((E, n_ctx), np.float32)
        kc_or[:N], vc_or[:, :N] = kv.k[il, :N], kv.v[il, :N].T
        a3 = ops.QAct(N, E)
        hip.check(L.fl_quantize_q8_layout(a3.handle, dev(mid["att"]).data_ptr(), E, N, E, 16, None))     # bookkeeping; overwritten below
        scale = float(np.float32(1.0) / np.sqrt(np.float32(E) / np.float32(H)))
        ao = torch.zeros((N, E), device="cuda")
        hip.check(L.fl_debug_prefill_attention(dev(qkv_or).data_ptr(), 3 * E, D, H, N, 0, n_ctx, E, dev(kc_or).data_ptr(),
                                               dev(vc_or).data_ptr(), exp_d.data_ptr(), scale, ao.data_ptr(), E, a3.handle, None))
        a3.N, a3.K = N, E
        nf, step, nd, n = _flips(a3.export().cpu().numpy(), _q8_rows(port, mid["att"]))
        rec["attention_q8"] = dict(quants=n, differing=nf, max_step=step, scales_differing=nd)
        assert step <= 1 and nf <= 64, rec                 # measured 5-14 of 262 144 (profiles/r02_parity_layers.json)
        # S4: wo + residual on the oracle's attention output
        Wo = ops.QTensor(qtype, w.q(p + "attention.wo.weight"), E, E)
        x2 = torch.empty((N, E), device="cuda")
        hip.check(L.fl_debug_mul_mat_q_resid(Wo.handle, qact(mid["att"]).handle, x2.data_ptr(), E, dev(x).data_ptr(), E, None))
        x2_or = (port.mul_mat_q(qtype, w.q(p + "attention.wo.weight"), mid["att"]) + x).astype(np.float32)
        rec["wo_err"] = _relmax(x2.cpu().numpy(), x2_or)
        assert rec["wo_err"] <= 1e-5
        # S5: ffn_norm -> Q8_0 on the oracle's x2: bit-exact
        a5 = ops.QAct(N, E)
        hip.check(L.fl_quantize_q8_layout(a5.handle, dev(mid["ffn_in"]).data_ptr(), E, N, E, 16, None))
        hip.check(L.fl_debug_rmsnorm_quant(dev(x2_or).data_ptr(), E, dev(w.f(p + "ffn_norm.weight")).data_ptr(), N, E, None, 0, a5.handle, 16, None))
        a5.N, a5.K = N, E
        assert np.array_equal(a5.export().cpu().numpy(), _q8_rows(port, mid["ffn_in"]))
        # S6: woven w1|w3 matmul + silu*mul -> Q8_0: the second discrete stage (fp16 silu index, Q8_0 rounding)
        w1, w3 = w.q(p + "feed_forward.w1.weight"), w.q(p + "feed_forward.w3.weight")
        woven = np.stack([w1.reshape(F // 16, 16, -1), w3.reshape(F // 16, 16, -1)], axis=1).reshape(2 * F, -1)
        W13 = ops.QTensor(qtype, woven, 2 * F, E)
        a6 = ops.QAct(N, F)
        hip.check(L.fl_debug_gemm_silu(W13.handle, a5.handle, silu_d.data_ptr(), a6.handle, None))
        a6.N, a6.K = N, F
        nf, step, nd, n = _flips(a6.export().cpu().numpy(), _q8_rows(port, mid["act"]))
        rec["silu_q8"] = dict(quants=n, differing=nf, max_step=step, scales_differing=nd)
        assert step <= 1 and nf <= 64, rec                 # measured 5-11 of 704 512
        # S7: w2 + residual on the oracle's activation
        W2 = ops.QTensor(qtype, w.q(p + "feed_forward.w2.weight"), E, F)
        xo = torch.empty((N, E), device="cuda")
        hip.check(L.fl_debug_mul_mat_q_resid(W2.handle, qact(mid["act"]).handle, xo.data_ptr(), E, dev(x2_or).data_ptr(), E, None))
        rec["w2_err"] = _relmax(xo.cpu().numpy(), x_next)
        assert rec["w2_err"] <= 1e-5
        # the whole layer from the oracle's layer input: what the flips of S3 / S6 do to one layer's output
        out = np.empty_like(x)
        hip.check(L.fl_model_debug_layers(m.h, il, il + 1, np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), N, 0,
                                          out.ctypes.data_as(C.c_void_p)), "fl_model_debug_layers")
        rec["whole_layer_err"] = _relmax(out, x_next)
        assert rec["whole_layer_err"] <= 2e-2
        # ... and in exact mode: no flips anywhere, the layer output and its three Q8_0 operands are the oracle's, bit for bit
        m.set_exact(True)
        hip.check(L.fl_model_debug_layers(m.h, il, il + 1, np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), N, 0,
                                          out.ctypes.data_as(C.c_void_p)), "fl_model_debug_layers")
        m.set_exact(False)
        rec["whole_layer_exact_mode_bits_differing"] = int((out.view(np.uint32) != x_next.view(np.uint32)).sum())
        assert rec["whole_layer_exact_mode_bits_differing"] == 0, rec
        layers.append(rec)
        for t in (Wqkv, Wo, W13, W2):
            t.free()
        x = x_next                                                              # teacher forcing
        torch.cuda.synchronize()
        keep.clear()
    m.free()
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f"parity_layers_{tag}.json"), "w") as f:
        json.dump(dict(config=f"7B-width layers ({tag.upper()}), N = {N}, every stage fed the pinned numpy oracle's input", layers=layers), f, indent=1)
    print(json.dumps(layers))
