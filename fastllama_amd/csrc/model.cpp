// model.cpp -- device-resident replacement of fastllama::Model (weights, KV cache, eval).
//
// Replaces, for the GPU path (all file:line into /root/reference):
//   Model::load  tensor placement        lib/llama.cpp:223-258   -> fl_model_set_tensor / fl_model_finalize
//   KVCacheBuffer (f32 K and V)          lib/llama.cpp:24-51     -> k/v caches in HBM (V stored transposed, as the
//                                                                   reference's own view at lib/llama.cpp:341-343)
//   Model::eval                          lib/llama.cpp:272-499   -> fl_model_eval: one stream of kernels, no host
//                                                                   round trip between ops, logits copied out once
//   ggml_graph_compute dispatch          lib/ggml.c:10811        -> the fixed kernel sequence below
//
// wq|wk|wv and w1|w3 are stacked row-wise into single QW16 tensors at load: rows of a mul_mat are
// independent dots (lib/ggml.c:8140-8163), so one GEMM over the stacked matrix is bit-identical to three
// (two) separate ones and shares the Q8_0 activation quantization, which the reference repeats per matmul.
//
// Tensor parallelism (tp_size > 1) follows the split the reference's loader already knows for the original
// multi-part checkpoints (include/tensor/utils.hpp:93-112): wq/wk/wv/w1/w3 by rows (whole heads per rank),
// wo/w2 by columns (K blocks), so each layer needs two all-reduces of the [N, n_embd] partial sums.
#include <hip/hip_runtime.h>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fastllama_hip.h"
#include "comm.h"
#include "tp_tail.h"
#include "eval_kernels.h"
#include "q4_kernels.h"
#include "runtime.h"
#include "internal.h"

using namespace fl;

// The arithmetic new models (and the operator-level entry points) run by default: 1 = the reference's summation order -- logits
// bit-identical to the reference's x86 build, the mode that honours the llama_eval() contract; the faster kernels with their own
// f32 order (~1e-2 on 7B logits) are the opt-in: FL_FAST=1 in the environment or fl_model_set_exact(m, 0).
#ifndef FL_DEFAULT_EXACT
#define FL_DEFAULT_EXACT 1
#endif

namespace {

struct StagedFuse {            // device AoS staging of a row-stacked tensor (wq|wk|wv or w1|w3)
    void *aos = nullptr;
    int parts_needed = 0;
    unsigned parts_mask = 0;   // which parts have arrived (a file listing one part twice must not complete the tensor)
    int rows_per_part = 0, K = 0;
};

struct Layer {
    float *attn_norm = nullptr, *ffn_norm = nullptr;
    fl_qtensor *wqkv = nullptr, *wo = nullptr, *w13 = nullptr, *w2 = nullptr;
    StagedFuse s_qkv, s_13;
};

}  // namespace

// Everything ONE eval in flight owns: its token ids, activations, Q8_0 operands and the stream its kernels run on.  The model
// has a second set (fl_model::alt, allocated on first use) so that two consecutive chunks of a long prompt can be in flight at
// once (fl_model_ingest).
struct Act {
    int *tok_dev = nullptr;
    float *x = nullptr, *x2 = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *ao = nullptr, *h13 = nullptr,
          *part = nullptr, *logits = nullptr;
    fl_qact qE{}, qEl{}, qF{};                      // K = E, K = E/G (input of wo), K = F/G (input of w2)
    float *pair_ws = nullptr;                       // reference-order decode: where the w1 / w3 workgroups of a feature meet (q4_kernels.h)
    hipStream_t stream = nullptr;
};

struct fl_model : Act {
    fl_model_params hp{};
    int E = 0, H = 0, D = 0, F = 0, V = 0, L = 0, n_ctx = 0, B = 0, qtype = 0;
    int Vl = 0, ldp = 0;        // tensor parallel with n_vocab % tp_size == 0: the lm-head is split by rows (V/G logits per rank,
                                // all-gathered): local rows, row stride of `logits_part`.  Vl == 0: lm-head replicated
    float *logits_part = nullptr, *gather_tmp = nullptr;
    bool tp_graph_failed = false;   // capturing the RCCL collectives into the decode hipGraph failed once: plain launches from then on
    int ldl = 0;                // row stride of `logits` (n_vocab rounded up to 4: 16-byte rows for the GEMM's vector stores)
    int G = 1, rank = 0;        // tensor parallel
    // tp_rows: EVERY matmul split by output rows (wo / w2 too: rows E/G over the full K), the reference's own split across threads
    // (lib/ggml.c:8127-8135): each output is one rank's reference-order dot, nothing is summed across ranks -- the tensor-parallel
    // form of the reference-order mode.  What travels: the Q8_0 operands of wo / w2 and the f32 output rows, by all-gather.
    // !tp_rows: wo / w2 split by K blocks, two all-reduces of partial sums per layer (the fast mode's split).
    bool tp_rows = false;
    fl_qact qFf{};              // tp_rows: the gathered Q8_0 operand of w2 (K = n_ff); the one of wo is gathered into qE
    unsigned char *ag_send = nullptr, *ag_tmp = nullptr;   // tp_rows: one rank's message / the G gathered messages
    // tp_rows decode over the fold region of the communicator (tp_tail.h): the exchanges are the tails of the producing launches
    int fold_state = 0;         // 0 not tried, 1 ready, -1 unavailable (no peer-mapped exchange behind the communicator, or the vectors do not fit)
    TpTail *fold_dev = nullptr; // [4] device records: attention planes, wo rows, silu features, w2 rows
    float *fx = nullptr, *fx2 = nullptr, *fh13 = nullptr;   // the layer input / middle rows and the silu features, full width, in this rank's region
    fl_qact fq{}, fql{};        // the attention output's Q8_0 planes (K = n_embd) there, and the view of this rank's blocks
    size_t graph_nodes = 0;     // kernel nodes of the last captured decode graph (fl_model_graph_nodes)
    int El = 0, Hl = 0, Fl = 0; // local (per rank) widths
    fl_qtensor *tok_emb = nullptr, *output = nullptr;
    float *norm_w = nullptr;
    std::vector<Layer> layers;
    float *kc = nullptr, *vc = nullptr;            // [L][n_ctx][El], [L][El][n_ctx]
    uint16_t *exp_tab = nullptr, *silu_tab = nullptr;
    float *rope_tab = nullptr;                      // [n_ctx][D/2][2]
    // work buffers: the Act base (primary set) and, for pipelined ingests, a second one
    Act alt;
    bool alt_ready = false;
    std::vector<hipEvent_t> kv_ev[2];               // [set][layer]: that set's chunk has stored its K/V rows of the layer
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    fl_comm *comm = nullptr;
    bool finalized = false;
    size_t dev_bytes = 0;
    bool lean = false;           // fl_model_prepare(.., 4): only the named copies, and no QW16 nibble plane beside BOTH derived copies
    bool lean_h16 = false, lean_qwd = false;      // lean mode: which copies fl_model_prepare named
    bool qs_dropped = false;     // the matmul tensors' QW16 nibble planes are gone (restore_qs brings them back from QWD)
    // decode hipGraph
    bool graph_enabled = true;
    bool fuse_decode = true;     // N == 1: norm-in-GEMV + one attention kernel per layer
    bool kv_prefetch = true;     // reference-order decode: the wq|wk|wv launch touches the K / V history the attention reads (fl_model_set_graph bit 9 switches it off: A/B)
    bool fuse_prefill_attn = true;   // N >= 9: KQ + soft_max + KQV in one launch
    int force_deep_attn = 0;         // debugging: always the key-tiled form of that launch
    bool ingest_one_stream = false;  // debugging: fl_model_ingest takes its chunks one after the other
    bool split_eval = false;         // opt-in (measured slower, profiles/r04_split_eval.txt): a prefill eval of >= 256 tokens as two halves on the two streams
    bool exact = false;              // reference-order kernels (exact_kernels.hip): logits bit-identical to the reference's x86 build
    bool w13_il = false;             // w1|w3 woven by 16-row groups (n_ff/tp a multiple of 32): silu epilogue in the matmul
    int h16_state = 0;               // WH16 copies of the matmul weights (reference-order prefill): 0 not built, 1 ready, -1 no memory for them
    int qwd_state = 0;               // QWD copies (reference-order decode), likewise
    bool xh = false;                 // the eval in flight takes the H16 form of the reference-order GEMM
    int exp_tab_n = 0;               // fp16 exp-table entries after 0x8000 that are non-zero (rounded up to 8): the LDS copy
    // LoRA: originals of the tensors an adapter touched (the reference's use_mmap path keeps them too, llama.cpp:868-874)
    struct LoraBackup { fl_qtensor *t; void *qs, *d, *mm; };
    std::vector<LoraBackup> lora_backups;
    hipGraphExec_t graph_exec = nullptr;         // [0] short contexts: one attention launch per layer
    hipGraphExec_t graph_exec_long = nullptr;    // [1] n_past >= split_past: attention split over (head, slice) workgroups
    int split_past = 256;                        // first position that takes the two-launch decode attention
    int *npast_dev = nullptr;
    // live per-kernel timing of the quantized matmuls (bench.py roofline leg)
    bool profile = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t ev_used = 0;
    double prof_mm_ms = 0.0;
    long prof_mm_launches = 0;
};

#define M_HIP(call)                                        \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hip_fail(e_, #call);  \
    } while (0)

static int dev_alloc(fl_model *m, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(model)");
    m->dev_bytes += bytes;
    return FL_OK;
}

static int qact_alloc(fl_model *m, fl_qact *a, int maxN, int K) {
    memset(a, 0, sizeof *a);
    int rc = dev_alloc(m, (void **)&a->q, qact_bytes_q(maxN, K));
    if (rc == FL_OK) rc = dev_alloc(m, (void **)&a->d, qact_bytes_scale(maxN, K));
    if (rc == FL_OK) rc = dev_alloc(m, (void **)&a->s, qact_bytes_scale(maxN, K));
    if (rc == FL_OK && maxN >= 9) {      // XH16 operand of the reference-order prefill GEMM (q4_layout.h)
        rc = dev_alloc(m, (void **)&a->h16, xh16_bytes(maxN, K));
        if (rc == FL_OK) M_HIP(hipMemset(a->h16, 0, xh16_bytes(maxN, K)));
    }
    a->KB = K / FL_QK;
    return rc;
}

static int act_alloc(fl_model *m, Act &a) {      // the work buffers of one eval in flight (a.stream is the caller's business)
    const size_t B = (size_t)m->B, E = (size_t)m->E, El = (size_t)m->El, Fl = (size_t)m->Fl;
    int rc;
    if ((rc = dev_alloc(m, (void **)&a.tok_dev, B * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.x, B * E * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.x2, B * E * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.xn, B * E * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.part, B * E * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.qkv, B * 3 * El * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.att, (size_t)m->Hl * B * (size_t)m->n_ctx * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.ao, B * El * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.h13, B * 2 * Fl * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&a.logits, B * (size_t)m->ldl * 4)) != FL_OK) return rc;
    if ((rc = qact_alloc(m, &a.qE, m->B, m->E)) != FL_OK) return rc;
    if ((rc = qact_alloc(m, &a.qEl, m->B, m->El)) != FL_OK) return rc;
    if ((rc = qact_alloc(m, &a.qF, m->B, m->Fl)) != FL_OK) return rc;
    const size_t wsb = gemv1_llc_pair_ws_bytes(2 * m->Fl);
    if ((rc = dev_alloc(m, (void **)&a.pair_ws, wsb)) != FL_OK) return rc;
    M_HIP(hipMemset(a.pair_ws, 0, wsb));             // (its flags return to zero by themselves after every launch)
    return FL_OK;
}
static void act_free(Act &a) {
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(a.tok_dev); fr(a.x); fr(a.x2); fr(a.xn); fr(a.part); fr(a.qkv); fr(a.att); fr(a.ao); fr(a.h13); fr(a.logits); fr(a.pair_ws);
    for (fl_qact *q : {&a.qE, &a.qEl, &a.qF}) { fr(q->q); fr(q->d); fr(q->s); fr(q->h16); }
    a = Act{};
}

static uint16_t f32_to_f16_bits(float f) {  // round-to-nearest-even, as _cvtss_sh(x, 0) (GGML_FP32_TO_FP16 with F16C)
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float f16_bits_to_f32(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

namespace fl {
// the two 65536-entry fp16 tables of ggml_init (exp and silu of every fp16 value, host libm -- lib/ggml.c:3676-3693; either pointer may
// be NULL) -> the number of entries of exp's [-0, -inf] half up to where it becomes (and stays) zero, rounded up to 8
int build_f16_tables(uint16_t *te, uint16_t *ts) {
    int last_nz = 0;
    for (int i = 0; i < (1 << 16); ++i) {
        const float f = f16_bits_to_f32((uint16_t)i);
        const uint16_t e = f32_to_f16_bits(expf(f));
        if (te) te[i] = e;
        if (ts) ts[i] = f32_to_f16_bits(f / (1.0f + expf(-f)));   // ggml_silu_f32, lib/ggml.c:3196-3198
        if (i >= 0x8000 && i <= 0xFC00 && e != 0) last_nz = i - 0x8000;
    }
    return (last_nz + 1 + 7) & ~7;
}
// [n_ctx][D/2][2] {cos, sin}: theta = p, then theta *= theta_scale per pair -- lib/ggml.c:8655-8667
void build_rope_table(float *rt, int n_ctx, int D) {
    const float theta_scale = powf(10000.0f, -2.0f / (float)D);
    for (int p = 0; p < n_ctx; ++p) {
        float theta = (float)p;
        for (int i = 0; i < D / 2; ++i) {
            float sn, cs;
            sincosf(theta, &sn, &cs);   // the reference's gcc build calls sincosf (merged cosf/sinf)
            rt[((size_t)p * (D / 2) + i) * 2 + 0] = cs;
            rt[((size_t)p * (D / 2) + i) * 2 + 1] = sn;
            theta *= theta_scale;
        }
    }
}
}  // namespace fl

extern "C" {

fl_model *fl_model_create(const fl_model_params *p) {
    if (ensure_device() != FL_OK) return nullptr;
    if (!p || p->n_embd <= 0 || p->n_head <= 0 || p->n_layer <= 0 || p->n_ff <= 0 || p->n_vocab <= 0 || p->n_ctx <= 0 || p->n_ctx % 4 != 0 ||
        p->max_batch <= 0 || (p->qtype != FL_TYPE_Q4_0 && p->qtype != FL_TYPE_Q4_1)) {
        set_error(FL_EINVAL, "fl_model_create: bad hyper-parameters");
        return nullptr;
    }
    const int G = p->tp_size > 0 ? p->tp_size : 1;
    if (p->n_embd % p->n_head || (p->n_embd / p->n_head) % 2 || p->n_embd % 64 || p->n_ff % 64 || p->n_head % G ||
        (p->n_embd / G) % 32 || (p->n_ff / G) % 32 || p->tp_rank < 0 || p->tp_rank >= G) {
        set_error(FL_EINVAL, "fl_model_create: n_embd/n_ff must be multiples of 64 (ggml.c:2372) and divisible by tp_size");
        return nullptr;
    }
    fl_model *m = new fl_model();
    m->hp = *p;
    m->E = p->n_embd; m->H = p->n_head; m->D = m->E / m->H; m->F = p->n_ff; m->V = p->n_vocab; m->L = p->n_layer;
    m->n_ctx = p->n_ctx; m->B = p->max_batch; m->qtype = p->qtype;
    m->G = G; m->rank = p->tp_rank;
    m->El = m->E / G; m->Hl = m->H / G; m->Fl = m->F / G;
    m->w13_il = m->Fl % 32 == 0;
    m->exact = fl_default_exact() != 0;
    m->tp_rows = G > 1 && (getenv("FL_TP_ROWS") ? atoi(getenv("FL_TP_ROWS")) != 0 : m->exact);
    if (G > 1 && m->V % G == 0) { m->Vl = m->V / G; m->ldp = fl_roundup(m->Vl, 4); }
    m->layers.resize(m->L);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error(FL_EHIP, "hipStreamCreate failed");
        delete m;
        return nullptr;
    }
    return m;
}

static int upload_f32(fl_model *m, float **dst, const void *host, size_t n) {
    int rc = dev_alloc(m, (void **)dst, n * 4);
    if (rc != FL_OK) return rc;
    M_HIP(hipMemcpy(*dst, host, n * 4, hipMemcpyDefault));   // host or device source
    return FL_OK;
}

// copy `rows` x blocks [kb0, kb1) of a host AoS tensor (row length KB_full blocks) to a dense device AoS buffer
static int stage_rows(const void *host, int bs, int KB_full, int row0, int rows, int kb0, int kb1, void *dst_dev) {
    const size_t src_pitch = (size_t)KB_full * bs, width = (size_t)(kb1 - kb0) * bs;
    const char *src = (const char *)host + (size_t)row0 * src_pitch + (size_t)kb0 * bs;
    M_HIP(hipMemcpy2D(dst_dev, width, src, src_pitch, width, rows, hipMemcpyDefault));   // host or device source
    return FL_OK;
}

static int make_qtensor(fl_model *m, fl_qtensor **out, const void *aos_dev, int M, int K) {
    *out = fl_qtensor_from_device(m->qtype, aos_dev, M, K, nullptr);
    if (!*out) return FL_EHIP;
    m->dev_bytes += fl_qtensor_device_bytes(*out);
    return FL_OK;
}

// interleave16: the parts are woven by 16-row groups (group 2p = part 0 rows [16p, 16p+16), group 2p+1 = part 1 ...):
// the layout the silu epilogue of the w1|w3 matmul needs (gemm_q4_mfma.hip, GemmSiluEpi)
static int stage_part(fl_model *m, StagedFuse &sf, fl_qtensor **out, int nparts, int part, const void *host, int bs,
                      int KB_full, int row0, int rows, int K, bool interleave16 = false) {
    if (*out) return set_error(FL_EINVAL, "a part of an already complete fused tensor was set again");
    if (sf.parts_mask & (1u << part)) return set_error(FL_EINVAL, "part %d of a fused tensor was set twice", part);
    if (!sf.aos) {
        sf.parts_needed = nparts;
        sf.rows_per_part = rows;
        sf.K = K;
        M_HIP(hipMalloc(&sf.aos, (size_t)nparts * rows * (K / FL_QK) * bs));
    }
    int rc;
    if (interleave16) {
        const size_t rowb = (size_t)(K / FL_QK) * bs;     // (KB_full == K/32: w1 / w3 are never split along K)
        const char *src = (const char *)host + (size_t)row0 * rowb;
        M_HIP(hipMemcpy2D((char *)sf.aos + (size_t)part * 16 * rowb, (size_t)nparts * 16 * rowb, src, 16 * rowb, 16 * rowb,
                          (size_t)rows / 16, hipMemcpyDefault));
        rc = FL_OK;
    } else {
        void *dst = (char *)sf.aos + (size_t)part * rows * (K / FL_QK) * bs;
        rc = stage_rows(host, bs, KB_full, row0, rows, 0, K / FL_QK, dst);
    }
    if (rc != FL_OK) return rc;
    sf.parts_mask |= 1u << part;
    if (sf.parts_mask == (1u << sf.parts_needed) - 1u) {
        rc = make_qtensor(m, out, sf.aos, nparts * rows, K);
        (void)hipFree(sf.aos);
        sf.aos = nullptr;
    }
    return rc;
}

int fl_model_set_tensor(fl_model *m, const char *name, int type, const void *host, int ne0, int ne1) {
    if (!m || !name || !host) return set_error(FL_EINVAL, "fl_model_set_tensor: null argument");
    if (m->finalized) return set_error(FL_EINVAL, "model already finalized");
    const int E = m->E, F = m->F, V = m->V, G = m->G, r = m->rank;
    const int bs = m->qtype == FL_TYPE_Q4_0 ? 20 : 24;
    auto want_q = [&](int k, int rows) -> int {
        if (type != m->qtype) return set_error(FL_EINVAL, "%s: type %d, model is type %d", name, type, m->qtype);
        if (ne0 != k || ne1 != rows) return set_error(FL_EINVAL, "%s: shape [%d,%d], expected [%d,%d]", name, ne0, ne1, k, rows);
        return FL_OK;
    };
    auto want_f = [&](int k) -> int {
        if (type != 0) return set_error(FL_EINVAL, "%s: norm vectors must be f32", name);
        if (ne0 != k || (ne1 != 1 && ne1 != 0)) return set_error(FL_EINVAL, "%s: bad norm shape", name);
        return FL_OK;
    };
    int rc;
    std::string nm(name);
    if (nm == "tok_embeddings.weight" || nm == "output.weight") {
        if ((rc = want_q(E, V)) != FL_OK) return rc;
        fl_qtensor **dst = nm[0] == 't' ? &m->tok_emb : &m->output;
        if (*dst) return set_error(FL_EINVAL, "%s was set twice", name);
        const bool split = nm[0] == 'o' && m->Vl > 0;      // lm-head: rows [r Vl, (r+1) Vl) of rank r
        const int rows = split ? m->Vl : V, row0 = split ? r * m->Vl : 0;
        void *tmp = nullptr;
        M_HIP(hipMalloc(&tmp, (size_t)rows * (E / FL_QK) * bs));
        rc = stage_rows(host, bs, E / FL_QK, row0, rows, 0, E / FL_QK, tmp);
        if (rc == FL_OK) rc = make_qtensor(m, dst, tmp, rows, E);
        (void)hipFree(tmp);
        return rc;
    }
    if (nm == "norm.weight") {
        if ((rc = want_f(E)) != FL_OK) return rc;
        if (m->norm_w) return set_error(FL_EINVAL, "%s was set twice", name);
        return upload_f32(m, &m->norm_w, host, E);
    }
    int il = -1, off = 0;
    if (sscanf(name, "layers.%d.%n", &il, &off) != 1 || il < 0 || il >= m->L || off == 0)
        return set_error(FL_EINVAL, "unknown tensor name '%s'", name);
    Layer &ly = m->layers[il];
    const std::string sub(name + off);
    if (sub == "attention_norm.weight" || sub == "ffn_norm.weight") {
        float **dstv = sub[0] == 'a' ? &ly.attn_norm : &ly.ffn_norm;
        if ((rc = want_f(E)) != FL_OK) return rc;
        if (*dstv) return set_error(FL_EINVAL, "%s was set twice", name);
        return upload_f32(m, dstv, host, E);
    }
    if (sub == "attention.wq.weight" || sub == "attention.wk.weight" || sub == "attention.wv.weight") {
        if ((rc = want_q(E, E)) != FL_OK) return rc;
        const int part = sub[11] == 'q' ? 0 : sub[11] == 'k' ? 1 : 2;
        return stage_part(m, ly.s_qkv, &ly.wqkv, 3, part, host, bs, E / FL_QK, r * m->El, m->El, E);   // rows of rank r
    }
    if (sub == "feed_forward.w1.weight" || sub == "feed_forward.w3.weight") {
        if ((rc = want_q(E, F)) != FL_OK) return rc;
        const int part = sub[14] == '1' ? 0 : 1;
        return stage_part(m, ly.s_13, &ly.w13, 2, part, host, bs, E / FL_QK, r * m->Fl, m->Fl, E, m->w13_il);
    }
    if (sub == "attention.wo.weight" || sub == "feed_forward.w2.weight") {
        const bool is_wo = sub[0] == 'a';
        const int K = is_wo ? E : F, Kl = K / G;
        if ((rc = want_q(K, E)) != FL_OK) return rc;
        if (is_wo ? ly.wo : ly.w2) return set_error(FL_EINVAL, "%s was set twice", name);
        void *tmp = nullptr;
        if (m->tp_rows) {                                   // rows [r E/G, (r+1) E/G) over the full K
            M_HIP(hipMalloc(&tmp, (size_t)m->El * (K / FL_QK) * bs));
            rc = stage_rows(host, bs, K / FL_QK, r * m->El, m->El, 0, K / FL_QK, tmp);
            if (rc == FL_OK) rc = make_qtensor(m, is_wo ? &ly.wo : &ly.w2, tmp, m->El, K);
            (void)hipFree(tmp);
            return rc;
        }
        M_HIP(hipMalloc(&tmp, (size_t)E * (Kl / FL_QK) * bs));
        rc = stage_rows(host, bs, K / FL_QK, 0, E, r * (Kl / FL_QK), (r + 1) * (Kl / FL_QK), tmp);   // K blocks of rank r
        if (rc == FL_OK) rc = make_qtensor(m, is_wo ? &ly.wo : &ly.w2, tmp, E, Kl);
        (void)hipFree(tmp);
        return rc;
    }
    return set_error(FL_EINVAL, "unknown tensor name '%s'", name);
}

int fl_model_finalize(fl_model *m) {
    if (!m) return set_error(FL_EINVAL, "null model");
    if (m->finalized) return FL_OK;
    if (!m->tok_emb || !m->output || !m->norm_w) return set_error(FL_EINVAL, "model is missing tok_embeddings/norm/output");
    for (int l = 0; l < m->L; ++l) {
        const Layer &ly = m->layers[l];
        if (!ly.attn_norm || !ly.ffn_norm || !ly.wqkv || !ly.wo || !ly.w13 || !ly.w2)
            return set_error(FL_EINVAL, "layer %d is missing tensors", l);
    }
    int rc;
    const int E = m->E, El = m->El, Fl = m->Fl, B = m->B, V = m->V, n_ctx = m->n_ctx, D = m->D;
    const size_t kv_elems = (size_t)m->L * n_ctx * El;
    if ((rc = dev_alloc(m, (void **)&m->kc, kv_elems * 4)) != FL_OK) return rc;
    if ((rc = dev_alloc(m, (void **)&m->vc, kv_elems * 4)) != FL_OK) return rc;
    M_HIP(hipMemset(m->kc, 0, kv_elems * 4));
    M_HIP(hipMemset(m->vc, 0, kv_elems * 4));
    // fp16 tables, host libm -- lib/ggml.c:3676-3693
    {
        std::vector<uint16_t> te(1 << 16), ts(1 << 16);
        m->exp_tab_n = build_f16_tables(te.data(), ts.data());
        if ((rc = dev_alloc(m, (void **)&m->exp_tab, 2 << 16)) != FL_OK) return rc;
        if ((rc = dev_alloc(m, (void **)&m->silu_tab, 2 << 16)) != FL_OK) return rc;
        M_HIP(hipMemcpy(m->exp_tab, te.data(), 2 << 16, hipMemcpyHostToDevice));
        M_HIP(hipMemcpy(m->silu_tab, ts.data(), 2 << 16, hipMemcpyHostToDevice));
    }
    // rope table: theta = p, then theta *= theta_scale per pair -- lib/ggml.c:8655-8667
    {
        std::vector<float> rt((size_t)n_ctx * (D / 2) * 2);
        build_rope_table(rt.data(), n_ctx, D);
        if ((rc = dev_alloc(m, (void **)&m->rope_tab, rt.size() * 4)) != FL_OK) return rc;
        M_HIP(hipMemcpy(m->rope_tab, rt.data(), rt.size() * 4, hipMemcpyHostToDevice));
    }
    if ((rc = dev_alloc(m, (void **)&m->npast_dev, 16)) != FL_OK) return rc;
    m->ldl = fl_roundup(V, 4);   // e.g. the 32001-token vocabularies of Alpaca / Vicuna style checkpoints
    if ((rc = act_alloc(m, *m)) != FL_OK) return rc;
    if (m->tp_rows) {
        if ((rc = qact_alloc(m, &m->qFf, m->B, m->F)) != FL_OK) return rc;
        const size_t B16 = (size_t)fl_roundup(B, 16);
        const size_t Km = (size_t)std::max(El, Fl);                                           // the larger Q8_0 message (operand of w2, or of wo when n_embd > n_ff)
        const size_t msg = B16 * Km + 2 * B16 * (Km / FL_QK) * 4;
        const size_t all = std::max((size_t)m->G * msg, std::max((size_t)B * E * 4, (size_t)m->F * 4));
        if ((rc = dev_alloc(m, (void **)&m->ag_send, msg)) != FL_OK) return rc;
        if ((rc = dev_alloc(m, (void **)&m->ag_tmp, all)) != FL_OK) return rc;
    }
    if (m->Vl > 0) {
        if ((rc = dev_alloc(m, (void **)&m->logits_part, (size_t)B * m->ldp * 4)) != FL_OK) return rc;
        if ((rc = dev_alloc(m, (void **)&m->gather_tmp, (size_t)m->G * B * m->ldp * 4)) != FL_OK) return rc;
    }
    M_HIP(hipDeviceSynchronize());
    m->finalized = true;
    return FL_OK;
}

int fl_model_set_comm(fl_model *m, fl_comm *c) {
    if (!m) return set_error(FL_EINVAL, "null model");
    if (m->G > 1 && !c) return set_error(FL_EINVAL, "tensor-parallel model needs a communicator");
    if (m->comm != c) {                      // (the fold records point into the old communicator's regions)
        if (m->fold_dev) { (void)hipFree(m->fold_dev); m->fold_dev = nullptr; }
        m->fold_state = 0;
        if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
        if (m->graph_exec_long) { (void)hipGraphExecDestroy(m->graph_exec_long); m->graph_exec_long = nullptr; }
    }
    m->comm = c;
    return FL_OK;
}

// Row-split tensor-parallel decode over the communicator's fold regions (tp_tail.h): lays the four exchanged vectors out in the region
// and builds the exchange records.  Before any graph capture (it allocates).  fold_local decides for THIS rank (the environment, the region's
// size, an allocation); ensure_fold then has the ranks agree -- a sum over the communicator -- and folds only if every rank can: a rank on the
// collective sequence next to peers spinning in fused tails would hang both.  FL_TP_FOLD=0: the collective sequence (A/B, tests).
static void fold_local(fl_model *m) {
    m->fold_state = -1;
    if (const char *e = getenv("FL_TP_FOLD")) if (e[0] == '0') return;
    TpFold f;
    if (!m->comm || !comm_fold(m->comm, &f) || f.world != m->G || f.rank != m->rank) return;
    const size_t E = (size_t)m->E, F = (size_t)m->G * m->Fl, El = (size_t)m->El, Fl = (size_t)m->Fl, KB = E / FL_QK, KBl = El / FL_QK;
    const size_t off_x = 0, off_x2 = E * 4, off_h = 2 * E * 4, off_q = fl_roundup((int)(off_h + F * 4), 16), off_d = off_q + E, off_s = off_d + KB * 4,
                 total = off_s + KB * 4;
    if (total > TP_FOLD_BYTES) {
        warn("tensor-parallel decode: the exchanged vectors (%zu B) do not fit the communicator's fold region (%zu B): the collective sequence runs "
             "(same results, 9 kernels + 4 collectives per layer instead of 5 launches)", total, TP_FOLD_BYTES);
        return;
    }
    TpTail t[4] = {};
    for (int k = 0; k < 4; ++k) {
        t[k].world = f.world; t[k].rank = f.rank;
        for (int r = 0; r < f.world; ++r) { t[k].region[r] = f.region[r]; t[k].flag[r] = f.flag[r] + k * FL_COMM_MAX_LOCAL; }
        t[k].ticket = f.ticket + k; t[k].epoch = f.epoch + k; t[k].timeouts = f.timeouts;
        t[k].timeout_ticks = p2p_timeout_ticks();          // ~20 s of the 100 MHz wall clock: far beyond any legitimate skew between ranks
        t[k].n_ranges = 1;
    }
    const unsigned r = (unsigned)m->rank;
    t[0].n_ranges = 3;                                     // the attention's Q8_0 blocks of this rank's heads: q | d | s
    t[0].off[0] = (unsigned)(off_q + r * El);      t[0].bytes[0] = (unsigned)El;
    t[0].off[1] = (unsigned)(off_d + r * KBl * 4); t[0].bytes[1] = (unsigned)(KBl * 4);
    t[0].off[2] = (unsigned)(off_s + r * KBl * 4); t[0].bytes[2] = (unsigned)(KBl * 4);
    t[1].off[0] = (unsigned)(off_x2 + r * El * 4); t[1].bytes[0] = (unsigned)(El * 4);      // wo rows (+ residual)
    t[2].off[0] = (unsigned)(off_h + r * Fl * 4);  t[2].bytes[0] = (unsigned)(Fl * 4);      // silu(w1 x) * (w3 x) features
    t[3].off[0] = (unsigned)(off_x + r * El * 4);  t[3].bytes[0] = (unsigned)(El * 4);      // w2 rows (+ residual)
    if (hipMalloc((void **)&m->fold_dev, sizeof t) != hipSuccess || hipMemcpy(m->fold_dev, t, sizeof t, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (m->fold_dev) { (void)hipFree(m->fold_dev); m->fold_dev = nullptr; }
        warn("tensor-parallel decode: no device memory for the exchange records: the collective sequence runs");
        return;
    }
    unsigned char *own = f.region[f.rank];
    m->fx = reinterpret_cast<float *>(own + off_x);
    m->fx2 = reinterpret_cast<float *>(own + off_x2);
    m->fh13 = reinterpret_cast<float *>(own + off_h);
    m->fq = fl_qact{reinterpret_cast<int8_t *>(own + off_q), reinterpret_cast<float *>(own + off_d), reinterpret_cast<float *>(own + off_s), 1, 16, (int)KB, nullptr};
    m->fql = fl_qact{m->fq.q + r * El, m->fq.d + r * KBl, m->fq.s + r * KBl, 1, 16, (int)KBl, nullptr};
    m->fold_state = 1;
}
static void ensure_fold(fl_model *m) {
    if (m->fold_state != 0) return;
    fold_local(m);
    if (!m->comm || m->G < 2) return;
    // COLLECTIVE: every rank of the communicator gets here before its first single-token eval, whatever it decided; the word travels in a work
    // buffer that exists on every rank (nothing that could fail locally sits in front of the collective)
    float mine = m->fold_state > 0 ? 1.f : 0.f, all = 0.f;
    const bool moved = hipMemcpy(m->part, &mine, sizeof mine, hipMemcpyHostToDevice) == hipSuccess;
    if (!moved) (void)hipMemset(m->part, 0, sizeof mine);
    const bool summed = fl_comm_allreduce_sum_f32(m->comm, m->part, 1, m->stream) == FL_OK && hipStreamSynchronize(m->stream) == hipSuccess &&
                        hipMemcpy(&all, m->part, sizeof all, hipMemcpyDeviceToHost) == hipSuccess;
    if (summed && all == (float)m->G) return;              // every rank folds (or, with fold_state < 0 here, this rank would have made the sum smaller)
    if (m->fold_state > 0) {
        warn("tensor-parallel decode: %s: every rank runs the collective sequence (same results)",
             summed ? "another rank cannot fold the exchanges into the decode launches" : "the ranks could not agree on folding the exchanges");
        if (m->fold_dev) { (void)hipFree(m->fold_dev); m->fold_dev = nullptr; }
        m->fold_state = -1;
    }
    (void)hipGetLastError();
}

namespace {
// a producer launch that may carry the exchange as its tail: the record is offered to the launcher (tp_pending_tail) for exactly this call
struct TailOffer {
    explicit TailOffer(const TpTail *t) { tp_pending_tail = t; }
    ~TailOffer() { tp_pending_tail = nullptr; }
    bool taken() const { return tp_pending_tail == nullptr; }
};
}  // namespace
#define M_TAILED(k, call)                                                                              \
    do {                                                                                               \
        bool taken_;                                                                                   \
        hipError_t e_;                                                                                 \
        { TailOffer offer_(m->fold_dev + (k)); e_ = (call); taken_ = offer_.taken(); }                 \
        if (e_ != hipSuccess) return hip_fail(e_, #call);                                              \
        if (!taken_) M_HIP(tp_tail_launch(m->fold_dev + (k), st));      /* (a kernel without the tail: the exchange is a launch of its own) */ \
    } while (0)

// bench hook: bracket one quantized-matmul launch with HIP events on the eval stream
static hipError_t prof_begin(fl_model *m, hipEvent_t *e1) {
    *e1 = nullptr;
    if (!m->profile) return hipSuccess;
    while (m->ev_used + 2 > m->ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return hipErrorOutOfMemory;
        m->ev.push_back(e);
    }
    (void)hipEventRecord(m->ev[m->ev_used++], m->stream);
    *e1 = m->ev[m->ev_used++];
    return hipSuccess;
}
static inline void prof_end(fl_model *m, hipEvent_t e1) {
    if (e1) (void)hipEventRecord(e1, m->stream);
}

static hipError_t mm(fl_model *m, const fl_qtensor *W, const fl_qact &a, int N, float *y, int ldy, const float *resid,
                     int ldr) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    if (m->exact && m->xh && gemm_q4_exact_h16_supports(*W, a, N)) r = gemm_q4_exact_h16(*W, a, N, y, ldy, m->stream, resid, ldr);
    else if (m->exact) r = N == 1 ? gemv_q4_exact(*W, a, N, y, ldy, m->stream, resid, ldr) : gemm_q4_exact(*W, a, N, y, ldy, m->stream, resid, ldr);
    else r = N <= 8 ? gemv_q4(*W, a, N, y, ldy, m->stream, resid, ldr) : gemm_q4_mfma(*W, a, N, y, ldy, m->stream, resid, ldr);
    prof_end(m, e1);
    return r;
}

// decode: y = W . Q8_0(norm_w * rms_norm(x)) in one launch
static hipError_t mm_norm(fl_model *m, const fl_qtensor *W, const float *x, const float *norm_w, float *ynorm, float *y) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    r = (m->exact ? gemv_q4_norm_exact : gemv_q4_norm)(*W, x, norm_w, ynorm, y, m->stream);
    prof_end(m, e1);
    return r;
}

// prefill: the wq|wk|wv matmul with rope + KV-cache stores as its epilogue
static hipError_t mm_qkv_rope(fl_model *m, const fl_qtensor *W, const fl_qact &a, int N, float *kc, float *vc, int n_past) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    r = (m->exact ? gemm_q4_exact_h16_qkv : gemm_q4_mfma_qkv)(*W, a, N, m->qkv, 3 * m->El, m->rope_tab, kc, vc, m->El, m->D, n_past, m->n_ctx, m->stream);
    prof_end(m, e1);
    return r;
}

// prefill: Q8_0(silu(w1 x) * (w3 x)) straight from the w1|w3 matmul (rows woven by 16) into the w2 matmul's operand
static hipError_t mm_silu_gemm(fl_model *m, const fl_qtensor *W, const fl_qact &a, int N, const fl_qact &out) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    r = (m->exact ? gemm_q4_exact_h16_silu : gemm_q4_mfma_silu)(*W, a, N, m->silu_tab, out, m->stream);
    prof_end(m, e1);
    return r;
}

// decode: act = silu(w1 . q) * (w3 . q), q = Q8_0(norm_w * rms_norm(x)) in one launch (woven w1|w3)
static hipError_t mm_norm_silu(fl_model *m, const fl_qtensor *W, const float *x, const float *norm_w, float *act) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    if (m->exact) r = gemv_q4_norm_silu_exact(*W, x, norm_w, m->silu_tab, act, m->stream, m->pair_ws, 0);
    else r = gemv_q4_norm_silu(*W, x, norm_w, m->silu_tab, act, m->stream);
    prof_end(m, e1);
    return r;
}

// decode: y = W . Q8_0(act) (+ resid) in one launch
static hipError_t mm_quant(fl_model *m, const fl_qtensor *W, const float *act, float *y, const float *resid) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    r = (m->exact ? gemv_q4_quant_exact : gemv_q4_quant)(*W, act, y, resid, m->stream);
    prof_end(m, e1);
    return r;
}

// decode: y = W . Q8_0(silu(h13[:F]) * h13[F:]) (+ resid) in one launch
static hipError_t mm_silu(fl_model *m, const fl_qtensor *W, const float *h13, float *y, const float *resid) {
    hipEvent_t e1;
    hipError_t r = prof_begin(m, &e1);
    if (r != hipSuccess) return r;
    r = (m->exact ? gemv_q4_silu_exact : gemv_q4_silu)(*W, h13, m->silu_tab, y, resid, m->stream, m->w13_il);
    prof_end(m, e1);
    return r;
}

// The WH16 copies of every matmul weight (4 x the nibble bytes: 13 GB at 7B of the 288 GB), built on the first reference-order
// eval with N >= 9.  No memory for them: the round-3 kernel, which reads the nibbles, keeps the mode working.
static std::vector<fl_qtensor *> matmul_tensors(fl_model *m) {
    std::vector<fl_qtensor *> ts;
    for (Layer &ly : m->layers) for (fl_qtensor *t : {ly.wqkv, ly.wo, ly.w13, ly.w2}) ts.push_back(t);
    ts.push_back(m->output);
    return ts;
}
// Lean memory mode.  With BOTH derived copies resident nothing in the reference-order path reads a matmul tensor's QW16 nibble
// plane: N >= 2 takes WH16 (the H16 GEMM is exact at any N; run_eval_kernels lowers its threshold from 9 to 2 while the plane is gone), N = 1 takes
// QWD, the scales stay (both kernels read them from the QW16 planes).  lean_drop frees the planes (3.3 GB at 7B); restore_qs rebuilds them from QWD -- a
// permutation of nibbles, lossless -- before anything that does read them: the fast mode, a LoRA merge, a tensor download, a rebuild of a copy.
static int restore_qs(fl_model *m) {
    if (!m->qs_dropped) return FL_OK;
    for (fl_qtensor *t : matmul_tensors(m)) {
        if (t->qs) continue;
        const size_t bytes = (size_t)t->M16 * t->KB * 16;
        uint32_t *qs = nullptr;
        M_HIP(hipMalloc((void **)&qs, bytes));
        const hipError_t e = qwd_to_qw16(*t, qs, m->stream);
        if (e != hipSuccess) { (void)hipFree(qs); return hip_fail(e, "qwd_to_qw16"); }
        t->qs = qs;
        m->dev_bytes += bytes;
    }
    M_HIP(hipStreamSynchronize(m->stream));
    m->qs_dropped = false;
    return FL_OK;
}
static void lean_drop(fl_model *m) {
    if (!m->lean || m->qs_dropped || m->h16_state <= 0 || m->qwd_state <= 0 || !m->lora_backups.empty() || !m->exact) return;
    (void)hipStreamSynchronize(m->stream);
    for (fl_qtensor *t : matmul_tensors(m)) {
        if (!t->qs || !t->h16 || !t->qwd || !t->owns) continue;
        (void)hipFree(t->qs);
        t->qs = nullptr;
        m->dev_bytes -= (size_t)t->M16 * t->KB * 16;
        m->qs_dropped = true;
    }
}

static void ensure_h16(fl_model *m) {
    if (m->h16_state != 0) return;
    if (m->lean && !m->lean_h16) { m->h16_state = -1; return; }      // (lean mode: a copy that was not asked for is never built -- same bits from the nibble-operand kernel)
    std::vector<fl_qtensor *> ts;
    for (Layer &ly : m->layers) for (fl_qtensor *t : {ly.wqkv, ly.wo, ly.w13, ly.w2}) ts.push_back(t);
    ts.push_back(m->output);
    m->h16_state = 1;
    size_t need = 0, built = 0;
    for (fl_qtensor *t : ts) if (!t->h16) need += wh16_bytes(*t);
    for (fl_qtensor *t : ts) {
        if (t->h16) continue;
        if (fl_qtensor_build_h16(t, m->stream) != FL_OK) { m->h16_state = -1; break; }
        m->dev_bytes += wh16_bytes(*t);
        built += wh16_bytes(*t);
    }
    if (m->h16_state < 0) {
        (void)hipGetLastError();
        for (fl_qtensor *t : ts) if (t->h16) { m->dev_bytes -= wh16_bytes(*t); fl_qtensor_drop_h16(t); }
        // same bits, slower kernel: say so (no silent cliffs)
        warn("no device memory for the f16 operand copies of the reference-order prefill GEMM (%.1f GB needed, %.1f GB fit): "
             "prefill runs the nibble-operand kernel (gemm_q4_exact_mfma, same results, ~1.45x the time)", need / 1e9, built / 1e9);
    }
}

// The QWD copies (reference-order decode kernel; the nibbles' size again), built before the first reference-order single-token eval
// -- outside any graph capture.  No memory for them: round 3's producer / chain-wave kernel, which reads QW16, keeps the mode working.
static void ensure_qwd(fl_model *m) {
    if (m->qwd_state != 0) return;
    if (m->lean && !m->lean_qwd) { m->qwd_state = -1; return; }
    std::vector<fl_qtensor *> ts;
    for (Layer &ly : m->layers) for (fl_qtensor *t : {ly.wqkv, ly.wo, ly.w13, ly.w2}) ts.push_back(t);
    ts.push_back(m->output);
    m->qwd_state = 1;
    size_t need = 0, built = 0;
    for (fl_qtensor *t : ts) if (!t->qwd) need += qwd_bytes(*t);
    for (fl_qtensor *t : ts) {
        if (t->qwd) continue;
        if (fl_qtensor_build_qwd(t, m->stream) != FL_OK) { m->qwd_state = -1; break; }
        m->dev_bytes += qwd_bytes(*t);
        built += qwd_bytes(*t);
    }
    if (m->qwd_state < 0) {
        (void)hipGetLastError();
        for (fl_qtensor *t : ts) if (t->qwd) { m->dev_bytes -= qwd_bytes(*t); fl_qtensor_drop_qwd(t); }
        warn("no device memory for the decode copies of the weights (%.1f GB needed, %.1f GB fit): single-token evals run the "
             "producer / chain-wave kernel on the primary layout (same results, ~1.3x the time)", need / 1e9, built / 1e9);
    }
}

static int allreduce_if_tp(fl_model *m, float *buf, size_t count) {
    if (m->G == 1) return FL_OK;
    if (!m->comm) return set_error(FL_EINVAL, "tensor-parallel eval without a communicator");
    return fl_comm_allreduce_sum_f32(m->comm, buf, count, m->stream);
}

// tp_rows: the Q8_0 operand `loc` (K = Kl: the blocks of this rank's features) of every rank -> `full` (K = G Kl), the operand of
// the row-split wo / w2 matmul; xh: also its XH16 copy (the H16 form of the reference-order GEMM)
static int tp_gather_qact(fl_model *m, const fl_qact &loc, const fl_qact &full, int N, int Kl, int layout, bool xh) {
    if (!m->comm) return set_error(FL_EINVAL, "tensor-parallel eval without a communicator");
    const int KBl = Kl / FL_QK, G = m->G;
    const size_t N16 = (size_t)fl_roundup(N, 16);
    const size_t nv = layout == 16 ? N16 : (size_t)N;                          // vectors in the planes (QA1: q [n][KB][32], d / s [n][KB])
    const size_t nq = nv * (size_t)Kl, nd = nv * (size_t)KBl * 4;
    const size_t msg = nq + 2 * nd;
    const int rows = layout == 16 ? (int)(N16 / 16) : N;
    const size_t cq = layout == 16 ? (size_t)KBl * 512 : (size_t)Kl, cd = layout == 16 ? (size_t)KBl * 64 : (size_t)KBl * 4;
    M_HIP(pack3(m->ag_send, loc.q, nq, loc.d, nd, loc.s, nd, m->stream));
    const int rc = fl_comm_allgather_f32(m->comm, reinterpret_cast<const float *>(m->ag_send), msg / 4, reinterpret_cast<float *>(m->ag_tmp), m->stream);
    if (rc != FL_OK) return rc;
    M_HIP(unpack3(m->ag_tmp, msg, G, rows, full.q, cq, full.d, cd, full.s, cd, m->stream));
    if (xh) M_HIP(qa16_to_h16(full, N, m->stream));
    return FL_OK;
}

// tp_rows: out[N][E] = all ranks' output rows part[N][El] (+ resid): all-gather, then the ggml_add that follows wo / w2
static int tp_gather_rows_add(fl_model *m, const float *part, int N, const float *resid, float *out) {
    if (!m->comm) return set_error(FL_EINVAL, "tensor-parallel eval without a communicator");
    const int rc = fl_comm_allgather_f32(m->comm, part, (size_t)N * m->El, reinterpret_cast<float *>(m->ag_tmp), m->stream);
    if (rc != FL_OK) return rc;
    M_HIP(gather_rows_add(reinterpret_cast<const float *>(m->ag_tmp), m->G, N, m->El, resid, m->E, out, m->E, m->stream));
    return FL_OK;
}

// The fixed kernel sequence of one eval (what ggml_graph_compute walks node by node in the reference).  `dyn` != null:
// positions are read from device memory (m->npast_dev) instead of the n_past argument -- the decode hipGraph.
// [l0, l1): the layers to run; body_only: neither the token-embedding lookup before nor the final norm + lm-head after
// (fl_model_debug_layers: the teacher-forced per-layer parity tests feed m->x themselves).
// kv_wait / kv_rec (pipelined ingest, per layer): the stream waits for kv_wait[l] before the layer's attention reads the K/V cache
// (the previous chunk, running on the other stream, has stored its rows), and records kv_rec[l] once this chunk's rows are stored.
static int run_eval_kernels(fl_model *m, int N, int n_past, const int *dyn, bool split_attn = false, int l0 = 0, int l1 = -1,
                            bool body_only = false, bool skip_head = false, const hipEvent_t *kv_wait = nullptr,
                            const hipEvent_t *kv_rec = nullptr, float *logits_dst = nullptr, float *xn_dst = nullptr) {
    const int E = m->E, El = m->El, Fl = m->Fl, D = m->D, Hl = m->Hl, V = m->V, n_ctx = m->n_ctx;
    const int layout = N <= (m->exact ? 1 : 8) ? 1 : 16;    // exact mode: only N = 1 takes the single-vector layout
    const int P = n_past + N;
    hipStream_t st = m->stream;
    const bool tp = m->G > 1;
    const bool exact = m->exact;        // reference-order matmuls: the per-op sequence, every matmul through exact_kernels.hip
    const bool fused = N == 1 && D <= 128 && D % 32 == 0 && E <= 8192 && Fl <= 32768 && m->fuse_decode;   // single-token kernels (both modes)
    const bool fuse_pa = m->fuse_prefill_attn && !exact;
    if (exact && N >= 9 && !dyn) ensure_h16(m);
    // reference-order prefill: the H16 form of the GEMM (gemm_q4_exact_h16.hip) with rope / K-V stores and silu * mul -> Q8_0 as its
    // epilogues; every Q8_0 operand gets its XH16 copy
    // (the fused launchers' own preconditions are part of the decision: a shape they refuse takes the unfused reference-order sequence)
    // (lean mode with the QW16 nibble planes dropped: every N >= 2 takes the H16 form -- exact at any N -- since nothing else can read the weights)
    const bool xh = exact && N >= (m->qs_dropped ? 2 : 9) && !dyn && m->h16_state > 0 && m->qE.h16 && m->qEl.h16 && m->qF.h16 &&
                    D % 4 == 0 && El % 4 == 0 && l0 < m->L && gemm_q4_exact_h16_supports(*m->layers[l0].wqkv, m->qE, N) &&
                    gemm_q4_exact_h16_supports(*m->layers[l0].w13, m->qE, N) && m->layers[l0].w13->M % 64 == 0;
    m->xh = xh;
    if (m->qs_dropped && (!exact || (N >= 2 && !xh))) {      // a path that reads the QW16 nibble planes after all: bring them back (they stay)
        const int rc = restore_qs(m);
        if (rc != FL_OK) return rc;
    }
    // Row-split tensor-parallel decode over the fold region (tp_tail.h): the rows between the launches live in this rank's region, every
    // producer writes its slice there, and the exchange is the tail of the producing launch -- the layer is its five decode launches.
    const bool fold = tp && m->tp_rows && exact && fused && m->w13_il && m->fold_state > 0 && !kv_wait && !kv_rec && !body_only;
    float *inp = fold ? m->fx : m->x, *mid = fold ? m->fx2 : m->x2;
    if (!body_only) M_HIP(get_rows_qw16(*m->tok_emb, dyn ? dyn + 1 : m->tok_dev, N, inp, E, st));       // inpL = get_rows  llama.cpp:304 (a decode graph: its token lies behind its position)
    if (l1 < 0) l1 = m->L;
    // Row-split tensor parallelism, prefill (round 6): per exchange ONE collective and at most one kernel, instead of pack3 ->
    // all-gather -> unpack3 -> qa16_to_h16 and all-gather -> gather_rows_add:
    //   * the producers of the Q8_0 operands (the attention's P.V epilogue, the silu epilogue of w1|w3) write their planes straight into the send
    //     buffer, packed [q | d | s] for THIS eval's columns (tp_send_view): no pack launch;
    //   * one launch turns the G gathered messages into what the consumer GEMM reads -- d / s planes and the XH16 copy (gathered_qa16_to_operand);
    //   * the all-gathered output rows of wo / w2 are NOT added to the residual in a launch of their own: the rms_norm that follows reads them from
    //     the gather buffer, adds, writes the sum back as the next residual and normalises (rmsnorm_quant_gathered) -- `pend` carries the debt.
    // 20 -> 14 graph nodes per layer (10 kernels + 4 collectives); same arithmetic, element for element (tests/test_wide_models_gpu.py).
    const bool tpg = tp && m->tp_rows && exact && layout == 16 && !fused && m->comm && El % 32 == 0 && Fl % 32 == 0;
    struct { bool on; const float *resid; float *out; } pend = {false, nullptr, nullptr};
    auto tp_send_view = [&](int Kl) -> fl_qact {
        const size_t N16 = (size_t)fl_roundup(N, 16), nq = N16 * (size_t)Kl, nd = N16 * (size_t)(Kl / FL_QK) * 4;
        return fl_qact{reinterpret_cast<int8_t *>(m->ag_send), reinterpret_cast<float *>(m->ag_send + nq), reinterpret_cast<float *>(m->ag_send + nq + nd),
                       N, (int)N16, Kl / FL_QK, nullptr};
    };
    const fl_qact sEl = tpg ? tp_send_view(El) : m->qEl, sF = tpg ? tp_send_view(Fl) : m->qF;
    // the operand of a row-split wo / w2 matmul from every rank's blocks of it (in the send buffer), then this rank's rows, then their all-gather
    auto tp_matmul_rows = [&](const fl_qtensor *W, int Kl, const fl_qact &full, const float *resid, float *out) -> int {
        const size_t msg = (size_t)fl_roundup(N, 16) * (size_t)(Kl / FL_QK) * 40;
        int rc = fl_comm_allgather_f32(m->comm, reinterpret_cast<const float *>(m->ag_send), msg / 4, reinterpret_cast<float *>(m->ag_tmp), st);
        if (rc != FL_OK) return rc;
        M_HIP(gathered_qa16_to_operand(m->ag_tmp, msg, m->G, Kl / FL_QK, N, full, !xh, xh, st));
        M_HIP(mm(m, W, full, N, m->part, El, nullptr, 0));
        if ((rc = fl_comm_allgather_f32(m->comm, m->part, (size_t)N * El, reinterpret_cast<float *>(m->ag_tmp), st)) != FL_OK) return rc;
        pend = {true, resid, out};
        return FL_OK;
    };
    // rms_norm * w -> Q8_0 of `x`, which may still be owed the gathered rows + residual
    auto norm_q8 = [&](float *x, const float *w, float *y_f32) -> hipError_t {
        if (pend.on) {
            pend.on = false;
            return rmsnorm_quant_gathered(reinterpret_cast<const float *>(m->ag_tmp), m->G, El, pend.resid, E, x, E, w, N, E, y_f32, E, &m->qE, layout, st, xh);
        }
        return rmsnorm_quant(x, E, w, N, E, y_f32, E, &m->qE, layout, st, xh);
    };
    for (int l = l0; l < l1; ++l) {
        const Layer &ly = m->layers[l];
        float *kc = m->kc + (size_t)l * n_ctx * El, *vc = m->vc + (size_t)l * n_ctx * El;
        if (fold) {
            const size_t r0 = (size_t)m->rank * El;
            M_HIP(mm_norm(m, ly.wqkv, inp, ly.attn_norm, nullptr, m->qkv));
            const float kq_scale = 1.0f / sqrtf((float)E / (float)m->H);
            if (split_attn)
                M_TAILED(0, decode_attention_split(m->qkv, El, D, Hl, n_past, n_ctx, m->rope_tab, kc, vc, m->exp_tab, kq_scale, m->att, &m->fql, st, dyn, exact));
            else
                M_TAILED(0, decode_attention(m->qkv, El, D, Hl, n_past, n_ctx, m->rope_tab, kc, vc, m->exp_tab, kq_scale, &m->fql, st, dyn, exact));
            M_TAILED(1, mm(m, ly.wo, m->fq, 1, mid + r0, El, inp + r0, El));                             // this rank's rows of wo (+ its rows of the residual)
            M_TAILED(2, mm_norm_silu(m, ly.w13, mid, ly.ffn_norm, m->fh13 + (size_t)m->rank * Fl));      // this rank's silu(w1 x) * (w3 x) features
            M_TAILED(3, mm_quant(m, ly.w2, m->fh13, inp + r0, mid + r0));                                // this rank's rows of w2 (+ residual)
            continue;
        }
        if (fused) {
            // decode: norm folded into the matmul, attention in one launch per layer (rope .. KQV .. Q8_0)
            if (exact && dyn && !tp && !split_attn && m->kv_prefetch) gemv1_stream_offer_kv_prefetch(kc, vc, dyn, El, D, Hl, n_ctx);
            M_HIP(mm_norm(m, ly.wqkv, inp, ly.attn_norm, nullptr, m->qkv));
            gemv1_stream_offer_kv_prefetch(nullptr, nullptr, nullptr, 0, 0, 0, 0);      // (a launcher that did not take it must not leave it for the lm-head)
            const float kq_scale = 1.0f / sqrtf((float)E / (float)m->H);
            if (kv_wait) M_HIP(hipStreamWaitEvent(st, kv_wait[l], 0));        // (a one-token chunk of a pipelined ingest)
            if (split_attn)
                M_HIP(decode_attention_split(m->qkv, El, D, Hl, n_past, n_ctx, m->rope_tab, kc, vc, m->exp_tab, kq_scale,
                                             m->att, &m->qEl, st, dyn, exact));
            else
                M_HIP(decode_attention(m->qkv, El, D, Hl, n_past, n_ctx, m->rope_tab, kc, vc, m->exp_tab, kq_scale,
                                       &m->qEl, st, dyn, exact));
            if (kv_rec) M_HIP(hipEventRecord(kv_rec[l], st));
        } else {
            // norm + attention_norm*cur -> Q8_0                                                        llama.cpp:311-319
            M_HIP(norm_q8(inp, ly.attn_norm, nullptr));                                               // (xh: + the XH16 copy)
            if ((N >= 9 && !dyn && fuse_pa) || xh) {
                M_HIP(mm_qkv_rope(m, ly.wqkv, m->qE, N, kc, vc, n_past));                              // wq, wk, wv + rope + KV store
            } else {
                M_HIP(mm(m, ly.wqkv, m->qE, N, m->qkv, 3 * El, nullptr, 0));                           // wq, wk, wv  :328-334
                M_HIP(rope_kv(m->qkv, 3 * El, N, El, D, n_past, n_ctx, m->rope_tab, kc, vc, st, dyn)); // rope, store :328-347
            }
            if (kv_rec) M_HIP(hipEventRecord(kv_rec[l], st));
            if (kv_wait) M_HIP(hipStreamWaitEvent(st, kv_wait[l], 0));
            const float kq_scale = 1.0f / sqrtf((float)E / (float)m->H);
            // KQ, scale, mask, soft_max, KQV in one launch with the score rows in LDS when they fit       :364-398
            hipError_t pe = (N >= 9 && !dyn && fuse_pa)
                                ? prefill_attention(m->qkv, 3 * El, D, Hl, N, n_past, n_ctx, El, kc, vc, m->exp_tab, m->exp_tab_n, kq_scale, m->ao, El, st, &m->qEl,
                                                    m->att, n_ctx, (int64_t)N * n_ctx, m->force_deep_attn)
                                : hipErrorInvalidValue;
            if (pe != hipSuccess) {
                (void)hipGetLastError();
                // KQ, scale, mask, soft_max                                                            :364-379
                const bool xa = exact && !dyn && N >= 2 && D % 32 == 0 && D <= 128;   // the MFMA forms of the exact products
                // (contexts of up to 1024 keys: K.Q and soft_max in one launch, the score rows waiting in LDS)
                // (contexts of 513 .. 2048 keys: the probabilities travel compact -- fp16 table values + one factor per row -- between the launches)
                const bool compact = xa && P > 512 && P <= 2048 && (n_ctx & 3) == 0 && (El & 31) == 0;   // (... and nothing attn_pv_exact could refuse)
                bool softmaxed = false;
                hipError_t xe = xa ? attn_scores_softmax_exact(m->qkv, 3 * El, D, Hl, N, n_past, kc, El, kq_scale, m->att, n_ctx, (int64_t)N * n_ctx, m->exp_tab, st, compact)
                                   : hipErrorInvalidValue;
                if (xe == hipSuccess) softmaxed = true;
                else if (xa) {
                    (void)hipGetLastError();
                    xe = attn_scores_exact(m->qkv, 3 * El, D, Hl, N, n_past, kc, El, kq_scale, m->att, n_ctx, (int64_t)N * n_ctx, st);
                }
                if (xe == hipErrorInvalidValue) {                   // (a shape or alignment outside the MFMA form's reach: the half-wave-per-dot kernel)
                    (void)hipGetLastError();
                    xe = (exact ? dot_f32_abt_exact : gemm_f32_abt)(m->qkv, 3 * El, D, kc, El, D, m->att, n_ctx, (int64_t)N * n_ctx, N, P, D, Hl,
                                                                    kq_scale, 1, n_past, st, dyn, n_ctx);
                }
                M_HIP(xe);
                if (!softmaxed) M_HIP(softmax_rows(m->att, n_ctx, (int64_t)N * n_ctx, N, P, n_past, Hl, m->exp_tab, st, dyn, compact));
                // KQV, merged back to [N, n_embd]                                                      :389-398
                bool quantized = xa && layout == 16 && El % 32 == 0;        // the MFMA form writes the Q8_0 operand of wo itself
                xe = xa ? attn_pv_exact(m->att, n_ctx, (int64_t)N * n_ctx, D, Hl, N, n_past, vc, n_ctx, m->ao, El, st, quantized ? &sEl : nullptr,
                                        xh && !m->tp_rows, compact)
                        : hipErrorInvalidValue;
                if (xe == hipErrorInvalidValue) {
                    (void)hipGetLastError();
                    quantized = false;
                    xe = (exact ? dot_f32_abt_exact : gemm_f32_abt)(m->att, n_ctx, (int64_t)N * n_ctx, vc, n_ctx, (int64_t)D * n_ctx, m->ao, El, D, N,
                                                                    D, P, Hl, 1.0f, 2, n_past, st, dyn, n_ctx);
                }
                M_HIP(xe);
                if (quantized) {}
                else if (layout == 16) M_HIP(quantize_q8_qa16(m->ao, El, N, El, sEl, st, xh && !m->tp_rows));
                else M_HIP(quantize_q8_qa1(m->ao, El, N, El, m->qEl, st));
            }   // (the one-launch kernel wrote the Q8_0 operand of the wo matmul itself)
        }
        // wo projection + residual                                                                 :401-407
        if (!tp) {
            M_HIP(mm(m, ly.wo, m->qEl, N, mid, E, inp, E));
        } else if (tpg) {
            const int rc = tp_matmul_rows(ly.wo, El, m->qE, inp, mid);                              // mid = rows + inp: owed to the ffn norm
            if (rc != FL_OK) return rc;
        } else if (m->tp_rows) {
            int rc = tp_gather_qact(m, m->qEl, m->qE, N, El, layout, xh);                           // the operand, K = n_embd
            if (rc != FL_OK) return rc;
            M_HIP(mm(m, ly.wo, m->qE, N, m->part, El, nullptr, 0));                                 // this rank's rows
            if ((rc = tp_gather_rows_add(m, m->part, N, inp, mid)) != FL_OK) return rc;
        } else {
            M_HIP(mm(m, ly.wo, m->qEl, N, m->part, E, nullptr, 0));
            int rc = allreduce_if_tp(m, m->part, (size_t)N * E);
            if (rc != FL_OK) return rc;
            M_HIP(add_rows(m->part, E, inp, E, mid, E, N, E, st));
        }
        // feed-forward                                                                             :412-436
        const bool silu_in_gemm = N >= 9 && m->w13_il && (!exact || xh);   // silu * mul -> Q8_0 is the epilogue of the w1|w3 matmul
        const bool silu_in_gemv = fused && m->w13_il;   // decode: silu * mul is the epilogue of the w1|w3 GEMV
        // reference-order decode, unsharded: the w1|w3 workgroups own whole 32-feature blocks and write the Q8_0 operand of w2 themselves
        // (gemv1_q4_exact_stream.hip) -- w2 then is the plain N = 1 matmul on m->qF, its prologue a copy
        bool q8_from_w13 = false;
        if (silu_in_gemv && exact && !tp) {
            hipEvent_t e1;
            M_HIP(prof_begin(m, &e1));
            const hipError_t e = gemv_q4_norm_silu_q8_exact(*ly.w13, mid, ly.ffn_norm, m->silu_tab, m->qF, st);
            prof_end(m, e1);
            if (e == hipSuccess) q8_from_w13 = true;
            else if (e != hipErrorInvalidValue) M_HIP(e);
            else (void)hipGetLastError();
        }
        if (q8_from_w13) {
        } else if (silu_in_gemv) {
            M_HIP(mm_norm_silu(m, ly.w13, mid, ly.ffn_norm, m->h13));
        } else if (fused) {
            M_HIP(mm_norm(m, ly.w13, mid, ly.ffn_norm, nullptr, m->h13));
        } else {
            M_HIP(norm_q8(mid, ly.ffn_norm, nullptr));
            if (silu_in_gemm) M_HIP(mm_silu_gemm(m, ly.w13, m->qE, N, sF));
            else M_HIP(mm(m, ly.w13, m->qE, N, m->h13, 2 * Fl, nullptr, 0));
        }
        if (!fused && !silu_in_gemm) {
            M_HIP(silu_mul_quant(m->h13, 2 * Fl, N, Fl, m->silu_tab, &sF, layout, st, m->w13_il, xh && !m->tp_rows));
        }
        if (!tp) {
            if (q8_from_w13) M_HIP(mm(m, ly.w2, m->qF, 1, inp, E, mid, E));
            else if (silu_in_gemv) M_HIP(mm_quant(m, ly.w2, m->h13, inp, mid));
            else if (fused) M_HIP(mm_silu(m, ly.w2, m->h13, inp, mid));
            else M_HIP(mm(m, ly.w2, m->qF, N, inp, E, mid, E));                                   // + inpFF :441
        } else if (tpg) {
            const int rc = tp_matmul_rows(ly.w2, Fl, m->qFf, mid, inp);                             // inp = rows + mid: owed to the next layer's norm
            if (rc != FL_OK) return rc;
        } else if (m->tp_rows) {
            int rc;
            if (silu_in_gemv) {                   // decode: the f32 silu * mul features of every rank, quantized by the GEMV's prologue
                if (!m->comm) return set_error(FL_EINVAL, "tensor-parallel eval without a communicator");
                rc = fl_comm_allgather_f32(m->comm, m->h13, (size_t)Fl, reinterpret_cast<float *>(m->ag_tmp), st);
                if (rc != FL_OK) return rc;
                M_HIP(mm_quant(m, ly.w2, reinterpret_cast<const float *>(m->ag_tmp), m->part, nullptr));
            } else {
                if (fused) return set_error(FL_EINVAL, "row-split tensor parallelism needs n_ff / tp_size to be a multiple of 32");
                if ((rc = tp_gather_qact(m, m->qF, m->qFf, N, Fl, layout, xh)) != FL_OK) return rc;     // the operand, K = n_ff
                M_HIP(mm(m, ly.w2, m->qFf, N, m->part, El, nullptr, 0));
            }
            if ((rc = tp_gather_rows_add(m, m->part, N, mid, inp)) != FL_OK) return rc;
        } else {
            if (silu_in_gemv) M_HIP(mm_quant(m, ly.w2, m->h13, m->part, nullptr));
            else if (fused) M_HIP(mm_silu(m, ly.w2, m->h13, m->part, nullptr));
            else M_HIP(mm(m, ly.w2, m->qF, N, m->part, E, nullptr, 0));
            int rc = allreduce_if_tp(m, m->part, (size_t)N * E);
            if (rc != FL_OK) return rc;
            M_HIP(add_rows(m->part, E, mid, E, inp, E, N, E, st));
        }
    }
    if (pend.on && (body_only || skip_head)) {           // no norm follows in this call: the add is a launch of its own after all
        pend.on = false;
        M_HIP(gather_rows_add(reinterpret_cast<const float *>(m->ag_tmp), m->G, N, El, pend.resid, E, pend.out, E, st));
    }
    if (body_only || skip_head) return FL_OK;
    // final norm (kept in f32 for the embeddings) + lm head                                        :452-465
    float *lg = m->Vl > 0 ? m->logits_part : logits_dst ? logits_dst : m->logits;       // (the overrides: a half of a split eval writes
    float *xn = xn_dst ? xn_dst : m->xn;                                                //  its rows of the primary set's buffers)
    const int ldlg = m->Vl > 0 ? m->ldp : m->ldl;
    if (fused) {
        M_HIP(mm_norm(m, m->output, inp, m->norm_w, xn, lg));
    } else {
        M_HIP(norm_q8(inp, m->norm_w, xn));
        M_HIP(mm(m, m->output, m->qE, N, lg, ldlg, nullptr, 0));
    }
    if (m->Vl > 0) {                                          // rows V/G of the lm-head per rank -> gather the logits slices
        if (!m->comm) return set_error(FL_EINVAL, "tensor-parallel eval without a communicator");
        const int rc = fl_comm_allgather_f32(m->comm, m->logits_part, (size_t)N * m->ldp, m->gather_tmp, st);
        if (rc != FL_OK) return rc;
        M_HIP(gather_cols(m->gather_tmp, m->G, N, m->Vl, m->ldp, m->logits, m->ldl, st));
    }
    return FL_OK;
}

// the second set of work buffers, its stream, the per-layer events of two evals in flight (fl_model_ingest, split evals)
static int ensure_alt(fl_model *m) {
    if (m->alt_ready) return FL_OK;
    if (!m->alt.stream) M_HIP(hipStreamCreateWithFlags(&m->alt.stream, hipStreamNonBlocking));
    hipStream_t st1 = m->alt.stream;
    if (!m->alt.x) {                                  // (a previous call may have got this far and failed on an event below)
        const int rc = act_alloc(m, m->alt);
        if (rc != FL_OK) act_free(m->alt);            // (no half-allocated set left behind)
        m->alt.stream = st1;
        if (rc != FL_OK) return rc;
    }
    for (auto &v : m->kv_ev)
        while ((int)v.size() < m->L) {
            hipEvent_t e;
            M_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            v.push_back(e);
        }
    if (!m->ev_fork) M_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    if (!m->ev_join) M_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    m->alt_ready = true;
    return FL_OK;
}

/* One prefill eval as TWO halves in flight on the two streams (round 4).  OFF by default: built, bit-identical,
 * and measured SLOWER in both modes -- fast 11.83 -> 12.63 ms, reference-order 35.38 -> 36.03 ms per 512-token eval
 * (profiles/r04_split_eval.txt): half-size launches re-read every weight panel and double the launch count, which costs more than
 * the covered ramps and tails give back (two INDEPENDENT evals, fl_model_ingest, still gain).  The idea: every launch of an n_batch eval is a few rounds of
 * workgroups whose ramp, tail and -- for the latency-bound attention / norm kernels -- idle CUs are covered by nothing; the second
 * half needs the first only where its attention reads the K/V cache, layer by layer (the events of fl_model_ingest).  Tokens are
 * independent columns of every matmul, and a query's attention dots do not change when keys it cannot see leave its batch -- as
 * long as the first half ends on a 32-key boundary, where ggml_vec_dot_f32's 32-element steps end too (no leftover loop changes
 * hands): the split is chosen that way, so the logits are those of the single eval bit for bit in the reference-order mode
 * (tests/test_exact_gpu.py).  n1: tokens of the first half. */
static int eval_split(fl_model *m, const int32_t *tokens, int N, int n1, int n_past, bool all_logits) {
    int rc = ensure_alt(m);
    if (rc != FL_OK) return rc;
    Act &prim = *m;
    float *lg = m->logits, *xn = m->xn;                  // both halves write their rows of the PRIMARY set's logits / final-norm buffers
    M_HIP(hipEventRecord(m->ev_fork, prim.stream));
    M_HIP(hipStreamWaitEvent(m->alt.stream, m->ev_fork, 0));
    std::swap(prim, m->alt);                             // first half: the second set of buffers, the second stream
    hipError_t e = hipMemcpyAsync(m->tok_dev, tokens, (size_t)n1 * 4, hipMemcpyHostToDevice, m->stream);
    rc = e != hipSuccess ? hip_fail(e, "hipMemcpyAsync(tokens)")
                         : run_eval_kernels(m, n1, n_past, nullptr, false, 0, -1, false, !all_logits, nullptr, m->kv_ev[1].data(), lg, xn);
    std::swap(prim, m->alt);
    if (rc == FL_OK) {
        e = hipMemcpyAsync(m->tok_dev, tokens + n1, (size_t)(N - n1) * 4, hipMemcpyHostToDevice, m->stream);
        rc = e != hipSuccess ? hip_fail(e, "hipMemcpyAsync(tokens)")
                             : run_eval_kernels(m, N - n1, n_past + n1, nullptr, false, 0, -1, false, false, m->kv_ev[1].data(), m->kv_ev[0].data(),
                                                lg + (size_t)n1 * m->ldl, xn + (size_t)n1 * m->E);
    }
    e = hipEventRecord(m->ev_join, m->alt.stream);       // join: everything the second stream did happens-before what follows on the primary one
    if (e == hipSuccess) e = hipStreamWaitEvent(prim.stream, m->ev_join, 0);
    if (rc != FL_OK || e != hipSuccess) {
        (void)hipStreamSynchronize(prim.stream);
        (void)hipStreamSynchronize(m->alt.stream);
        return rc != FL_OK ? rc : hip_fail(e, "eval_split: join");
    }
    return FL_OK;
}

int fl_model_eval(fl_model *m, const int32_t *tokens, int N, int n_past, float *logits_host, int all_logits,
                  float *embeddings_host) {
    if (!m || !tokens) return set_error(FL_EINVAL, "fl_model_eval: null argument");
    if (!m->finalized) return set_error(FL_EINVAL, "fl_model_eval: model not finalized");
    if (N <= 0 || N > m->B) return set_error(FL_EINVAL, "N=%d exceeds max_batch=%d", N, m->B);
    if (n_past < 0 || n_past + N > m->n_ctx) return set_error(FL_EINVAL, "n_past+N=%d exceeds n_ctx=%d", n_past + N, m->n_ctx);
    const int E = m->E, V = m->V;
    hipStream_t st = m->stream;
    for (int i = 0; i < N; ++i)     // ggml_get_rows would read outside tok_embeddings (lib/ggml.c:8353); refuse instead
        if (tokens[i] < 0 || tokens[i] >= V) return set_error(FL_EINVAL, "token %d at position %d is outside the vocabulary (%d)", tokens[i], i, V);

    // Decode (N = 1) is launch-bound (~160 short kernels per token): the whole sequence is captured ONCE into a
    // hipGraph whose kernels read the position from device memory, and replayed per token.  Two captures: past
    // split_past positions a single workgroup per head no longer keeps up with the K/V stream (decode_attention_split).
    // Under tensor parallelism the RCCL all-reduces / all-gather are captured with the kernels (every rank replays the same
    // sequence); the single-process group of fl_comm_create_local rendezvouses on the host and cannot be captured.
    const bool tp_capturable = m->G == 1 || (m->comm && !fl_comm_is_local(m->comm) && !m->tp_graph_failed && !getenv("FL_TP_NO_GRAPH"));
    if (N == 1 && m->exact) ensure_qwd(m);          // (before any capture: it allocates)
    if (N == 1 && m->exact && m->G > 1 && m->tp_rows) ensure_fold(m);
    const bool use_graph = N == 1 && tp_capturable && m->graph_enabled && !m->profile;
    const bool split_attn = N == 1 && n_past >= m->split_past;
    if (use_graph) {
        hipGraphExec_t &exec = split_attn ? m->graph_exec_long : m->graph_exec;
        // the position and the token of this replay: npast_dev[0], npast_dev[1] (one launch, its arguments copied at enqueue)
        M_HIP(set_decode_inputs(m->npast_dev, n_past, tokens[0], st));
        if (!exec) {
            hipGraph_t g = nullptr;
            M_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = run_eval_kernels(m, 1, 0, m->npast_dev, split_attn);
            const hipError_t e = hipStreamEndCapture(st, &g);
            hipError_t ei = hipSuccess;
            if (rc == FL_OK && e == hipSuccess && hipGraphGetNodes(g, nullptr, &m->graph_nodes) != hipSuccess) { (void)hipGetLastError(); m->graph_nodes = 0; }
            if (rc == FL_OK && e == hipSuccess) ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (rc != FL_OK || e != hipSuccess || ei != hipSuccess) {
                exec = nullptr;
                if (m->G == 1) return rc != FL_OK ? rc : hip_fail(e != hipSuccess ? e : ei, "decode graph capture");
                (void)hipGetLastError();                   // collectives that cannot be captured here: plain launches
                m->tp_graph_failed = true;
            }
        }
        if (exec) {
            M_HIP(hipGraphLaunch(exec, st));
        } else {
            M_HIP(hipMemcpyAsync(m->tok_dev, m->npast_dev + 1, 4, hipMemcpyDeviceToDevice, st));      // (plain launches read the token where a prefill's are)
            const int rc = run_eval_kernels(m, 1, n_past, nullptr, split_attn);
            if (rc != FL_OK) return rc;
        }
    } else {
        // a prefill eval of >= 256 tokens: two halves in flight on the two streams, the first ending on a 32-key boundary (eval_split)
        int n1 = 0;
        if (N >= 256 && m->split_eval && m->G == 1 && m->E < 8192 && !m->profile) {
            n1 = ((n_past + N / 2 + 31) & ~31) - n_past;
            if (n1 < 64 || N - n1 < 64) n1 = 0;
        }
        if (n1 > 0) {
            const int rc = eval_split(m, tokens, N, n1, n_past, all_logits != 0);
            if (rc != FL_OK) return rc;
        } else {
            M_HIP(hipMemcpyAsync(m->tok_dev, tokens, (size_t)N * 4, hipMemcpyHostToDevice, st));
            const int rc = run_eval_kernels(m, N, n_past, nullptr, split_attn);
            if (rc != FL_OK) return rc;
        }
    }
    if (logits_host) {
        if (all_logits) M_HIP(hipMemcpy2DAsync(logits_host, (size_t)V * 4, m->logits, (size_t)m->ldl * 4, (size_t)V * 4, N, hipMemcpyDeviceToHost, st));
        else M_HIP(hipMemcpyAsync(logits_host, m->logits + (size_t)(N - 1) * m->ldl, (size_t)V * 4, hipMemcpyDeviceToHost, st));
    }
    if (embeddings_host)
        M_HIP(hipMemcpyAsync(embeddings_host, m->xn + (size_t)(N - 1) * E, (size_t)E * 4, hipMemcpyDeviceToHost, st));
    M_HIP(hipStreamSynchronize(st));
    if (m->G > 1 && m->comm) {                  // a peer exchange that gave up waiting for a peer summed / gathered stale slots: fail the eval
        const int rc = fl_comm_p2p_check(m->comm);
        if (rc != FL_OK) return rc;
    }
    if (m->profile) {
        for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m->ev[i], m->ev[i + 1]) == hipSuccess) m->prof_mm_ms += ms;
            m->prof_mm_launches++;
        }
        m->ev_used = 0;
    }
    return FL_OK;
}

/* A long prompt as consecutive chunks -- what the session's ingest loop (lib/bridge.cpp:186-238) evaluates one llama_eval at a
 * time: chunk c is tokens[sum(len[0..c-1]) ...] at position n_past + that sum.  The chunks are the same evals, bit for bit, but
 * two of them are in flight at once on two streams: chunk c+1 needs chunk c only where its attention reads the K/V cache, layer
 * by layer, so its kernels fill the launch ramps and tails of chunk c's (every launch of a 512-token eval is a single round of
 * workgroups; two independent evals overlap to 1.14x the throughput, profiles/r02_two_stream_probe.txt).  The lm-head runs for
 * the LAST chunk only (the reference computes and discards the others' logits).  logits_host: n_vocab floats of the last token,
 * or NULL.  Tensor-parallel models and profiling runs take the chunks one after the other. */
int fl_model_ingest(fl_model *m, const int32_t *tokens, const int *chunk_len, int n_chunks, int n_past, float *logits_host) {
    if (!m || !tokens || !chunk_len) return set_error(FL_EINVAL, "fl_model_ingest: null argument");
    if (!m->finalized) return set_error(FL_EINVAL, "fl_model_ingest: model not finalized");
    if (n_chunks < 1) return set_error(FL_EINVAL, "fl_model_ingest: no chunks");
    long total = 0;
    for (int c = 0; c < n_chunks; ++c) {
        if (chunk_len[c] < 1 || chunk_len[c] > m->B) return set_error(FL_EINVAL, "chunk %d: %d tokens (1..%d)", c, chunk_len[c], m->B);
        total += chunk_len[c];
    }
    if (n_past < 0 || n_past + total > m->n_ctx) return set_error(FL_EINVAL, "n_past %d + %ld tokens exceed n_ctx %d", n_past, total, m->n_ctx);
    for (long i = 0; i < total; ++i)
        if (tokens[i] < 0 || tokens[i] >= m->V) return set_error(FL_EINVAL, "token %d at position %ld is outside the vocabulary (%d)", tokens[i], i, m->V);
    if (n_chunks == 1 || m->profile) {
        long off = 0;
        for (int c = 0; c < n_chunks; ++c) {
            const int rc = fl_model_eval(m, tokens + off, chunk_len[c], n_past + (int)off, c == n_chunks - 1 ? logits_host : nullptr, 0, nullptr);
            if (rc != FL_OK) return rc;
            off += chunk_len[c];
        }
        return FL_OK;
    }
    // Two chunks in flight pay where a launch is one short round of workgroups (7B: +10 %, 13B: +15 %); at 65B width the launches are
    // four times longer, their ramps and tails are 3 % of them, and two evals sharing the L2 cost more than that (4.59 k vs 4.66 k
    // tok/s, profiles/r02_bench_configs.jsonl).  Tensor-parallel models have one communicator: one stream.
    const bool two = m->G == 1 && m->E < 8192 && !m->ingest_one_stream;
    if (!two) {
        long off = 0;
        for (int c = 0; c < n_chunks; ++c) {
            M_HIP(hipMemcpyAsync(m->tok_dev, tokens + off, (size_t)chunk_len[c] * 4, hipMemcpyHostToDevice, m->stream));
            const int rc = run_eval_kernels(m, chunk_len[c], n_past + (int)off, nullptr, false, 0, -1, false, c != n_chunks - 1);
            if (rc != FL_OK) return rc;
            off += chunk_len[c];
        }
        if (logits_host)
            M_HIP(hipMemcpyAsync(logits_host, m->logits + (size_t)(chunk_len[n_chunks - 1] - 1) * m->ldl, (size_t)m->V * 4, hipMemcpyDeviceToHost, m->stream));
        M_HIP(hipStreamSynchronize(m->stream));
        if (m->G > 1 && m->comm) return fl_comm_p2p_check(m->comm);
        return FL_OK;
    }
    { const int rca = ensure_alt(m); if (rca != FL_OK) return rca; }
    Act &prim = *m;
    // the second stream starts behind whatever the primary stream still has queued (an earlier eval's K/V stores)
    M_HIP(hipEventRecord(m->ev_fork, prim.stream));
    M_HIP(hipStreamWaitEvent(m->alt.stream, m->ev_fork, 0));
    long off = 0;
    int rc = FL_OK;
    for (int c = 0; c < n_chunks && rc == FL_OK; ++c) {
        const int set = (n_chunks - 1 - c) & 1;           // the last chunk runs on the primary set: its logits / embeddings stay there
        if (set) std::swap(prim, m->alt);
        hipError_t e = hipMemcpyAsync(m->tok_dev, tokens + off, (size_t)chunk_len[c] * 4, hipMemcpyHostToDevice, m->stream);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync(tokens)");
        if (rc == FL_OK)
            rc = run_eval_kernels(m, chunk_len[c], n_past + (int)off, nullptr, false, 0, -1, false, c != n_chunks - 1,
                                  c > 0 ? m->kv_ev[set ^ 1].data() : nullptr, m->kv_ev[set].data());
        if (set) std::swap(prim, m->alt);
        off += chunk_len[c];
    }
    // join: everything the second stream did happens-before whatever follows on the primary one
    hipError_t e = hipEventRecord(m->ev_join, m->alt.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(prim.stream, m->ev_join, 0);
    if (e == hipSuccess && rc == FL_OK && logits_host)
        e = hipMemcpyAsync(logits_host, m->logits + (size_t)(chunk_len[n_chunks - 1] - 1) * m->ldl, (size_t)m->V * 4, hipMemcpyDeviceToHost, prim.stream);
    const hipError_t es = hipStreamSynchronize(prim.stream);
    (void)hipStreamSynchronize(m->alt.stream);
    if (rc != FL_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "fl_model_ingest: join");
    if (es != hipSuccess) return hip_fail(es, "fl_model_ingest: synchronize");
    return FL_OK;
}

/* bit 0 (default 1): decode evals replay a captured hipGraph, else plain launches; bit 1 (default 0): decode uses the
 * generic per-op kernels instead of the fused single-token ones; bit 2: prefill attention as three kernels; bit 3: decode
 * attention always in one launch per layer; bit 4: always the two-launch split form (default: split from position 256 on);
 * bit 5: prefill attention always in its key-tiled (deep-context) form; bit 7: fl_model_ingest on one stream; bit 8: a prefill
 * eval of >= 256 tokens runs as two halves on the two streams (same bits; measured slower: profiles/r04_split_eval.txt).
 * Debugging / A-B timing; results do not depend on bits 0, 2, 3, 4, 5, 7, 8. */
int fl_model_set_graph(fl_model *m, int mode) {
    if (!m) return set_error(FL_EINVAL, "null model");
    m->graph_enabled = (mode & 1) != 0;
    const bool fuse = (mode & 2) == 0;
    m->fuse_prefill_attn = (mode & 4) == 0;
    m->force_deep_attn = (mode & 32) ? 1 : 0;
    m->ingest_one_stream = (mode & 128) != 0;
    m->split_eval = (mode & 256) != 0;
    m->split_past = (mode & 8) ? INT_MAX : (mode & 16) ? 0 : 256;
    const bool kvp = (mode & 512) == 0;
    if (fuse != m->fuse_decode || kvp != m->kv_prefetch) {
        if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
        if (m->graph_exec_long) (void)hipGraphExecDestroy(m->graph_exec_long);
        m->graph_exec = m->graph_exec_long = nullptr;
    }
    m->fuse_decode = fuse;
    m->kv_prefetch = kvp;
    return FL_OK;
}

/* 1: every matmul of the following evals runs in the reference's summation order (exact_kernels.hip) -- logits bit-identical to
 * the reference's x86 build; 0: the fast kernels (exact block dots, own f32 order).  The decode graphs are re-captured. */
int fl_model_set_exact(fl_model *m, int on) {
    if (!m) return set_error(FL_EINVAL, "null model");
    if (!on && m->qs_dropped) { const int rc = restore_qs(m); if (rc != FL_OK) return rc; }      // (the fast kernels read the QW16 nibble planes)
    if ((on != 0) != m->exact) {
        if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
        if (m->graph_exec_long) (void)hipGraphExecDestroy(m->graph_exec_long);
        m->graph_exec = m->graph_exec_long = nullptr;
    }
    m->exact = on != 0;
    return FL_OK;
}
int fl_model_get_exact(const fl_model *m) { return m && m->exact ? 1 : 0; }
/* build the derived weight copies of the reference-order kernels now instead of inside the first eval that needs them (fastllama_hip.h) */
int fl_model_prepare(fl_model *m, int flags) {
    if (!m) return set_error(FL_EINVAL, "null model");
    if (!m->finalized) return set_error(FL_EINVAL, "fl_model_prepare: model not finalized");
    if (flags & 4) {                       // lean: exactly the named copies, from now on
        m->lean = true;
        m->lean_h16 = (flags & 1) != 0;
        m->lean_qwd = (flags & 2) != 0;
        if (m->h16_state < 0 && m->lean_h16) m->h16_state = 0;      // (a copy an earlier lean call left out may be named now)
        if (m->qwd_state < 0 && m->lean_qwd) m->qwd_state = 0;
    }
    if ((flags & 3) && m->qs_dropped) { const int rc = restore_qs(m); if (rc != FL_OK) return rc; }      // (the copies are built from the nibble planes)
    if (flags & 1) ensure_h16(m);
    if (flags & 2) ensure_qwd(m);
    M_HIP(hipStreamSynchronize(m->stream));
    lean_drop(m);
    return FL_OK;
}
/* resident bytes by kind: [0] QW16 nibble planes, [1] scale planes (d, m), [2] WH16 copies, [3] QWD copies, [4] everything else (K / V cache, work
 * buffers, tables, token embeddings' share included in 0 / 1) */
int fl_model_memory(const fl_model *m_, size_t *bytes5) {
    fl_model *m = const_cast<fl_model *>(m_);
    if (!m || !bytes5) return set_error(FL_EINVAL, "null argument");
    size_t b[5] = {0, 0, 0, 0, 0};
    std::vector<fl_qtensor *> ts = matmul_tensors(m);
    if (m->tok_emb) ts.push_back(m->tok_emb);
    for (fl_qtensor *t : ts) {
        if (!t) continue;
        const size_t nblk = (size_t)t->M16 * t->KB;
        if (t->qs) b[0] += nblk * 16;
        b[1] += nblk * (t->m ? 8 : 4);
        if (t->h16) b[2] += wh16_bytes(*t);
        if (t->qwd) b[3] += qwd_bytes(*t);
    }
    const size_t known = b[0] + b[1] + b[2] + b[3];
    b[4] = m->dev_bytes > known ? m->dev_bytes - known : 0;
    for (int i = 0; i < 5; ++i) bytes5[i] = b[i];
    return FL_OK;
}
/* launches of one decode token: the nodes of the decode hipGraph captured last (0: none captured yet) */
int fl_model_graph_nodes(const fl_model *m) { return m ? (int)m->graph_nodes : 0; }
/* 1: a row-split tensor-parallel model whose decode exchanges run as the tails of the producing launches (tp_tail.h); decided at the first single-token eval */
int fl_model_tp_folded(const fl_model *m) { return m && m->fold_state > 0 ? 1 : 0; }
int fl_model_prepared(const fl_model *m) { return m ? (m->h16_state > 0 ? 1 : 0) | (m->qwd_state > 0 ? 2 : 0) : 0; }
/* the mode new models start in: FL_FAST=1 (or FL_EXACT=0) in the environment selects the fast kernels, FL_EXACT=1 the reference
 * order; else the built-in default (reference order) */
int fl_default_exact(void) {
    const char *e = getenv("FL_EXACT");
    if (e && *e) return atoi(e) != 0;
    const char *f = getenv("FL_FAST");
    if (f && *f) return atoi(f) == 0;
    return FL_DEFAULT_EXACT;
}

/* bench hook: time every quantized-matmul launch of the following evals with HIP events on the eval stream */
int fl_model_profile(fl_model *m, int enable, double *mm_ms_total, long *mm_launches) {
    if (!m) return set_error(FL_EINVAL, "null model");
    if (mm_ms_total) *mm_ms_total = m->prof_mm_ms;
    if (mm_launches) *mm_launches = m->prof_mm_launches;
    m->profile = enable != 0;
    if (enable == 1) { m->prof_mm_ms = 0.0; m->prof_mm_launches = 0; }
    return FL_OK;
}

/* test hook: layers [l0, l1) of Model::eval on a caller-provided layer input x_host [N][n_embd] (teacher forcing) at n_past;
 * x_out_host receives the output of layer l1-1.  Afterwards the Q8_0 operands the last layer fed to its wo / w1|w3 / w2
 * matmuls can be exported (which = 0 / 1 / 2) as block_q8_0 rows -- the discrete intermediates a rounding flip shows up in. */
int fl_model_debug_layers(fl_model *m, int l0, int l1, const float *x_host, int N, int n_past, float *x_out_host) {
    if (!m || !m->finalized || !x_host || !x_out_host) return set_error(FL_EINVAL, "fl_model_debug_layers: bad argument");
    if (m->G != 1 || N < 9 || N > m->B || l0 < 0 || l1 > m->L || l0 >= l1 || n_past < 0 || n_past + N > m->n_ctx)
        return set_error(FL_EINVAL, "fl_model_debug_layers: single-GPU prefill shapes only");
    M_HIP(hipMemcpyAsync(m->x, x_host, (size_t)N * m->E * 4, hipMemcpyHostToDevice, m->stream));
    const int rc = run_eval_kernels(m, N, n_past, nullptr, false, l0, l1, true);
    if (rc != FL_OK) return rc;
    M_HIP(hipMemcpyAsync(x_out_host, m->x, (size_t)N * m->E * 4, hipMemcpyDeviceToHost, m->stream));
    M_HIP(hipStreamSynchronize(m->stream));
    return FL_OK;
}
int fl_model_debug_export_q8(fl_model *m, int which, int N, void *blocks_host /* [N][K/32] block_q8_0 */) {
    if (!m || !m->finalized || !blocks_host || which < 0 || which > 2 || N < 9) return set_error(FL_EINVAL, "bad argument");
    const fl_qact &a = which == 0 ? m->qEl : which == 1 ? m->qE : m->qF;
    const int K = a.KB * FL_QK;
    void *tmp = nullptr;
    M_HIP(hipMalloc(&tmp, (size_t)N * a.KB * 40));
    hipError_t e = export_qa16_to_aos(a, N, K, tmp, m->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(blocks_host, tmp, (size_t)N * a.KB * 40, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(tmp);
    return e == hipSuccess ? FL_OK : hip_fail(e, "fl_model_debug_export_q8");
}

const float *fl_model_logits_dev(const fl_model *m) { return m ? m->logits : nullptr; }
int fl_model_logits_ld(const fl_model *m) { return m ? m->ldl : 0; }

/* rows [row0, row0 + rows) of the logits the last eval left in HBM -> host ([rows][n_vocab], dense) */
int fl_model_logits_read(fl_model *m, int row0, int rows, float *logits_host) {
    if (!m || !m->finalized || !logits_host || row0 < 0 || rows < 1 || row0 + rows > m->B) return set_error(FL_EINVAL, "fl_model_logits_read: bad argument");
    M_HIP(hipMemcpy2DAsync(logits_host, (size_t)m->V * 4, m->logits + (size_t)row0 * m->ldl, (size_t)m->ldl * 4, (size_t)m->V * 4, rows,
                           hipMemcpyDeviceToHost, m->stream));
    M_HIP(hipStreamSynchronize(m->stream));
    return FL_OK;
}

/* nll_host[i] = -log softmax(logits[row0 + i])[next_tokens_host[i]] of the last eval's logits, computed on the device:
 * the per-row loop of FastLlama::perplexity (lib/bridge.cpp:397-407) without shipping rows x n_vocab floats to the host */
int fl_model_logits_nll(fl_model *m, int row0, int rows, const int32_t *next_tokens_host, double *nll_host) {
    if (!m || !m->finalized || !next_tokens_host || !nll_host || row0 < 0 || rows < 1 || row0 + rows > m->B)
        return set_error(FL_EINVAL, "fl_model_logits_nll: bad argument");
    for (int i = 0; i < rows; ++i)
        if (next_tokens_host[i] < 0 || next_tokens_host[i] >= m->V) return set_error(FL_EINVAL, "fl_model_logits_nll: token outside the vocabulary");
    double *out = nullptr;
    M_HIP(hipMalloc((void **)&out, (size_t)rows * 8));
    hipError_t e = hipMemcpyAsync(m->tok_dev, next_tokens_host, (size_t)rows * 4, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = logits_nll(m->logits, m->ldl, m->V, m->tok_dev, row0, rows, out, m->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(nll_host, out, (size_t)rows * 8, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(out);
    return e == hipSuccess ? FL_OK : hip_fail(e, "fl_model_logits_nll");
}
void *fl_model_stream(const fl_model *m) { return m ? (void *)m->stream : nullptr; }
size_t fl_model_device_bytes(const fl_model *m) { return m ? m->dev_bytes : 0; }

/* KV cache <-> host in the REFERENCE's layout (K: [layer][n_ctx][n_embd], V: [layer][n_embd][n_ctx], f32;
 * KVCacheBuffer::save_state dumps both raw, lib/llama.cpp:57-78).  Only for tp_size == 1. */
int fl_model_kv_read(const fl_model *m, float *k_host, float *v_host) {
    if (!m || !m->finalized || m->G != 1) return set_error(FL_EINVAL, "kv_read: need a finalized single-GPU model");
    const size_t n = (size_t)m->L * m->n_ctx * m->E * 4;
    M_HIP(hipStreamSynchronize(m->stream));                      // (the eval stream is non-blocking: the null stream does not wait for it)
    M_HIP(hipMemcpy(k_host, m->kc, n, hipMemcpyDeviceToHost));
    M_HIP(hipMemcpy(v_host, m->vc, n, hipMemcpyDeviceToHost));
    return FL_OK;
}
int fl_model_kv_write(fl_model *m, const float *k_host, const float *v_host) {
    if (!m || !m->finalized || m->G != 1) return set_error(FL_EINVAL, "kv_write: need a finalized single-GPU model");
    const size_t n = (size_t)m->L * m->n_ctx * m->E * 4;
    M_HIP(hipStreamSynchronize(m->stream));
    M_HIP(hipMemcpy(m->kc, k_host, n, hipMemcpyHostToDevice));
    M_HIP(hipMemcpy(m->vc, v_host, n, hipMemcpyHostToDevice));
    M_HIP(hipDeviceSynchronize());                               // the copies have landed before the next eval's stream reads them
    return FL_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// LoRA on resident weights (reference lib/llama.cpp:697-944)
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct TensorRef {
    fl_qtensor *t = nullptr;
    int row0 = 0, rows = 0;        // rows of the (possibly fused) device tensor that hold this base tensor's shard
    int il_part = -1;              // >= 0: rows are woven by 16 (w1|w3): local row r sits at 32 (r / 16) + 16 il_part + r % 16
    int grow0 = 0, gcol0 = 0;      // where the shard sits in the FULL base tensor (tensor parallel): row / column offset
    int Kfull = 0, Mfull = 0;      // full base tensor: ne0, ne1
};
}  // namespace

static int locate_tensor(fl_model *m, const char *name, TensorRef *o) {
    const int E = m->E, F = m->F, V = m->V, G = m->G, r = m->rank;
    std::string nm(name);
    if (nm == "tok_embeddings.weight" || nm == "output.weight") {
        o->t = nm[0] == 't' ? m->tok_emb : m->output;
        o->rows = V; o->Kfull = E; o->Mfull = V;
        if (nm[0] == 'o' && m->Vl > 0) { o->rows = m->Vl; o->grow0 = r * m->Vl; }
        return FL_OK;
    }
    int il = -1, off = 0;
    if (sscanf(name, "layers.%d.%n", &il, &off) != 1 || il < 0 || il >= m->L || off == 0)
        return set_error(FL_EINVAL, "unknown tensor '%s'", name);
    Layer &ly = m->layers[il];
    const std::string sub(name + off);
    if (sub == "attention.wq.weight" || sub == "attention.wk.weight" || sub == "attention.wv.weight") {
        const int part = sub[11] == 'q' ? 0 : sub[11] == 'k' ? 1 : 2;
        o->t = ly.wqkv; o->row0 = part * m->El; o->rows = m->El; o->grow0 = r * m->El; o->Kfull = E; o->Mfull = E;
        return FL_OK;
    }
    if (sub == "feed_forward.w1.weight" || sub == "feed_forward.w3.weight") {
        const int part = sub[14] == '1' ? 0 : 1;
        o->t = ly.w13; o->row0 = part * m->Fl; o->rows = m->Fl; o->grow0 = r * m->Fl; o->Kfull = E; o->Mfull = F;
        if (m->w13_il) { o->il_part = part; o->row0 = 0; }
        return FL_OK;
    }
    if (sub == "attention.wo.weight") {
        o->t = ly.wo; o->rows = E; o->gcol0 = r * (E / G); o->Kfull = E; o->Mfull = E;
        if (m->tp_rows) { o->rows = m->El; o->grow0 = r * m->El; o->gcol0 = 0; }
        return FL_OK;
    }
    if (sub == "feed_forward.w2.weight") {
        o->t = ly.w2; o->rows = E; o->gcol0 = r * (F / G); o->Kfull = F; o->Mfull = E;
        if (m->tp_rows) { o->rows = m->El; o->grow0 = r * m->El; o->gcol0 = 0; }
        return FL_OK;
    }
    return set_error(FL_EINVAL, "tensor '%s' is not a quantized matrix of the model", name);
}

static int dev_copy_f32(const float *src, size_t n, float **out) {
    *out = nullptr;
    if (!src) return FL_OK;
    M_HIP(hipMalloc((void **)out, n * 4));
    M_HIP(hipMemcpy(*out, src, n * 4, hipMemcpyDefault));       // host or device source
    M_HIP(hipDeviceSynchronize());                               // (a device source copies on the null stream; the merge runs on the eval stream)
    return FL_OK;
}

extern "C" int fl_model_lora_shape(fl_model *m, const char *base_name, int *ne0, int *ne1) {
    if (!m || !base_name) return set_error(FL_EINVAL, "null argument");
    TensorRef tr;
    int rc = locate_tensor(m, base_name, &tr);
    if (rc != FL_OK) return rc;
    if (ne0) *ne0 = tr.Kfull;
    if (ne1) *ne1 = tr.Mfull;
    return FL_OK;
}

/* W_base <- quantize_row_q(dequantize_row_q(W_base) + sign * BA).  ba: [ne1][ne0] f32 (cached adapter), or a: [ne0][r],
 * b: [ne1][r] (BA[m][k] = ggml_vec_dot_f32(r, a_k, b_m)); FULL tensors even under tensor parallelism (the shard is
 * taken here); host or device pointers.  keep_backup != 0: the first touch of a tensor saves its current contents for
 * fl_model_lora_restore (the reference's use_mmap behaviour). */
extern "C" int fl_model_lora_apply(fl_model *m, const char *base_name, const float *ba, const float *a, const float *b, int r,
                                   float sign, int keep_backup) {
    if (!m || !base_name) return set_error(FL_EINVAL, "null argument");
    if (!m->finalized) return set_error(FL_EINVAL, "model not finalized");
    if (!ba && (!a || !b || r < 1)) return set_error(FL_EINVAL, "lora: need either BA or A, B and r >= 1");
    TensorRef tr;
    int rc = locate_tensor(m, base_name, &tr);
    if (rc != FL_OK) return rc;
    fl_qtensor *t = tr.t;
    if ((rc = restore_qs(m)) != FL_OK) return rc;          // (lean mode: the merge works on the QW16 planes; they stay until the next fl_model_prepare(.., 4))
    M_HIP(hipStreamSynchronize(m->stream));
    const size_t nblk = (size_t)t->M16 * t->KB;
    if (keep_backup) {
        bool have = false;
        for (auto &bk : m->lora_backups) have = have || bk.t == t;
        if (!have) {
            fl_model::LoraBackup bk{t, nullptr, nullptr, nullptr};
            M_HIP(hipMalloc(&bk.qs, nblk * 16));
            M_HIP(hipMalloc(&bk.d, nblk * 4));
            M_HIP(hipMemcpyAsync(bk.qs, t->qs, nblk * 16, hipMemcpyDeviceToDevice, m->stream));   // (on the stream the merge kernels run on)
            M_HIP(hipMemcpyAsync(bk.d, t->d, nblk * 4, hipMemcpyDeviceToDevice, m->stream));
            if (t->m) {
                M_HIP(hipMalloc(&bk.mm, nblk * 4));
                M_HIP(hipMemcpyAsync(bk.mm, t->m, nblk * 4, hipMemcpyDeviceToDevice, m->stream));
            }
            m->lora_backups.push_back(bk);
        }
    }
    const int bs = t->type == FL_TYPE_Q4_0 ? 20 : 24;
    void *aos = nullptr;
    float *dba = nullptr, *da = nullptr, *db = nullptr;
    int *flag = nullptr;
    int bad = 0;
    auto cleanup = [&] { for (void *p : {aos, (void *)dba, (void *)da, (void *)db, (void *)flag}) if (p) (void)hipFree(p); };
    hipError_t e = hipMalloc(&aos, (size_t)t->M * t->KB * bs);
    if (e == hipSuccess) e = hipMalloc((void **)&flag, 4);
    if (e != hipSuccess) { cleanup(); return hip_fail(e, "hipMalloc(lora staging)"); }
    if (ba) rc = dev_copy_f32(ba, (size_t)tr.Mfull * tr.Kfull, &dba);
    else {
        rc = dev_copy_f32(a, (size_t)tr.Kfull * r, &da);
        if (rc == FL_OK) rc = dev_copy_f32(b, (size_t)tr.Mfull * r, &db);
    }
    if (rc != FL_OK) { cleanup(); return rc; }
    e = unpack_from_qw16(t->type, t->qs, t->d, t->m, t->M, t->K, aos, m->stream);
    if (e == hipSuccess)
        e = lora_add_aos(t->type, aos, t->KB, tr.row0, tr.rows, tr.il_part, dba, tr.Kfull, da, db, r, tr.grow0, tr.gcol0, sign, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, 4, m->stream);
    if (e == hipSuccess) e = repack_to_qw16(t->type, aos, t->M, t->K, t->qs, t->d, t->m, flag, m->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    cleanup();
    if (e != hipSuccess) return hip_fail(e, "fl_model_lora_apply");
    if (bad) return set_error(FL_EINVAL, "lora: a merged Q4_0 block scale fell below 2^-122");
    if (t->h16 && (rc = fl_qtensor_build_h16(t, m->stream)) != FL_OK) return rc;
    if (t->qwd) return fl_qtensor_build_qwd(t, m->stream);
    return FL_OK;
}

/* put back every tensor saved by fl_model_lora_apply(keep_backup = 1) and drop the copies */
extern "C" int fl_model_lora_restore(fl_model *m) {
    if (!m) return set_error(FL_EINVAL, "null model");
    if (const int rc = restore_qs(m)) return rc;
    M_HIP(hipStreamSynchronize(m->stream));
    // (device-to-device copies on the EVAL stream: a plain hipMemcpy runs on the null stream, which a non-blocking stream does not wait for)
    for (auto &bk : m->lora_backups) {
        const size_t nblk = (size_t)bk.t->M16 * bk.t->KB;
        M_HIP(hipMemcpyAsync(bk.t->qs, bk.qs, nblk * 16, hipMemcpyDeviceToDevice, m->stream));
        M_HIP(hipMemcpyAsync(bk.t->d, bk.d, nblk * 4, hipMemcpyDeviceToDevice, m->stream));
        if (bk.mm) M_HIP(hipMemcpyAsync(bk.t->m, bk.mm, nblk * 4, hipMemcpyDeviceToDevice, m->stream));
        if (bk.t->h16) {
            const int rc = fl_qtensor_build_h16(bk.t, m->stream);
            if (rc != FL_OK) return rc;
        }
        if (bk.t->qwd) {
            const int rc = fl_qtensor_build_qwd(bk.t, m->stream);
            if (rc != FL_OK) return rc;
        }
    }
    M_HIP(hipStreamSynchronize(m->stream));
    for (auto &bk : m->lora_backups) {
        (void)hipFree(bk.qs); (void)hipFree(bk.d);
        if (bk.mm) (void)hipFree(bk.mm);
    }
    m->lora_backups.clear();
    return FL_OK;
}

/* this rank's rows of a base tensor as reference AoS blocks (tests, tooling): rows x (K_local/32) blocks */
extern "C" int fl_model_tensor_download(fl_model *m, const char *base_name, void *aos_host) {
    if (!m || !base_name || !aos_host) return set_error(FL_EINVAL, "null argument");
    TensorRef tr;
    int rc = locate_tensor(m, base_name, &tr);
    if (rc != FL_OK) return rc;
    const fl_qtensor *t = tr.t;
    if ((rc = restore_qs(m)) != FL_OK) return rc;
    const int bs = t->type == FL_TYPE_Q4_0 ? 20 : 24;
    void *aos = nullptr;
    M_HIP(hipMalloc(&aos, (size_t)t->M * t->KB * bs));
    hipError_t e = unpack_from_qw16(t->type, t->qs, t->d, t->m, t->M, t->K, aos, m->stream);
    const size_t rowb = (size_t)t->KB * bs;
    if (e == hipSuccess && tr.il_part >= 0)
        e = hipMemcpy2DAsync(aos_host, 16 * rowb, (char *)aos + (size_t)tr.il_part * 16 * rowb, 32 * rowb, 16 * rowb, (size_t)tr.rows / 16,
                             hipMemcpyDeviceToHost, m->stream);
    else if (e == hipSuccess)
        e = hipMemcpyAsync(aos_host, (char *)aos + (size_t)tr.row0 * rowb, (size_t)tr.rows * rowb, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(aos);
    return e == hipSuccess ? FL_OK : hip_fail(e, "fl_model_tensor_download");
}

void fl_model_free(fl_model *m) {
    if (!m) return;
    (void)hipDeviceSynchronize();
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fl_qtensor_free(m->tok_emb);
    fl_qtensor_free(m->output);
    fr(m->norm_w);
    for (auto &ly : m->layers) {
        fr(ly.attn_norm); fr(ly.ffn_norm);
        fl_qtensor_free(ly.wqkv); fl_qtensor_free(ly.wo); fl_qtensor_free(ly.w13); fl_qtensor_free(ly.w2);
        fr(ly.s_qkv.aos); fr(ly.s_13.aos);
    }
    fr(m->kc); fr(m->vc); fr(m->exp_tab); fr(m->silu_tab); fr(m->rope_tab);
    fr(m->logits_part); fr(m->gather_tmp);
    fr(m->qFf.q); fr(m->qFf.d); fr(m->qFf.s); fr(m->qFf.h16); fr(m->ag_send); fr(m->ag_tmp);      // (row-split tensor parallelism)
    fr(m->fold_dev);
    {
        hipStream_t st0 = m->stream, st1 = m->alt.stream;
        act_free(*m);
        act_free(m->alt);
        m->stream = st0;
        if (st1) (void)hipStreamDestroy(st1);
    }
    for (auto &v : m->kv_ev) for (hipEvent_t e : v) (void)hipEventDestroy(e);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    for (auto &bk : m->lora_backups) { fr(bk.qs); fr(bk.d); fr(bk.mm); }
    for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
    if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
    if (m->graph_exec_long) (void)hipGraphExecDestroy(m->graph_exec_long);
    fr(m->npast_dev);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

}  // extern "C"

