// test_hooks.cpp -> libfastllama_hip_hooks.so: the fl_debug_* entry points (include/fastllama_hip_test.h) -- single kernels and fused forms on
// caller-provided device buffers, kernel-family selection, table builders -- for the per-op parity tests, the tuning sweeps under
// scripts/ and the A/B legs of bench.py.  Linked against libfastllama_hip.so (same process, same globals); the product library exports
// none of them.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "../../include/fastllama_hip_test.h"
#include "internal.h"

using namespace fl;

// everything this library calls inside libfastllama_hip.so goes through the table of internal.h (the product library exports its C API only)
static const fl::InternalTable *const IT = fl_internal_table();
#define attn_pv_exact (IT->attn_pv_exact)
#define attn_scores_exact (IT->attn_scores_exact)
#define attn_scores_softmax_exact (IT->attn_scores_softmax_exact)
#define build_f16_tables (IT->build_f16_tables)
#define build_rope_table (IT->build_rope_table)
#define check_mm (IT->check_mm)
#define decode_attention (IT->decode_attention)
#define decode_attention_split (IT->decode_attention_split)
#define dot_f32_abt_exact (IT->dot_f32_abt_exact)
#define ensure_h16 (IT->ensure_h16)
#define gemm32_mixed_split (IT->gemm32_mixed_split)
#define gemm_f32_abt (IT->gemm_f32_abt)
#define gemm_q4_exact_h16 (IT->gemm_q4_exact_h16)
#define gemm_q4_exact_h16_qkv (IT->gemm_q4_exact_h16_qkv)
#define gemm_q4_exact_h16_silu (IT->gemm_q4_exact_h16_silu)
#define gemm_q4_mfma (IT->gemm_q4_mfma)
#define gemm_q4_mfma_qkv (IT->gemm_q4_mfma_qkv)
#define gemm_q4_mfma_silu (IT->gemm_q4_mfma_silu)
#define gemv1_llc_pair_ws_bytes (IT->gemv1_llc_pair_ws_bytes)
#define gemv_q4_norm (IT->gemv_q4_norm)
#define gemv_q4_norm_exact (IT->gemv_q4_norm_exact)
#define gemv_q4_norm_silu (IT->gemv_q4_norm_silu)
#define gemv_q4_norm_silu_exact (IT->gemv_q4_norm_silu_exact)
#define gemv_q4 (IT->gemv_q4)
#define gemv_q4_exact (IT->gemv_q4_exact)
#define gemv_q4_norm_silu_q8_exact (IT->gemv_q4_norm_silu_q8_exact)
#define gemv_q4_quant (IT->gemv_q4_quant)
#define gemv_q4_quant_exact (IT->gemv_q4_quant_exact)
#define gemv_q4_silu (IT->gemv_q4_silu)
#define gemv_q4_silu_exact (IT->gemv_q4_silu_exact)
#define hip_fail (IT->hip_fail)
#define mul_mat_q_which (IT->mul_mat_q_which)
#define op_exact (IT->op_exact)
#define prefill_attention (IT->prefill_attention)
#define rmsnorm_quant (IT->rmsnorm_quant)
#define rope_kv (IT->rope_kv)
#define set_error (IT->set_error)
#define silu_mul_quant (IT->silu_mul_quant)
#define softmax_rows (IT->softmax_rows)
#define g_gemm_force_cfg (*IT->g_gemm_force_cfg)
#define g_gemv_force_waves (*IT->g_gemv_force_waves)
#define g_stream_min_groups (*IT->g_stream_min_groups)
#define g_stream_force_nw (*IT->g_stream_force_nw)
#define g_pv_waves (*IT->g_pv_waves)
#define g_stream_helpers (*IT->g_stream_helpers)
#define g_op_mode (*IT->g_op_mode)

#define FL_HIP(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)
#define M_HIP(call) FL_HIP(call)

static inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

/* ------------------------------------------------------------------------------------------------
 * switches
 * ---------------------------------------------------------------------------------------------- */
static int g_debug_pair1 = 0;      // fl_debug_set(5, form): the form of the reference-order w1|w3 kernel fl_debug_gemv_norm_silu runs (q4_kernels.h; 0 automatic)
static int g_debug_exact = 0;      // fl_debug_set(2, 1): the single-token hooks below run the reference-order kernels

/* host logic of the mixed-tile GEMM launch (gemm_q4_mfma32.hip, cfg 116): how M16/16 row groups x ceil(N/16) column groups are
 * split into workgroups of 128 x 64 tiles (row groups [0, mg_split)) and of 128 x 32 tiles (the rest) */
int fl_debug_gemm_mixed_split(int row_groups, int col_groups, int *n_a, int *mg_split, int *n_b) {
    if (row_groups < 1 || col_groups < 1 || !n_a || !mg_split || !n_b) return set_error(FL_EINVAL, "fl_debug_gemm_mixed_split: bad arguments");
    gemm32_mixed_split(row_groups, col_groups, n_a, mg_split, n_b);
    return FL_OK;
}
int fl_debug_set(int what, int value) {
    if (what == 0) g_gemm_force_cfg = value;
    if (what == 1) g_gemv_force_waves = value;   // 0 = automatic, else 4 / 8 / 16 waves per 16-row group
    if (what == 2) g_debug_exact = value;            // the single-token test hooks (fl_debug_gemv_*, fl_debug_decode_attention*) in exact mode
    if (what == 9) g_stream_helpers = value != 0;    // prologue-only waves in front of the streaming ones (wq|wk|wv-sized launches of the one-wave-per-row-group form)
    if (what == 8) g_pv_waves = value == 4 ? 4 : 8;  // waves per workgroup of the reference-order V.P kernel behind a deep context
    if (what == 7) g_stream_force_nw = value;        // ... and its row groups per workgroup (0 automatic; rounded up to whole 32-feature blocks for the woven forms)
    if (what == 6) g_stream_min_groups = value;      // reference-order N = 1 matmuls: row groups from which the one-wave-per-row-group form runs (-1 automatic, 1 always, 1 << 30 never)
    if (what == 5) g_debug_pair1 = value;            // fl_debug_gemv_norm_silu in exact mode: 1 / 2 pins a form of the w1|w3 kernel, 0 automatic
    if (what == 4) g_op_mode = value;            // (= fl_set_op_mode: kept for the sweep scripts)
    return FL_OK;
}

int fl_debug_qact_layout(const fl_qact *a) { return a ? static_cast<const fl_qact_impl *>(a)->layout : 0; }

/* ------------------------------------------------------------------------------------------------
 * the quantized matmul, one kernel family at a time, and the fused forms of the prefill GEMM (model.cpp run_eval_kernels)
 * ---------------------------------------------------------------------------------------------- */
int fl_debug_mul_mat_q(const fl_qtensor *W, const fl_qact *a, float *y, int ldy, int which, void *st) {
    return mul_mat_q_which(W, a, y, ldy, which, st);      /* which: internal.h */
}

/* test hooks: the fused forms of the prefill GEMM (model.cpp run_eval_kernels) on caller-provided buffers */
int fl_debug_mul_mat_q_resid(const fl_qtensor *W, const fl_qact *a_, float *y, int ldy, const float *resid, int ldr, void *st) {
    const fl_qact_impl *a = static_cast<const fl_qact_impl *>(a_);
    int rc = check_mm(W, a, y, ldy);
    if (rc != FL_OK) return rc;
    if (a->layout != 16) return set_error(FL_EINVAL, "gemm needs the QA16 layout");
    if (op_exact()) {
        fl_qact_impl *am = const_cast<fl_qact_impl *>(a);
        if ((rc = ensure_h16(W, am, st)) != FL_OK) return rc;
        FL_HIP(gemm_q4_exact_h16(*W, *a, a->N, y, ldy, S(st), resid, ldr));
        return FL_OK;
    }
    FL_HIP(gemm_q4_mfma(*W, *a, a->N, y, ldy, S(st), resid, ldr));
    return FL_OK;
}
int fl_debug_gemm_qkv(const fl_qtensor *W, const fl_qact *a_, float *y, int ldy, const float *rope_tab_dev, float *kc, float *vc,
                      int El, int D, int n_past, int n_ctx, void *st) {
    const fl_qact_impl *a = static_cast<const fl_qact_impl *>(a_);
    int rc = check_mm(W, a, y, ldy);
    if (rc != FL_OK) return rc;
    if (a->layout != 16) return set_error(FL_EINVAL, "gemm needs the QA16 layout");
    if (op_exact()) {
        fl_qact_impl *am = const_cast<fl_qact_impl *>(a);
        if ((rc = ensure_h16(W, am, st)) != FL_OK) return rc;
        FL_HIP(gemm_q4_exact_h16_qkv(*W, *a, a->N, y, ldy, rope_tab_dev, kc, vc, El, D, n_past, n_ctx, S(st)));
        return FL_OK;
    }
    FL_HIP(gemm_q4_mfma_qkv(*W, *a, a->N, y, ldy, rope_tab_dev, kc, vc, El, D, n_past, n_ctx, S(st)));
    return FL_OK;
}
int fl_debug_gemm_silu(const fl_qtensor *W, const fl_qact *a_, const uint16_t *silu_tab_dev, fl_qact *out_, void *st) {
    const fl_qact_impl *a = static_cast<const fl_qact_impl *>(a_);
    fl_qact_impl *out = static_cast<fl_qact_impl *>(out_);
    if (!W || !a || !out || !silu_tab_dev) return set_error(FL_EINVAL, "null argument");
    if (a->layout != 16 || a->KB != W->KB) return set_error(FL_EINVAL, "gemm_silu: bad activation workspace");
    if ((size_t)a->N16 * (size_t)(W->M / 2) > out->q_bytes) return set_error(FL_EINVAL, "gemm_silu: output workspace too small");
    out->N = a->N; out->N16 = a->N16; out->KB = W->M / 64; out->layout = 16;
    out->h16_valid = 0;
    if (op_exact()) {
        const int rc = ensure_h16(W, const_cast<fl_qact_impl *>(a), st);
        if (rc != FL_OK) return rc;
        FL_HIP(gemm_q4_exact_h16_silu(*W, *a, a->N, silu_tab_dev, *out, S(st)));
        out->h16_valid = 1;
        return FL_OK;
    }
    FL_HIP(gemm_q4_mfma_silu(*W, *a, a->N, silu_tab_dev, *out, S(st)));
    return FL_OK;
}
/* test hook: the P.V product of the reference-order prefill attention with its Q8_0 epilogue: att = soft_max'ed probabilities [H][N][n_ctx]
 * (as fl_debug_attn_exact leaves them) -> out = Q8_0 of the merged [N, E] rows (QA16 + the XH16 copy) */
int fl_debug_attn_pv_exact_q8(const float *att, int n_ctx, int D, int H, int N, int n_past, const float *vc, int E, fl_qact *out_, int compact, void *st) {
    fl_qact_impl *out = static_cast<fl_qact_impl *>(out_);
    if (!att || !vc || !out) return set_error(FL_EINVAL, "null argument");
    if (E != out->K || N > out->cap_N16 || E != H * D) return set_error(FL_EINVAL, "attn_pv_exact_q8: bad output workspace");
    out->N = N; out->N16 = fl_roundup(N, 16); out->KB = E / 32; out->layout = 16;
    FL_HIP(attn_pv_exact(att, n_ctx, (int64_t)N * n_ctx, D, H, N, n_past, vc, n_ctx, nullptr, E, S(st), out, true, compact != 0));
    out->h16_valid = 1;
    return FL_OK;
}

/* ------------------------------------------------------------------------------------------------
 * the individual eval kernels on caller-provided device buffers (per-op parity tests)
 * ---------------------------------------------------------------------------------------------- */
int fl_debug_tables(uint16_t *exp_host, uint16_t *silu_host) {   /* the two 65536-entry fp16 tables, as built for a model */
    (void)build_f16_tables(exp_host, silu_host);
    return FL_OK;
}

int fl_debug_rope_table(float *out_host, int n_ctx, int D) {     /* [n_ctx][D/2][2] {cos, sin}, as built for a model */
    build_rope_table(out_host, n_ctx, D);
    return FL_OK;
}

int fl_debug_rmsnorm_quant(const float *x, int ldx, const float *w, int N, int E, float *y_f32, int ldy, fl_qact *out,
                           int layout, void *stream) {
    M_HIP(rmsnorm_quant(x, ldx, w, N, E, y_f32, ldy, out, layout, (hipStream_t)stream, false));
    return FL_OK;
}
// (the reference-order single-token hooks run the kernel of record: it reads the tensor's QWD copy)
static int dbg_qwd(const fl_qtensor *W, void *stream) {
    if (!g_debug_exact || !W || W->qwd) return FL_OK;
    return fl_qtensor_build_qwd(const_cast<fl_qtensor *>(W), stream);
}
int fl_debug_gemv_norm(const fl_qtensor *W, const float *x, const float *norm_w, float *ynorm, float *y, void *stream) {
    if (int rc = dbg_qwd(W, stream)) return rc;
    M_HIP((g_debug_exact ? gemv_q4_norm_exact : gemv_q4_norm)(*W, x, norm_w, ynorm, y, (hipStream_t)stream));
    return FL_OK;
}
int fl_debug_gemv_silu(const fl_qtensor *W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid,
                       void *stream) {
    if (int rc = dbg_qwd(W, stream)) return rc;
    M_HIP((g_debug_exact ? gemv_q4_silu_exact : gemv_q4_silu)(*W, h13, silu_tab, y, resid, (hipStream_t)stream, false));
    return FL_OK;
}
static float *g_pa_scratch = nullptr;
static int g_pa_ld = 0;
static long g_pa_head = 0;
int fl_debug_prefill_attention(const float *qkv, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc,
                               const float *vc, const uint16_t *exp_tab_dev, float scale, float *ao, int ldo, fl_qact *qout,
                               void *stream) {
    static const int tab_n = build_f16_tables(nullptr, nullptr);      // same bound as fl_model_finalize computes
    M_HIP(prefill_attention(qkv, ldq, D, H, N, n_past, n_ctx, E, kc, vc, exp_tab_dev, tab_n, scale, ao, ldo, (hipStream_t)stream, qout,
                            g_pa_scratch, g_pa_ld, g_pa_head, g_pa_scratch ? 1 : 0));
    return FL_OK;
}
// the next fl_debug_prefill_attention calls run the key-tiled (deep-context) form with this scratch ([H][N][ld] floats); NULL: back
int fl_debug_prefill_attention_scratch(float *scratch, int ld, long head_stride) {
    g_pa_scratch = scratch; g_pa_ld = ld; g_pa_head = head_stride;
    return FL_OK;
}
int fl_debug_gemv_norm_silu(const fl_qtensor *W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act,
                            void *stream) {
    if (int rc = dbg_qwd(W, stream)) return rc;
    if (!g_debug_exact) {
        M_HIP(gemv_q4_norm_silu(*W, x, norm_w, silu_tab, act, (hipStream_t)stream));
        return FL_OK;
    }
    static float *ws = nullptr;                      // (test hook: one workspace, grown as needed, calls one at a time)
    static size_t ws_bytes = 0;
    const size_t need = gemv1_llc_pair_ws_bytes(W->M);
    if (need > ws_bytes) {
        if (ws) (void)hipFree(ws);
        ws = nullptr; ws_bytes = 0;
        M_HIP(hipMalloc((void **)&ws, need));
        M_HIP(hipMemset(ws, 0, need));
        ws_bytes = need;
    }
    M_HIP(gemv_q4_norm_silu_exact(*W, x, norm_w, silu_tab, act, (hipStream_t)stream, ws, g_debug_pair1));      // (fl_debug_set(5, form): 0 automatic)
    return FL_OK;
}
/* the woven w1|w3 matmul of the reference-order decode whose workgroups write the Q8_0 operand of w2 themselves (gemv1_q4_exact_stream.hip):
 * out = Q8_0(silu(w1 . q) * (w3 . q)) as one QA1 vector of K = M / 2 */
int fl_debug_gemv_norm_silu_q8(const fl_qtensor *W, const float *x, const float *norm_w, const uint16_t *silu_tab, fl_qact *out_, void *stream) {
    fl_qact_impl *out = static_cast<fl_qact_impl *>(out_);
    if (!W || !x || !norm_w || !silu_tab || !out) return set_error(FL_EINVAL, "null argument");
    if ((size_t)(W->M / 2) > out->q_bytes) return set_error(FL_EINVAL, "gemv_norm_silu_q8: output workspace too small");
    if (!W->qwd) { if (int rc = fl_qtensor_build_qwd(const_cast<fl_qtensor *>(W), stream)) return rc; }      // (this form reads the QWD copy)
    out->N = 1; out->N16 = 16; out->KB = W->M / 64; out->layout = 1;
    out->h16_valid = 0;
    M_HIP(gemv_q4_norm_silu_q8_exact(*W, x, norm_w, silu_tab, *out, (hipStream_t)stream));
    return FL_OK;
}
/* y = W . a (+ resid) for ONE ready-made Q8_0 vector (QA1): the wo / w2 matmul of a decode token */
int fl_debug_gemv_q8(const fl_qtensor *W, const fl_qact *a_, float *y, const float *resid, void *stream) {
    const fl_qact_impl *a = static_cast<const fl_qact_impl *>(a_);
    if (!W || !a || !y) return set_error(FL_EINVAL, "null argument");
    if (a->layout != 1 || a->KB != W->KB) return set_error(FL_EINVAL, "gemv_q8: needs one QA1 vector of the tensor's K");
    if (int rc = dbg_qwd(W, stream)) return rc;
    M_HIP((g_debug_exact ? gemv_q4_exact : gemv_q4)(*W, *a, 1, y, W->M, (hipStream_t)stream, resid, 0));
    return FL_OK;
}
int fl_debug_gemv_quant(const fl_qtensor *W, const float *x, float *y, const float *resid, void *stream) {
    if (int rc = dbg_qwd(W, stream)) return rc;
    M_HIP((g_debug_exact ? gemv_q4_quant_exact : gemv_q4_quant)(*W, x, y, resid, (hipStream_t)stream));
    return FL_OK;
}
int fl_debug_decode_attention(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab_dev,
                              float *kc, float *vc, const uint16_t *exp_tab_dev, float scale, fl_qact *out, void *stream) {
    M_HIP(decode_attention(qkv, E, D, H, n_past, n_ctx, rope_tab_dev, kc, vc, exp_tab_dev, scale, out, (hipStream_t)stream, nullptr, g_debug_exact != 0));
    return FL_OK;
}
int fl_debug_decode_attention_split(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab_dev,
                                    float *kc, float *vc, const uint16_t *exp_tab_dev, float scale, float *scores,
                                    fl_qact *out, const int *dyn_past, void *stream) {
    M_HIP(decode_attention_split(qkv, E, D, H, n_past, n_ctx, rope_tab_dev, kc, vc, exp_tab_dev, scale, scores, out,
                                 (hipStream_t)stream, dyn_past, g_debug_exact != 0));
    return FL_OK;
}
int fl_debug_silu_mul_quant(const float *h13, int ld, int N, int F, const uint16_t *silu_tab_dev, fl_qact *out, int layout,
                            void *stream) {
    M_HIP(silu_mul_quant(h13, ld, N, F, silu_tab_dev, out, layout, (hipStream_t)stream, false, false));
    return FL_OK;
}
int fl_debug_silu_mul_quant_woven(const float *h13, int ld, int N, int F, const uint16_t *silu_tab_dev, fl_qact *out, int layout,
                                  void *stream) {   /* h13 = [w1 x 16 | w3 x 16 | ...]: the woven w1|w3 matmul's output */
    M_HIP(silu_mul_quant(h13, ld, N, F, silu_tab_dev, out, layout, (hipStream_t)stream, true, false));
    return FL_OK;
}
int fl_debug_rope_kv(float *qkv, int ld, int N, int E, int D, int n_past, int n_ctx, const float *rope_tab_dev, float *kc,
                     float *vc, void *stream) {
    M_HIP(rope_kv(qkv, ld, N, E, D, n_past, n_ctx, rope_tab_dev, kc, vc, (hipStream_t)stream, nullptr));
    return FL_OK;
}
int fl_debug_gemm_f32_abt(const float *A, int lda, long sAz, const float *B, int ldb, long sBz, float *Cc, int ldc, long sCz,
                          int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past, void *stream) {
    M_HIP(gemm_f32_abt(A, lda, sAz, B, ldb, sBz, Cc, ldc, sCz, M, Nn, K, batch, alpha, causal_mode, n_past, (hipStream_t)stream, nullptr, 0));
    return FL_OK;
}
int fl_debug_gemm_f32_abt_exact(const float *A, int lda, long sAz, const float *B, int ldb, long sBz, float *Cc, int ldc, long sCz,
                                int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past, void *stream) {
    M_HIP(dot_f32_abt_exact(A, lda, sAz, B, ldb, sBz, Cc, ldc, sCz, M, Nn, K, batch, alpha, causal_mode, n_past, (hipStream_t)stream, nullptr, 0));
    return FL_OK;
}
/* test hook: exact-mode prefill attention on caller-provided buffers: scores (MFMA form when which >= 1, one half-wave per dot when 0)
 * -> soft_max -> P.V; att: [H][N][n_ctx] scratch, ao: [N][E] f32 result.  which = 2: the probabilities compact between the launches (contexts of
 * 513 .. 2048 keys); which = 3: that with K.Q and soft_max as one launch (up to 1024 keys) */
int fl_debug_attn_exact(const float *qkv, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc, const float *vc,
                        const uint16_t *exp_tab_dev, float scale, float *att, float *ao, int which, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int P = n_past + N;
    const bool compact = which >= 2;
    if (which == 3) {
        M_HIP(attn_scores_softmax_exact(qkv, ldq, D, H, N, n_past, kc, E, scale, att, n_ctx, (int64_t)N * n_ctx, exp_tab_dev, st, true));
    } else {
        if (which) M_HIP(attn_scores_exact(qkv, ldq, D, H, N, n_past, kc, E, scale, att, n_ctx, (int64_t)N * n_ctx, st));
        else M_HIP(dot_f32_abt_exact(qkv, ldq, D, kc, E, D, att, n_ctx, (int64_t)N * n_ctx, N, P, D, H, scale, 1, n_past, st, nullptr, 0));
        M_HIP(softmax_rows(att, n_ctx, (int64_t)N * n_ctx, N, P, n_past, H, exp_tab_dev, st, nullptr, compact));
    }
    if (which) M_HIP(attn_pv_exact(att, n_ctx, (int64_t)N * n_ctx, D, H, N, n_past, vc, n_ctx, ao, E, st, nullptr, false, compact));
    else M_HIP(dot_f32_abt_exact(att, n_ctx, (int64_t)N * n_ctx, vc, n_ctx, (int64_t)D * n_ctx, ao, E, D, N, D, P, H, 1.0f, 2, n_past, st, nullptr, 0));
    return FL_OK;
}
int fl_debug_softmax_rows(float *S, int ld, long sz, int N, int P, int n_past, int batch, const uint16_t *exp_tab_dev,
                          void *stream) {
    M_HIP(softmax_rows(S, ld, sz, N, P, n_past, batch, exp_tab_dev, (hipStream_t)stream, nullptr, false));
    return FL_OK;
}

}  // extern "C"
