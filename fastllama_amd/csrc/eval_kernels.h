// eval_kernels.h -- launchers of the non-matmul eval kernels (see eval_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "q4_layout.h"

namespace fl {

// rms_norm(x) * w -> optional f32 copy y_f32 (may be null) -> optional Q8_0 workspace `out` (may be null)
// with_h16 (layout 16): also the XH16 copy of the quants (q4_layout.h), the operand form of the reference-order prefill GEMM
// row-split tensor parallelism, prefill (model.cpp): x = all-gathered rows + residual, written, then rms_norm * w -> Q8_0 -- one launch; and the
// all-gathered packed QA16 planes of every rank's K blocks -> the consumer's d / s planes + XH16 copy (+ q plane) -- one launch
hipError_t rmsnorm_quant_gathered(const float *gathered, int G, int El, const float *resid, int ldr, float *x, int ldx, const float *w, int N, int E,
                                  float *y_f32, int ldy, const fl_qact *out, int layout, hipStream_t st, bool with_h16);
hipError_t gathered_qa16_to_operand(const void *stage, size_t msg_bytes, int G, int KBl, int N, const fl_qact &full, bool with_q, bool with_h16, hipStream_t st);
hipError_t rmsnorm_quant(const float *x, int ldx, const float *w, int N, int E, float *y_f32, int ldy,
                         const fl_qact *out, int layout, hipStream_t st, bool with_h16 = false);
// silu_table(h13[:, :F]) * h13[:, F:2F] -> Q8_0
hipError_t silu_mul_quant(const float *h13, int ld, int N, int F, const uint16_t *silu_tab, const fl_qact *out,
                          int layout, hipStream_t st,
                          bool woven = false,    // woven: h13 = [w1 x 16 | w3 x 16 | ...] instead of [w1 x (F) | w3 x (F)]
                          bool with_h16 = false);
// rope on q (in place) and k (-> kc rows n_past..), v -> vc columns n_past..
hipError_t rope_kv(float *qkv, int ld, int N, int E, int D, int n_past, int n_ctx, const float *rope_tab, float *kc,
                   float *vc, hipStream_t st, const int *dyn_past = nullptr);
// C[z] = alpha * A[z] * B[z]^T on the exact-f32 MFMA; causal_mode 0 none, 1 scores, 2 KQV.
// dyn_past (decode hipGraph): n_past is read from device memory, so one captured graph serves every position.
hipError_t gemm_f32_abt(const float *A, int lda, int64_t sAz, const float *B, int ldb, int64_t sBz, float *C, int ldc,
                        int64_t sCz, int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past,
                        hipStream_t st, const int *dyn_past = nullptr, int nn_max = 0);
// the same interface in ggml_vec_dot_f32's order (4 x 8 FMA lanes, fixed tree, compiled leftover loop): exact_kernels.hip
hipError_t dot_f32_abt_exact(const float *A, int lda, int64_t sAz, const float *B, int ldb, int64_t sBz, float *C, int ldc,
                             int64_t sCz, int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past,
                             hipStream_t st, const int *dyn_past = nullptr, int nn_max = 0);
// exact mode, prefill (any context length): the same two products on the f32-input MFMA, whose k = 0, 1 chain IS the reference's
// fma chain (exact_kernels.hip); hipErrorInvalidValue: shape outside their reach -> dot_f32_abt_exact
hipError_t attn_scores_exact(const float *qkv, int ldq, int D, int H, int N, int n_past, const float *kc, int ldk, float scale,
                             float *att, int ld_att, int64_t head_stride, hipStream_t st);
hipError_t attn_scores_softmax_exact(const float *qkv, int ldq, int D, int H, int N, int n_past, const float *kc, int ldk, float scale,
                                     float *att, int ld_att, int64_t head_stride, const uint16_t *exp_tab, hipStream_t st,
                                     bool compact = false);   // + soft_max in the same launch (<= 1024 keys)
// out != NULL: the result leaves as the Q8_0 operand of the wo matmul (QA16, K = ldo; with_h16: + its XH16 copy) instead of f32 rows in ao
hipError_t attn_pv_exact(const float *att, int ld_att, int64_t head_stride, int D, int H, int N, int n_past, const float *vc, int n_ctx,
                         float *ao, int ldo, hipStream_t st, const fl_qact *out = nullptr, bool with_h16 = false, bool compact = false);
// compact: the probabilities leave as fp16 table values + one f32 factor per row (softmax_rows_reg_kernel); attn_pv_exact(compact) reads that
// dst[0] = n_past, dst[1] = token: what a replayed decode graph reads its position and its token from -- set by a launch, whose arguments are copied
// when it is enqueued (two 4-byte copies from one pinned staging pair raced with the host's next token when the caller does not wait between evals)
hipError_t set_decode_inputs(int *dst, int n_past, int token, hipStream_t st);
hipError_t softmax_rows(float *S, int ld, int64_t sz, int N, int P, int n_past, int batch, const uint16_t *exp_tab,
                        hipStream_t st, const int *dyn_past = nullptr, bool compact = false);
// prefill: KQ*scale + mask + soft_max + KQV per (head, 32 query rows), score rows in LDS; q = roped Q rows of qkv,
// kc / vc already hold the new positions.  When the rows do not fit LDS (deep contexts) the key-tiled form runs instead:
// same launch count, the scores take one trip through `scratch` ([H] x s_head floats, rows of ld_s >= n_past + N floats,
// ld_s % 32 == 0).  hipErrorInvalidValue: shape fits neither form -> use the three-kernel path.
hipError_t prefill_attention(const float *qkv, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc,
                             const float *vc, const uint16_t *exp_tab, int tab_n, float scale, float *ao, int ldo,
                             hipStream_t st, const fl_qact *qout = nullptr,   // qout: write Q8_0 (QA16) instead of ao
                             float *scratch = nullptr, int ld_s = 0, int64_t s_head = 0, int force_deep = 0);
hipError_t prefill_attention_deep(const float *qkv, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc,
                                  const float *vc, const uint16_t *exp_tab, int tab_n, float scale, float *scratch, int ld_s,
                                  int64_t s_head, float *ao, int ldo, hipStream_t st, const fl_qact *qout);
// decode (N = 1): rope + KV store + KQ + soft_max + KQV + Q8_0 of the result, one workgroup per head
hipError_t decode_attention(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab, float *kc,
                            float *vc, const uint16_t *exp_tab, float scale, const fl_qact *out, hipStream_t st,
                            const int *dyn_past = nullptr, bool exact = false);   // exact: ggml_vec_dot_f32's order in both dots
// the same result (bit for bit) from two launches that spread a head over many CUs: (head, 128 positions) scores into
// `scores` [H][n_ctx], then (head, 32 features) soft_max + KQV + Q8_0.  For long contexts.
hipError_t decode_attention_split(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab,
                                  float *kc, float *vc, const uint16_t *exp_tab, float scale, float *scores,
                                  const fl_qact *out, hipStream_t st, const int *dyn_past = nullptr, bool exact = false);
// LoRA merge on reference AoS blocks (lora_kernels.hip): rows [row0, row0+rows) <- quantize(dequantize + sign * BA)
hipError_t lora_add_aos(int type, void *aos, int KB, int row0, int rows, int il_part, const float *ba, int64_t ldba, const float *A,
                        const float *B, int r, int ba_row0, int ba_col0, float sign, hipStream_t st);
// quantize_row_q4_{0,1} (SIMD flavour) / *_reference on AoS blocks: f32 [k] -> block_q4_x [k/32]
hipError_t quantize_row_q4_aos(int type, bool reference, const float *x, void *y, int64_t k, hipStream_t st);
hipError_t logits_nll(const float *logits, int ld, int V, const int *next_tok_dev, int j0, int rows, double *out_dev, hipStream_t st);
hipError_t gather_cols(const float *tmp, int G, int N, int Vl, int ldp, float *out, int ldo, hipStream_t st);
hipError_t add_rows(const float *a, int lda, const float *b, int ldb, float *o, int ldo, int N, int E, hipStream_t st);
// row-split tensor parallelism (reference-order mode): the byte work around its all-gathers (eval_kernels.hip)
hipError_t pack3(void *dst, const void *s0, size_t b0, const void *s1, size_t b1, const void *s2, size_t b2, hipStream_t st);
hipError_t unpack3(const void *stage, size_t msg_bytes, int G, int rows, void *o0, size_t chunk0, void *o1, size_t chunk1, void *o2,
                   size_t chunk2, hipStream_t st);
hipError_t gather_rows_add(const float *tmp, int G, int N, int Ml, const float *resid, int ldr, float *out, int ldo, hipStream_t st);

}  // namespace fl
