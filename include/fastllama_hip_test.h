/* fastllama_hip_test.h -- the fl_debug_* entry points of libfastllama_hip_hooks.so (fastllama_amd/csrc/test_hooks.cpp): single kernels
 * and fused forms on caller-provided device buffers, kernel-family selection, table builders.  For the per-op parity tests
 * (tests/test_*_gpu.py), the tuning sweeps under scripts/ and A/B measurements; not part of the drop-in boundary, and not exported by
 * libfastllama_hip.so.  Link (or dlopen) libfastllama_hip_hooks.so next to libfastllama_hip.so: it resolves against the product
 * library loaded in the same process. */
#ifndef FASTLLAMA_HIP_TEST_H
#define FASTLLAMA_HIP_TEST_H
#include "fastllama_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* test hooks: the individual non-matmul eval kernels (fastllama_amd/csrc/eval_kernels.hip) and their tables */
int fl_debug_tables(uint16_t *exp_host, uint16_t *silu_host);
int fl_debug_rope_table(float *out_host, int n_ctx, int D);
int fl_debug_rmsnorm_quant(const float *x_dev, int ldx, const float *w_dev, int N, int E, float *y_f32_dev, int ldy,
                           fl_qact *out, int layout, void *stream);
int fl_debug_gemv_norm(const fl_qtensor *W, const float *x_dev, const float *norm_w_dev, float *ynorm_dev, float *y_dev,
                       void *stream);  /* y = W . Q8_0(norm_w * rms_norm(x)), one launch (decode) */
int fl_debug_gemv_silu(const fl_qtensor *W, const float *h13_dev, const uint16_t *silu_tab_dev, float *y_dev,
                       const float *resid_dev, void *stream); /* y = W . Q8_0(silu(h13[:K]) * h13[K:]) + resid, one launch */
int fl_debug_gemv_norm_silu(const fl_qtensor *W_woven, const float *x_dev, const float *norm_w_dev, const uint16_t *silu_tab_dev,
                            float *act_dev, void *stream);  /* act = silu(w1.q) * (w3.q), q = Q8_0(norm_w * rms_norm(x)) */
int fl_debug_gemv_norm_silu_q8(const fl_qtensor *W_woven, const float *x_dev, const float *norm_w_dev, const uint16_t *silu_tab_dev,
                               fl_qact *out, void *stream);  /* reference order only: out = Q8_0(act) as one QA1 vector, written by the matmul's workgroups */
int fl_debug_gemv_q8(const fl_qtensor *W, const fl_qact *a /* one QA1 vector */, float *y_dev, const float *resid_dev, void *stream);  /* y = W . a (+ resid): the wo / w2 matmul of a decode token */
int fl_debug_gemv_quant(const fl_qtensor *W, const float *x_dev, float *y_dev, const float *resid_dev, void *stream);
int fl_debug_prefill_attention(const float *qkv_dev, int ldq, int D, int H, int N, int n_past, int n_ctx, int E,
                               const float *kc, const float *vc, const uint16_t *exp_tab_dev, float scale, float *ao_dev,
                               int ldo, fl_qact *qout /* NULL: f32 result to ao; else Q8_0 (QA16) of it */, void *stream);
                               /* KQ*scale + mask + soft_max + KQV (+ quantize_row_q8_0), one launch (prefill) */
int fl_debug_prefill_attention_scratch(float *scratch_dev, int ld, long head_stride);
                               /* non-NULL: the following fl_debug_prefill_attention calls run the key-tiled (deep-context) form,
                                  scores staged in scratch [H][head_stride], rows of ld floats (ld % 32 == 0, ld >= n_past + N) */
int fl_debug_decode_attention(const float *qkv_dev, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab_dev,
                              float *kc, float *vc, const uint16_t *exp_tab_dev, float scale, fl_qact *out, void *stream);
int fl_debug_decode_attention_split(const float *qkv_dev, int E, int D, int H, int n_past, int n_ctx,
                                    const float *rope_tab_dev, float *kc, float *vc, const uint16_t *exp_tab_dev, float scale,
                                    float *scores_dev /* [H][n_ctx] workspace */, fl_qact *out,
                                    const int *dyn_past_dev /* NULL, or the position in device memory (grid sized for n_ctx) */,
                                    void *stream);  /* the long-context form of the above: two launches, same bits */
int fl_debug_silu_mul_quant(const float *h13_dev, int ld, int N, int F, const uint16_t *silu_tab_dev, fl_qact *out,
                            int layout, void *stream);
int fl_debug_silu_mul_quant_woven(const float *h13_dev, int ld, int N, int F, const uint16_t *silu_tab_dev, fl_qact *out,
                                  int layout, void *stream);
int fl_debug_rope_kv(float *qkv_dev, int ld, int N, int E, int D, int n_past, int n_ctx, const float *rope_tab_dev,
                     float *kc_dev, float *vc_dev, void *stream);
int fl_debug_gemm_f32_abt(const float *A, int lda, long sAz, const float *B, int ldb, long sBz, float *C, int ldc, long sCz,
                          int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past, void *stream);
int fl_debug_gemm_f32_abt_exact(const float *A, int lda, long sAz, const float *B, int ldb, long sBz, float *C, int ldc, long sCz,
                                int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past, void *stream);
                                /* the same product in ggml_vec_dot_f32's order (exact mode's attention matmuls) */
int fl_debug_attn_exact(const float *qkv_dev, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc_dev,
                        const float *vc_dev, const uint16_t *exp_tab_dev, float scale, float *att_dev /* [H][N][n_ctx] scratch */,
                        float *ao_dev /* [N][E] */, int which /* 1: MFMA forms, 0: one half-wave per dot, 2: MFMA forms with the probabilities compact between soft_max and P.V (513 .. 2048 keys), 3: that with K.Q and soft_max as one launch (<= 1024 keys) */, void *stream);
int fl_debug_softmax_rows(float *S_dev, int ld, long sz, int N, int P, int n_past, int batch, const uint16_t *exp_tab_dev,
                          void *stream);
int fl_debug_attn_pv_exact_q8(const float *att_dev /* probabilities, as fl_debug_attn_exact leaves them */, int n_ctx, int D, int H, int N,
                              int n_past, const float *vc_dev, int E, fl_qact *out /* Q8_0 of the [N][E] result */,
                              int compact /* att_dev as fl_debug_attn_exact(which >= 2) leaves it */, void *stream);

/* test hooks: force one kernel family regardless of N (N must suit the layout of `a`) */
/* the fused forms of the prefill GEMM: + residual; wq|wk|wv with rope + KV-cache stores; woven w1|w3 with silu*mul -> Q8_0 */
int fl_debug_mul_mat_q_resid(const fl_qtensor *W, const fl_qact *a, float *y_dev, int ldy, const float *resid_dev, int ldr, void *stream);
int fl_debug_gemm_qkv(const fl_qtensor *W, const fl_qact *a, float *y_dev, int ldy, const float *rope_tab_dev, float *kc_dev,
                      float *vc_dev, int El, int D, int n_past, int n_ctx, void *stream);
int fl_debug_gemm_silu(const fl_qtensor *W, const fl_qact *a, const uint16_t *silu_tab_dev, fl_qact *out, void *stream);
int fl_debug_mul_mat_q(const fl_qtensor *W, const fl_qact *a, float *y_dev, int ldy, int which /* 0 naive, 1 fast MFMA GEMM, 2 fast GEMV, 3 reference order (kernel of record for the layout and N), 4 reference-order VALU tiles, 5 reference-order H16 tiles, 6 round 3's reference-order tiles */,
                       void *stream);
int fl_debug_qact_layout(const fl_qact *a); /* 16 = QA16, 1 = QA1 */
int fl_debug_gemm_mixed_split(int row_groups, int col_groups, int *n_a, int *mg_split, int *n_b);  /* host logic of the two-tile-shape launch */
int fl_debug_set(int what, int value);      /* what 0: force GEMM tile configuration id (-1 = automatic); 1: GEMV waves per row group (0 = automatic); 2: the single-token hooks run the reference-order kernels; 4: = fl_set_op_mode; 5: the form of the reference-order w1|w3 kernel fl_debug_gemv_norm_silu runs (1 / 2, 0 automatic); 6: row groups from which a reference-order N = 1 matmul takes the one-wave-per-row-group form (-1 automatic, 1 always, 1 << 30 never); 7: row groups per workgroup of that form (0 automatic: one workgroup per CU); 8: waves per workgroup of the reference-order V.P kernel behind a deep context (8, the default: two waves per SIMD; 4: the one-wave-per-SIMD form); 9: prologue-only waves in front of the streaming ones of the one-wave-per-row-group form (1, the default / 0) */

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* FASTLLAMA_HIP_TEST_H */
