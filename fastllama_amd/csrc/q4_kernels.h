// q4_kernels.h -- host-callable launchers of the HIP kernels (internal C++ interface; the public
// C-ABI is include/fastllama_hip.h, implemented in capi.cpp on top of these).
#pragma once
#include <hip/hip_runtime.h>
#include "q4_layout.h"

namespace fl {

// ---- weights: AoS blocks (reference file layout, device) <-> QW16 ----
hipError_t repack_to_qw16(int type, const void *aos_dev, int M, int K, uint32_t *qs, float *d, float *m,
                          int *bad_scale_flag_dev, hipStream_t st);
hipError_t unpack_from_qw16(int type, const uint32_t *qs, const float *d, const float *m, int M, int K,
                            void *aos_dev, hipStream_t st);

// ---- a4: quantize_row_q8_0 (lib/ggml.c:1299-1441), three output layouts ----
// x: N rows of K floats, row stride ldx (elements).
hipError_t quantize_q8_aos(const float *x, int ldx, int N, int K, void *aos_dev, hipStream_t st);
hipError_t quantize_q8_qa16(const float *x, int ldx, int N, int K, const fl_qact &out, hipStream_t st,
                            bool with_h16 = false);   // with_h16: also out.h16, the XH16 copy (q4_layout.h)
hipError_t quantize_q8_qa1(const float *x, int ldx, int N, int K, const fl_qact &out, hipStream_t st);
// QA16 / QA1 workspace -> reference AoS block_q8_0 (parity tests of the internal quantizers)
hipError_t export_qa16_to_aos(const fl_qact &in, int N, int K, void *aos_dev, hipStream_t st);
hipError_t export_qa1_to_aos(const fl_qact &in, int N, int K, void *aos_dev, hipStream_t st);

// ---- a7: dequantize_row_q4_{0,1} on AoS blocks (lib/ggml.c:1443-1665) ----
hipError_t dequantize_aos(int type, const void *aos_dev, float *y, int64_t k, hipStream_t st);
// dequantize selected rows of a QW16 tensor (get_rows_q, lib/ggml.c:8333-8360)
hipError_t get_rows_qw16(const fl_qtensor &W, const int *rows_dev, int nrows, float *y, int ldy, hipStream_t st);

// ---- a5/a6: ggml_vec_dot_q4_{0,1}_q8_0 on AoS operands (lib/ggml.c:2368-2714) ----
hipError_t vec_dot_aos(int type, int n, float *s_dev, const void *x_aos, const void *y_aos, hipStream_t st);

// ---- a9: mul_mat_q_f32 compute phase on QW16 x QA* ----
// y: N rows of M floats, row stride ldy.
// resid (optional): y = mul_mat + resid, the ggml_add that follows wo / w2 (lib/llama.cpp:407,441), fused into the store
hipError_t gemv_q4_norm(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st);
hipError_t gemv_q4_silu(const fl_qtensor &W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid,
                        hipStream_t st, bool woven = false);
hipError_t gemv_q4_norm_silu(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act,
                             hipStream_t st);
hipError_t gemv_q4_quant(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st);
hipError_t gemv_q4(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                   const float *resid = nullptr, int ldr = 0);
hipError_t gemm_q4_mfma(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                        const float *resid = nullptr, int ldr = 0);
hipError_t gemm_q4_mfma_silu(const fl_qtensor &W, const fl_qact &xq, int N, const uint16_t *silu_tab, const fl_qact &out,
                             hipStream_t st);
hipError_t gemm_q4_mfma_qkv(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, const float *rope_tab, float *kc,
                            float *vc, int El, int D, int n_past, int n_ctx, hipStream_t st);
// reference-order ("exact") forms: 8 lane accumulators per output in block order + the AVX2 hsum (exact_kernels.hip)
hipError_t gemv_q4_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                         const float *resid = nullptr, int ldr = 0);
hipError_t gemv_q4_norm_exact(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st);
hipError_t gemv_q4_silu_exact(const fl_qtensor &W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid,
                              hipStream_t st, bool woven = false);
// pair_ws: gemv1_llc_pair_ws_bytes(W.M) bytes of zeroed device memory owned by the caller's eval (one launch at a time per workspace): the
// w1 / w3 groups of a feature then run as separate workgroups and the second to finish forms silu * mul; NULL: one workgroup per pair
// form: 0 automatic; 1 one workgroup per feature pair, groups in turn; 2 two workgroups meeting in pair_ws
hipError_t gemv_q4_norm_silu_exact(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act,
                                   hipStream_t st, float *pair_ws = nullptr, int form = 0);
size_t gemv1_llc_pair_ws_bytes(int M);
hipError_t gemv_q4_quant_exact(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st);
hipError_t gemm_q4_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                         const float *resid = nullptr, int ldr = 0);
hipError_t gemm_q4_exact_mfma(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                              const float *resid = nullptr, int ldr = 0);   // K = 4 f16 MFMA form (gemm_q4_exact_mfma.hip)
// round 4 form: ready-made f16 fragments (WH16 x XH16, q4_layout.h) by LDS-DMA, optional fused epilogues (gemm_q4_exact_h16.hip)
size_t wh16_bytes(const fl_qtensor &W);
size_t xh16_bytes(int N, int K);
hipError_t qw16_to_h16(const fl_qtensor &W, uint16_t *wh, hipStream_t st);   // W.qs (QW16) -> WH16
hipError_t qa16_to_h16(const fl_qact &xq, int N, hipStream_t st);            // xq.q (QA16) -> xq.h16
bool gemm_q4_exact_h16_supports(const fl_qtensor &W, const fl_qact &xq, int N);
hipError_t gemm_q4_exact_h16(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                             const float *resid = nullptr, int ldr = 0);
hipError_t gemm_q4_exact_h16_qkv(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, const float *rope_tab, float *kc,
                                 float *vc, int El, int D, int n_past, int n_ctx, hipStream_t st);
hipError_t gemm_q4_exact_h16_silu(const fl_qtensor &W, const fl_qact &xq, int N, const uint16_t *silu_tab, const fl_qact &out,
                                  hipStream_t st);
// round 4 form of the N = 1 reference-order kernel: lane-local chains on the QWD nibble copy (gemv1_q4_exact_llc.hip); the bool
// launchers return false when the tensor has no QWD copy or the shape is outside the kernel's reach (-> round 3's kernel)
size_t qwd_bytes(const fl_qtensor &W);
hipError_t qw16_to_qwd(const fl_qtensor &W, uint32_t *qwd, hipStream_t st);
hipError_t qwd_to_qw16(const fl_qtensor &W, uint32_t *qs, hipStream_t st);   // the inverse: W.qwd -> a QW16 nibble plane (lean memory mode, model.cpp)
bool gemv1_llc(const fl_qtensor &W, const fl_qact &xq, float *y, hipStream_t st, const float *resid);
bool gemv1_llc_norm(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st);
bool gemv1_llc_silu(const fl_qtensor &W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid, hipStream_t st, bool woven);
bool gemv1_llc_norm_silu(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act, hipStream_t st,
                         float *pair_ws, int form);
bool gemv1_llc_quant(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st);
// round 6 form for matrices of many row groups: one wave per row group streaming the whole of K, four row groups per workgroup sharing one
// activation prologue (gemv1_q4_exact_stream.hip); false: no QWD copy / too few row groups / shape outside its reach (-> the llc kernel).
// gemv1_stream_norm_silu_q8: the woven w1|w3 matmul whose workgroups write the Q8_0 operand of the w2 matmul themselves (out: QA1 planes)
// the next gemv1_stream_norm launch of this thread also touches the K / V history its decode layer's attention will read (an L2 hint; nullptr: none)
void gemv1_stream_offer_kv_prefetch(const float *kc, const float *vc, const int *pos_dev, int E, int D, int H, int n_ctx);
bool gemv1_stream(const fl_qtensor &W, const fl_qact &xq, float *y, hipStream_t st, const float *resid);
bool gemv1_stream_norm(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st);
bool gemv1_stream_quant(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st);
bool gemv1_stream_norm_silu(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act, hipStream_t st);
bool gemv1_stream_norm_silu_q8(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, const fl_qact &out, hipStream_t st);
hipError_t gemv_q4_norm_silu_q8_exact(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, const fl_qact &out,
                                      hipStream_t st);   // hipErrorInvalidValue: shape outside the form's reach (-> f32 features + gemv_q4_quant_exact)
hipError_t gemm_q4_exact_valu(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                              const float *resid = nullptr, int ldr = 0);   // v_dot4 form (exact_kernels.hip), cross-check
hipError_t gemm_q4_naive(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st);


size_t qact_bytes_q(int N, int K);      // bytes of the q plane for N columns (padded to 16)
size_t qact_bytes_scale(int N, int K);  // bytes of one scale plane

}  // namespace fl
