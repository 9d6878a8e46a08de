// internal.h -- what the translation units of libfastllama_hip.so share beyond the launchers (q4_kernels.h, eval_kernels.h), and what
// the test-hook library (test_hooks.cpp -> libfastllama_hip_hooks.so, linked against this one) reaches into.  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/fastllama_hip.h"
#include "q4_kernels.h"
#include "eval_kernels.h"
#include "runtime.h"

struct fl_qact_impl : fl_qact {
    int cap_N16, K;
    int layout;  // 16 or 1
    size_t q_bytes, s_bytes;
    size_t h16_bytes;   // bytes allocated for the XH16 copy (whole 32-column tiles of the K it was created with)
    int h16_valid;   // the XH16 copy (q4_layout.h) matches q: written by fl_quantize_q8* in reference-order mode, a fused epilogue, or on demand
};

namespace fl {
// capi.cpp
extern int g_op_mode;                    // fl_set_op_mode
bool op_exact();                         // the operator-level entry points run the reference's summation order
int ensure_h16(const fl_qtensor *W, fl_qact_impl *a, void *stream);      // the H16 copies of both operands of a reference-order GEMM
int check_mm(const fl_qtensor *W, const fl_qact_impl *a, const float *y, int ldy);
int mul_mat_q_which(const fl_qtensor *W, const fl_qact *a, float *y, int ldy, int which, void *stream);
// kernel-selection overrides (tuning sweeps and tests; -1 / 0 = automatic): gemm_q4_mfma.hip, q4_kernels.hip
extern int g_gemm_force_cfg, g_gemv_force_waves;
extern int g_stream_helpers;              // gemv1_q4_exact_stream.hip: prologue-only waves in wq|wk|wv-sized launches (1 / 0)
extern int g_pv_waves;                     // exact_kernels.hip: waves per workgroup of the reference-order V.P kernel behind a deep context (8 / 4)
extern int g_stream_force_nw;              // gemv1_q4_exact_stream.hip: streaming waves per workgroup (0: automatic -- one workgroup per CU)
extern int g_stream_min_groups;            // gemv1_q4_exact_stream.hip: row groups from which a matrix takes the one-wave-per-row-group form (-1: automatic)
void gemm32_mixed_split(int MGT, int NGT, int *n_a, int *mg_split, int *n_b);   // gemm_q4_mfma32.hip
// model.cpp
int build_f16_tables(uint16_t *exp_tab, uint16_t *silu_tab);
void build_rope_table(float *rt, int n_ctx, int D);
}  // namespace fl

// ---- what the test-hook library reaches inside the product library -------------------------------------------------------------
// libfastllama_hip.so is built with -fvisibility=hidden: its dynamic symbol table holds the 17 llama_* symbols, the fl_* API of
// include/fastllama_hip.h and ONE more C symbol, fl_internal_table(), which hands the hook library (same process, same globals) the
// launchers and switches below.  test_hooks.cpp calls them through the table; nothing else can.
#define FL_INTERNAL_FUNCS(X) \
    X(attn_pv_exact) \
    X(attn_scores_exact) \
    X(attn_scores_softmax_exact) \
    X(build_f16_tables) \
    X(build_rope_table) \
    X(check_mm) \
    X(decode_attention) \
    X(decode_attention_split) \
    X(dot_f32_abt_exact) \
    X(ensure_h16) \
    X(gemm32_mixed_split) \
    X(gemm_f32_abt) \
    X(gemm_q4_exact_h16) \
    X(gemm_q4_exact_h16_qkv) \
    X(gemm_q4_exact_h16_silu) \
    X(gemm_q4_mfma) \
    X(gemm_q4_mfma_qkv) \
    X(gemm_q4_mfma_silu) \
    X(gemv1_llc_pair_ws_bytes) \
    X(gemv_q4) \
    X(gemv_q4_exact) \
    X(gemv_q4_norm) \
    X(gemv_q4_norm_exact) \
    X(gemv_q4_norm_silu) \
    X(gemv_q4_norm_silu_exact) \
    X(gemv_q4_norm_silu_q8_exact) \
    X(gemv_q4_quant) \
    X(gemv_q4_quant_exact) \
    X(gemv_q4_silu) \
    X(gemv_q4_silu_exact) \
    X(hip_fail) \
    X(mul_mat_q_which) \
    X(op_exact) \
    X(prefill_attention) \
    X(rmsnorm_quant) \
    X(rope_kv) \
    X(set_error) \
    X(silu_mul_quant) \
    X(softmax_rows)
namespace fl {
struct InternalTable {
#define X(name) decltype(&fl::name) name;
    FL_INTERNAL_FUNCS(X)
#undef X
    int *g_gemm_force_cfg, *g_gemv_force_waves, *g_op_mode, *g_stream_min_groups, *g_stream_force_nw, *g_pv_waves, *g_stream_helpers;
};
}  // namespace fl
extern "C" __attribute__((visibility("default"))) const fl::InternalTable *fl_internal_table(void);
