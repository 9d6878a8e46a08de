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
void gemm32_mixed_split(int MGT, int NGT, int *n_a, int *mg_split, int *n_b);   // gemm_q4_mfma32.hip
// model.cpp
int build_f16_tables(uint16_t *exp_tab, uint16_t *silu_tab);
void build_rope_table(float *rt, int n_ctx, int D);
}  // namespace fl
