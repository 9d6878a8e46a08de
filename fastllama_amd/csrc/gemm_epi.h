// gemm_epi.h -- optional fused epilogues shared by the two MFMA GEMM kernels (gemm_q4_mfma.hip, gemm_q4_mfma32.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fl {

// Optional epilogue for the fused w1|w3 matrix (rows interleaved by 16-row groups: even groups w1, odd groups w3, see
// model.cpp): instead of storing y, the workgroup computes silu(w1 x) * (w3 x) for its features, and writes it
// re-quantized to Q8_0 in the QA16 layout the w2 matmul reads -- ggml_silu + ggml_mul (lib/llama.cpp:428-431) and the
// INIT phase of the next mul_mat (quantize_row_q8_0) without the [N][2 n_ff] f32 round trip through HBM.
struct GemmSiluEpi {
    const uint16_t *silu_tab;   // fp16 SiLU table (null: plain store)
    int8_t *oq;
    float *od, *os;
    int KBo;                    // n_ff / 32
    uint16_t *oh;               // optional XH16 copy of the output (q4_layout.h), written next to oq
    // second optional epilogue, for the fused wq|wk|wv matrix: rope on Q (stored to y) and on K (stored to the K cache
    // row of the token's position), V stored transposed into the V cache -- ggml_rope + the two ggml_cpy into memory_k /
    // memory_v (lib/llama.cpp:328-347) without a pass over the [N][3 n_embd] result.  Arithmetic of rope_kv_kernel.
    const float2 *rope_tab;     // [n_ctx][D/2] {cos, sin} (null: off)
    float *kc, *vc;             // [n_ctx][El], [El][n_ctx]
    int El, D, n_past, n_ctx;
};

}  // namespace fl
