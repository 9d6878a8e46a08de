// lora_kernels.hip -- LoRA merge on resident Q4 weights.
//
// Reference: Model::attach_lora / detach_lora (/root/reference/lib/llama.cpp:697-944) build, per adapted tensor,
//   BA = ggml_mul_mat(loraA[r,K], loraB[r,M])        f32 x f32 -> ggml_vec_dot_f32        (lib/ggml.c:2295-2330)
//   W  = ggml_add_inplace(W_q4, +-BA)                 ggml_compute_forward_add_q_f32       (lib/ggml.c:6414-6520)
// i.e. row by row: dequantize_row_q -> += BA row -> quantize_row_q, the latter being the SIMD quantizer (AVX2 on the
// reference's x86 build: :757-803 for Q4_0, :965-1037 for Q4_1), not the *_reference one.  This file restates exactly
// that arithmetic (its CPU twin lives in oracle/q4_oracle.c, pinned byte for byte to the reference) on the
// reference's AoS blocks; the caller unpacks QW16 -> AoS before and repacks after (a load-time operation, not a hot path).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "eval_kernels.h"
#include "q4_layout.h"

// hipcc contracts a*b+c into an FMA by default -- also across the header-defined __fmul_rn/__fadd_rn, whose operations carry
// the header's contract flag -- so this file uses plain operators under contract(off); every fusion the reference's build
// has is written out as __fmaf_rn, everything else stays a separately rounded operation.
#pragma clang fp contract(off)

namespace fl {

// ggml_vec_dot_f32 as compiled for AVX2: 4 accumulators of 8 lanes over 32-element steps (FMA), the reduction tree of
// GGML_F32x8_REDUCE, then the n % 32 leftovers as rounded products added in order.
__device__ __forceinline__ float vec_dot_f32_avx2(int n, const float *__restrict__ x, const float *__restrict__ y) {
    float sum[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int l = 0; l < 8; ++l) sum[j][l] = 0.f;
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int l = 0; l < 8; ++l) sum[j][l] = __fmaf_rn(x[i + 8 * j + l], y[i + 8 * j + l], sum[j][l]);
    float t0[4];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        sum[0][l] = ((sum[0][l]) + (sum[1][l]));
        sum[2][l] = ((sum[2][l]) + (sum[3][l]));
        sum[0][l] = ((sum[0][l]) + (sum[2][l]));
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) t0[l] = ((sum[0][l]) + (sum[0][l + 4]));
    float sumf = ((((t0[0]) + (t0[1]))) + (((t0[2]) + (t0[3]))));
    for (int i = np; i < n; ++i) sumf = ((sumf) + (((x[i]) * (y[i]))));
    return sumf;
}

// one thread = one quant block of one row.  aos: rows_total x KB blocks (20 / 24 B); rows [row0, row0+rows) are merged.
// delta(m, k) = ba[(ba_row0 + m) * ldba + ba_col0 + k]                          (cached adapter), or
//             = vec_dot_f32(r, A[(ba_col0 + k) * r ...], B[(ba_row0 + m) * r ...])   (A: [K_full][r], B: [M_full][r])
template <int TYPE>
__global__ __launch_bounds__(256) void lora_add_aos_kernel(unsigned char *__restrict__ aos, int KB, int row0, int rows, int il_part,
                                                           const float *__restrict__ ba, int64_t ldba,
                                                           const float *__restrict__ A, const float *__restrict__ B, int r,
                                                           int ba_row0, int ba_col0, float sign) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)rows * KB) return;
    const int m = (int)(gid / KB), b = (int)(gid % KB);
    constexpr int BS = TYPE == FL_TYPE_Q4_0 ? 20 : 24;
    // il_part >= 0: the tensor's rows are woven by 16 with a sibling (w1|w3): row m sits at 32 (m / 16) + 16 il_part + m % 16
    const int frow = il_part >= 0 ? ((m >> 4) << 5) + (il_part << 4) + (m & 15) : row0 + m;
    unsigned char *blk = aos + ((int64_t)frow * KB + b) * BS;
    const float d = *reinterpret_cast<const float *>(blk);
    const float mn0 = TYPE == FL_TYPE_Q4_1 ? *reinterpret_cast<const float *>(blk + 4) : 0.f;
    unsigned char *qs = blk + (TYPE == FL_TYPE_Q4_0 ? 4 : 8);
    float w[32];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int lo = qs[j] & 0xF, hi = qs[j] >> 4;
        if (TYPE == FL_TYPE_Q4_0) {                                  // dequantize_row_q4_0: (nib - 8) * d
            w[2 * j] = (((float)(lo - 8)) * (d));
            w[2 * j + 1] = (((float)(hi - 8)) * (d));
        } else {                                                     // dequantize_row_q4_1 (AVX2): fma(nib, d, m)
            w[2 * j] = __fmaf_rn((float)lo, d, mn0);
            w[2 * j + 1] = __fmaf_rn((float)hi, d, mn0);
        }
    }
    const float *brow = B ? B + (int64_t)(ba_row0 + m) * r : nullptr;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        const int k = b * 32 + i;
        float v = ba ? ba[(int64_t)(ba_row0 + m) * ldba + ba_col0 + k] : vec_dot_f32_avx2(r, A + (int64_t)(ba_col0 + k) * r, brow);
        if (sign != 1.0f) v = ((v) * (sign));                    // ggml_scale(BA, -1): exact
        w[i] = ((w[i]) + (v));                                   // ggml_vec_acc_f32
    }
    if (TYPE == FL_TYPE_Q4_0) {                                      // quantize_row_q4_0, AVX2 arithmetic
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(w[i]));
        const float nd = __fdiv_rn(amax, 7.0f);
        const float id = amax != 0.0f ? __fdiv_rn(7.0f, amax) : 0.0f;
        *reinterpret_cast<float *>(blk) = nd;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q0 = (int)rintf(((w[2 * j]) * (id))) + 8, q1 = (int)rintf(((w[2 * j + 1]) * (id))) + 8;
            qs[j] = (unsigned char)((q0 & 0xF) | ((q1 & 0xF) << 4));
        }
    } else {                                                         // quantize_row_q4_1, AVX2 arithmetic
        float mn = w[0], mx = w[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            mn = fminf(mn, w[i]);
            mx = fmaxf(mx, w[i]);
        }
        const float nd = __fdiv_rn(((mx) - (mn)), 15.0f);
        const float id = nd != 0.0f ? __fdiv_rn(1.0f, nd) : 0.0f;
        *reinterpret_cast<float *>(blk) = nd;
        *reinterpret_cast<float *>(blk + 4) = mn;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q0 = (int)rintf(((((w[2 * j]) - (mn))) * (id)));
            const int q1 = (int)rintf(((((w[2 * j + 1]) - (mn))) * (id)));
            qs[j] = (unsigned char)((q0 & 0xF) | ((q1 & 0xF) << 4));
        }
    }
}

hipError_t lora_add_aos(int type, void *aos, int KB, int row0, int rows, int il_part, const float *ba, int64_t ldba, const float *A,
                        const float *B, int r, int ba_row0, int ba_col0, float sign, hipStream_t st) {
    if ((!ba && (!A || !B || r < 1)) || rows < 1 || KB < 1) return hipErrorInvalidValue;
    const int64_t total = (int64_t)rows * KB;
    const dim3 grid((unsigned)((total + 255) / 256));
    unsigned char *p = static_cast<unsigned char *>(aos);
    if (type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(lora_add_aos_kernel<FL_TYPE_Q4_0>, grid, dim3(256), 0, st, p, KB, row0, rows, il_part, ba, ldba, A, B, r,
                           ba_row0, ba_col0, sign);
    else
        hipLaunchKernelGGL(lora_add_aos_kernel<FL_TYPE_Q4_1>, grid, dim3(256), 0, st, p, KB, row0, rows, il_part, ba, ldba, A, B, r,
                           ba_row0, ba_col0, sign);
    return hipGetLastError();
}

// quantize_row_q4_0 / quantize_row_q4_1 and their *_reference twins (the `quantize_row_q` / `quantize_row_q_reference` slots
// of quantize_fns_t, /root/reference/lib/ggml.c:1731-1745), one thread per 32-element block, on the reference's AoS blocks.
//   reference flavour (:630-664, :917-956):  id = d ? 1/d : 0, roundf (halves away from zero)
//   SIMD flavour as built for AVX2 (:757-803, :965-1037): Q4_0 id = 7/amax; both round halves to even
template <int TYPE, bool SIMD>
__global__ __launch_bounds__(256) void quantize_row_q4_kernel(const float *__restrict__ x, unsigned char *__restrict__ y, int nb) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    constexpr int BS = TYPE == FL_TYPE_Q4_0 ? 20 : 24;
    const float *xb = x + (int64_t)b * 32;
    unsigned char *blk = y + (int64_t)b * BS;
    unsigned char *qs = blk + (TYPE == FL_TYPE_Q4_0 ? 4 : 8);
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = xb[i];
    if (TYPE == FL_TYPE_Q4_0) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(w[i]));
        const float d = __fdiv_rn(amax, 7.0f);
        const float id = SIMD ? (amax != 0.0f ? __fdiv_rn(7.0f, amax) : 0.0f) : (d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f);
        *reinterpret_cast<float *>(blk) = d;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v0 = ((w[2 * j]) * (id)), v1 = ((w[2 * j + 1]) * (id));
            const int q0 = (int)(SIMD ? rintf(v0) : roundf(v0)) + 8, q1 = (int)(SIMD ? rintf(v1) : roundf(v1)) + 8;
            qs[j] = (unsigned char)((q0 & 0xF) | ((q1 & 0xF) << 4));
        }
    } else {
        float mn = w[0], mx = w[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            mn = fminf(mn, w[i]);
            mx = fmaxf(mx, w[i]);
        }
        const float d = __fdiv_rn(((mx) - (mn)), 15.0f);
        const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
        *reinterpret_cast<float *>(blk) = d;
        *reinterpret_cast<float *>(blk + 4) = mn;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v0 = ((((w[2 * j]) - (mn))) * (id)), v1 = ((((w[2 * j + 1]) - (mn))) * (id));
            const int q0 = (int)(SIMD ? rintf(v0) : roundf(v0)), q1 = (int)(SIMD ? rintf(v1) : roundf(v1));
            qs[j] = (unsigned char)((q0 & 0xF) | ((q1 & 0xF) << 4));
        }
    }
}

hipError_t quantize_row_q4_aos(int type, bool reference, const float *x, void *y, int64_t k, hipStream_t st) {
    if (k <= 0 || k % 32 != 0) return hipErrorInvalidValue;
    const int nb = (int)(k / 32);
    const dim3 grid((unsigned)((nb + 255) / 256));
    unsigned char *p = static_cast<unsigned char *>(y);
    if (type == FL_TYPE_Q4_0 && reference) hipLaunchKernelGGL((quantize_row_q4_kernel<FL_TYPE_Q4_0, false>), grid, dim3(256), 0, st, x, p, nb);
    else if (type == FL_TYPE_Q4_0) hipLaunchKernelGGL((quantize_row_q4_kernel<FL_TYPE_Q4_0, true>), grid, dim3(256), 0, st, x, p, nb);
    else if (reference) hipLaunchKernelGGL((quantize_row_q4_kernel<FL_TYPE_Q4_1, false>), grid, dim3(256), 0, st, x, p, nb);
    else hipLaunchKernelGGL((quantize_row_q4_kernel<FL_TYPE_Q4_1, true>), grid, dim3(256), 0, st, x, p, nb);
    return hipGetLastError();
}

}  // namespace fl
