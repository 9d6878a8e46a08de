// eval_kernels.hip -- the non-matmul ops of Model::eval (/root/reference/lib/llama.cpp:301-465) as gfx950
// kernels, so that an eval is device-resident end to end.  Each kernel fuses the
// reference ops that sit between two quantized matmuls and reproduces their numerics:
//
//   rmsnorm_quant   rms_norm (f64 sum of squares, eps 1e-6)    lib/ggml.c:7378-7430
//                   * norm weight (ggml_mul)                    lib/llama.cpp:312-318
//                   -> quantize_row_q8_0 of the result          lib/ggml.c:1299 (the INIT phase of the next mul_mat)
//   rope_kv         rope mode 0 on Q and K                      lib/ggml.c:8609-8682 (fma pattern of the gcc build)
//                   K -> k cache, V -> transposed v cache       lib/llama.cpp:336-347
//   gemm_f32_abt    KQ = K*Q, KQV = V*softmax (f32 mul_mat)     lib/ggml.c:7482, lib/llama.cpp:364,389
//                   (+ ggml_scale by 1/sqrt(head_dim))          lib/llama.cpp:367-371
//   softmax_rows    diag_mask_inf + soft_max with the fp16 exp TABLE and an f64 sum   lib/ggml.c:8466,8521-8580
//   silu_mul_quant  silu through the fp16 TABLE, * w3 branch, -> Q8_0                  lib/ggml.c:3207-3215
//
// The fp16 tables (exp, silu) and the rope sin/cos table are computed on the HOST with the same libm the
// reference uses (ggml_init, lib/ggml.c:3676-3693) and uploaded once, so table entries are bit-identical.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <climits>
#include "eval_kernels.h"
#include "comm.h"
#include "tp_tail.h"
#include "q4_device.h"

namespace fl {

__device__ __forceinline__ double block_sum_f64(double v, double *sh) {
    v = wave_sum_f64(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------------------
// rms_norm * weight  (-> optional f32 copy)  -> Q8_0 in QA16 / QA1
//   one workgroup per activation row; rows >= N (padding of QA16) produce all-zero blocks.
// ------------------------------------------------------------------------------------------------
constexpr int RN_MAXIT = 4;  // 256 threads * 8 elements * 4 = E <= 8192

// gath (row-split tensor parallelism, prefill): the row is not in x yet -- it is the all-gathered output rows of the wo / w2 matmul, tmp[rank][n][El],
// plus the residual (the ggml_add of lib/llama.cpp:407, :441: one plain f32 add per element, as gather_rows_add_kernel did in a launch of its own);
// the sum is written to x (the next residual) on the way.
struct RmsGather { const float *tmp; const float *resid; int ldr, El, rows; };      // rows = N of the all-gather (the stride between ranks is rows * El)
__global__ __launch_bounds__(256) void rmsnorm_quant_kernel(float *__restrict__ x, int ldx,
                                                            const float *__restrict__ w, int N, int E,
                                                            float *__restrict__ y_f32, int ldy, int layout,
                                                            int8_t *__restrict__ q, float *__restrict__ d,
                                                            float *__restrict__ s, uint16_t *__restrict__ h16, const RmsGather gath) {
    __shared__ double sh[4];
    const int n = blockIdx.x;
    const int gpr = E >> 3, KB = E >> 5;
    float v[RN_MAXIT][8], ww[RN_MAXIT][8];
    double sum = 0.0;
    // (the norm weights are requested with the row, not after the reduction: one L2 round trip less on the critical path)
#pragma unroll
    for (int it = 0; it < RN_MAXIT; ++it) {
        const int kg = threadIdx.x + it * 256;
        if (kg < gpr) {
            const float4 wa = *reinterpret_cast<const float4 *>(w + kg * 8);
            const float4 wc = *reinterpret_cast<const float4 *>(w + kg * 8 + 4);
            ww[it][0] = wa.x; ww[it][1] = wa.y; ww[it][2] = wa.z; ww[it][3] = wa.w;
            ww[it][4] = wc.x; ww[it][5] = wc.y; ww[it][6] = wc.z; ww[it][7] = wc.w;
        }
    }
#pragma unroll
    for (int it = 0; it < RN_MAXIT; ++it) {
        const int kg = threadIdx.x + it * 256;
        if (kg < gpr && n < N && gath.tmp) {
            const int e = kg * 8, r = e / gath.El, j = e - r * gath.El;      // (El % 8 == 0: the eight elements are one rank's)
            const float *tp = gath.tmp + ((int64_t)r * gath.rows + n) * gath.El + j, *rp = gath.resid + (int64_t)n * gath.ldr + e;
            const float4 a = *reinterpret_cast<const float4 *>(tp), c = *reinterpret_cast<const float4 *>(tp + 4);
            const float4 ra = *reinterpret_cast<const float4 *>(rp), rc = *reinterpret_cast<const float4 *>(rp + 4);
            v[it][0] = __fadd_rn(a.x, ra.x); v[it][1] = __fadd_rn(a.y, ra.y); v[it][2] = __fadd_rn(a.z, ra.z); v[it][3] = __fadd_rn(a.w, ra.w);
            v[it][4] = __fadd_rn(c.x, rc.x); v[it][5] = __fadd_rn(c.y, rc.y); v[it][6] = __fadd_rn(c.z, rc.z); v[it][7] = __fadd_rn(c.w, rc.w);
            float4 *xp = reinterpret_cast<float4 *>(x + (int64_t)n * ldx + e);
            xp[0] = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
            xp[1] = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
        } else if (kg < gpr && n < N) {
            const float4 a = *reinterpret_cast<const float4 *>(x + (int64_t)n * ldx + kg * 8);
            const float4 c = *reinterpret_cast<const float4 *>(x + (int64_t)n * ldx + kg * 8 + 4);
            v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
            v[it][4] = c.x; v[it][5] = c.y; v[it][6] = c.z; v[it][7] = c.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[it][i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += (double)__fmul_rn(v[it][i], v[it][i]);  // (ggml_float)(x*x)
    }
    sum = block_sum_f64(sum, sh);
    const float mean = (float)(sum / (double)E);
    const float scale = __fdiv_rn(1.0f, sqrtf(mean + 1e-6f));
#pragma unroll
    for (int it = 0; it < RN_MAXIT; ++it) {
        const int kg = threadIdx.x + it * 256;
        if (kg >= gpr) continue;   // whole quads leave together (gpr is a multiple of 4)
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = __fmul_rn(ww[it][i], __fmul_rn(v[it][i], scale));  // w * (x*scale)
        if (y_f32 && n < N) {
            float4 *yp = reinterpret_cast<float4 *>(y_f32 + (int64_t)n * ldy + kg * 8);
            yp[0] = make_float4(o[0], o[1], o[2], o[3]);
            yp[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (q) quantize_store_group(o, n, kg, KB, layout, q, d, s, h16);
    }
}

hipError_t rmsnorm_quant(const float *x, int ldx, const float *w, int N, int E, float *y_f32, int ldy,
                         const fl_qact *out, int layout, hipStream_t st, bool with_h16) {
    if (E % 32 != 0 || E > 256 * 8 * RN_MAXIT) return hipErrorInvalidValue;
    const int rows = (out && layout == 16) ? fl_roundup(N, 16) : N;
    hipLaunchKernelGGL(rmsnorm_quant_kernel, dim3(rows), dim3(256), 0, st, const_cast<float *>(x), ldx, w, N, E, y_f32, ldy, layout,
                       out ? out->q : nullptr, out ? out->d : nullptr, out ? out->s : nullptr,
                       out && with_h16 && layout == 16 ? out->h16 : nullptr, RmsGather{nullptr, nullptr, 0, 0, 0});
    return hipGetLastError();
}
// x[n][:] = gathered[rank][n][El] + resid[n][:] (written), then rms_norm * w -> Q8_0 as above: one launch instead of gather_rows_add + rmsnorm_quant
hipError_t rmsnorm_quant_gathered(const float *gathered, int G, int El, const float *resid, int ldr, float *x, int ldx, const float *w, int N, int E,
                                  float *y_f32, int ldy, const fl_qact *out, int layout, hipStream_t st, bool with_h16) {
    if (E % 32 != 0 || E > 256 * 8 * RN_MAXIT || G * El != E || El % 8 != 0 || (ldr & 3) || (ldx & 3)) return hipErrorInvalidValue;
    const int rows = (out && layout == 16) ? fl_roundup(N, 16) : N;
    hipLaunchKernelGGL(rmsnorm_quant_kernel, dim3(rows), dim3(256), 0, st, x, ldx, w, N, E, y_f32, ldy, layout,
                       out ? out->q : nullptr, out ? out->d : nullptr, out ? out->s : nullptr,
                       out && with_h16 && layout == 16 ? out->h16 : nullptr, RmsGather{gathered, resid, ldr, El, N});
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// silu(w1 x) * (w3 x) -> Q8_0.   h13: N rows of [w1 x (F) | w3 x (F)]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_quant_kernel(const float *__restrict__ h13, int ld, int N, int NP,
                                                             int F, const uint16_t *__restrict__ silu_tab,
                                                             int layout, int8_t *__restrict__ q,
                                                             float *__restrict__ d, float *__restrict__ s, int woven,
                                                             uint16_t *__restrict__ h16) {
    const int gpr = F >> 3, KB = F >> 5;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)NP * gpr;
    const bool live = gid < total;
    const int n = live ? (int)(gid / gpr) : 0, kg = live ? (int)(gid % gpr) : 0;
    float o[8];
    if (live && n < N) {
        // features 8kg .. 8kg+7 of w1 x, and the same features of w3 x (16-row groups woven: +16, else +F)
        const float *pa = h13 + (int64_t)n * ld + (woven ? ((kg >> 1) << 5) + ((kg & 1) << 3) : kg * 8);
        const int boff = woven ? 16 : F;
        const float4 a0 = *reinterpret_cast<const float4 *>(pa), a1 = *reinterpret_cast<const float4 *>(pa + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(pa + boff), b1 = *reinterpret_cast<const float4 *>(pa + boff + 4);
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint16_t hx = __half_as_ushort(__float2half_rn(a[i]));                 // GGML_FP32_TO_FP16
            const float sl = __half2float(__ushort_as_half(silu_tab[hx]));              // table_silu_f16
            o[i] = __fmul_rn(sl, b[i]);                                                 // ggml_mul(silu, tmp)
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
    }
    // all 4 lanes of a quad are live or dead together (gpr % 4 == 0); shuffles inside need full quads
    if (live) quantize_store_group(o, n, kg, KB, layout, q, d, s, h16);
}

hipError_t silu_mul_quant(const float *h13, int ld, int N, int F, const uint16_t *silu_tab, const fl_qact *out,
                          int layout, hipStream_t st, bool woven, bool with_h16) {
    const int NP = layout == 16 ? fl_roundup(N, 16) : N;
    const int64_t total = (int64_t)NP * (F >> 3);
    hipLaunchKernelGGL(silu_mul_quant_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, h13, ld, N, NP, F,
                       silu_tab, layout, out->q, out->d, out->s, woven ? 1 : 0, with_h16 && layout == 16 ? out->h16 : nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// rope (mode 0, interleaved pairs) on Q (in place) and K (-> k cache), V -> transposed v cache
//   qkv : N rows of [q (E) | k (E) | v (E)]        rope_tab : [n_ctx][D/2] float2 {cos, sin}
//   kc  : [n_ctx][E]   (row p = position)           vc : [E][n_ctx]
// The reference's gcc build computes  d0 = fma(x0, cos, -(x1*sin)),  d1 = fma(x0, sin, x1*cos).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kv_kernel(float *__restrict__ qkv, int ld, int N, int E, int D,
                                                      int n_past, int n_ctx, const float2 *__restrict__ rope_tab,
                                                      float *__restrict__ kc, float *__restrict__ vc,
                                                      const int *__restrict__ dyn_past) {
    if (dyn_past) n_past = *dyn_past;   // decode graph: the position lives in device memory
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (n, pair)
    const int ppr = E >> 1;
    if (gid >= (int64_t)N * ppr) return;
    const int n = (int)(gid / ppr), pr = (int)(gid % ppr);
    const int e0 = pr * 2, i = (e0 % D) >> 1, pos = n_past + n;
    const float2 cs = rope_tab[(int64_t)pos * (D >> 1) + i];
    float *row = qkv + (int64_t)n * ld;
    {
        const float2 x = *reinterpret_cast<const float2 *>(row + e0);
        float2 o;
        o.x = __fmaf_rn(x.x, cs.x, -__fmul_rn(x.y, cs.y));
        o.y = __fmaf_rn(x.x, cs.y, __fmul_rn(x.y, cs.x));
        *reinterpret_cast<float2 *>(row + e0) = o;
    }
    {
        const float2 x = *reinterpret_cast<const float2 *>(row + E + e0);
        float2 o;
        o.x = __fmaf_rn(x.x, cs.x, -__fmul_rn(x.y, cs.y));
        o.y = __fmaf_rn(x.x, cs.y, __fmul_rn(x.y, cs.x));
        *reinterpret_cast<float2 *>(kc + (int64_t)pos * E + e0) = o;
    }
    if (N < 16) {  // decode-sized: scatter the two V elements directly
        const float2 v = *reinterpret_cast<const float2 *>(row + 2 * E + e0);
        vc[(int64_t)e0 * n_ctx + pos] = v.x;
        vc[(int64_t)(e0 + 1) * n_ctx + pos] = v.y;
    }
}

// prefill: V [N][E] (inside qkv) -> vc[E][n_ctx] at columns n_past.. via a 32x32 LDS transpose
__global__ __launch_bounds__(256) void v_transpose_kernel(const float *__restrict__ qkv, int ld, int N, int E,
                                                          int n_past, int n_ctx, float *__restrict__ vc) {
    __shared__ float tile[32][33];
    const int e0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + ty + r * 8;
        tile[ty + r * 8][tx] = n < N ? qkv[(int64_t)n * ld + 2 * E + e0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = e0 + ty + r * 8, n = n0 + tx;
        if (n < N) vc[(int64_t)e * n_ctx + n_past + n] = tile[tx][ty + r * 8];
    }
}

hipError_t rope_kv(float *qkv, int ld, int N, int E, int D, int n_past, int n_ctx, const float *rope_tab, float *kc,
                   float *vc, hipStream_t st, const int *dyn_past) {
    const int64_t total = (int64_t)N * (E >> 1);
    hipLaunchKernelGGL(rope_kv_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, qkv, ld, N, E, D, n_past,
                       n_ctx, reinterpret_cast<const float2 *>(rope_tab), kc, vc, dyn_past);
    if (N >= 16)
        hipLaunchKernelGGL(v_transpose_kernel, dim3(E / 32, (N + 31) / 32), dim3(256), 0, st, qkv, ld, N, E, n_past,
                           n_ctx, vc);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// batched f32 GEMM  C[z][m][n] = alpha * sum_k A[z][m][k] * B[z][n][k]   (both operands k-contiguous)
// on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, same numerics as a scalar loop).
//   KQ  : A = Q rows, B = k-cache rows, K = head_dim, alpha = 1/sqrt(head_dim) applied after the dot
//   KQV : A = softmax rows, B = transposed v-cache rows, K = n_past + N
// causal != 0: row m only needs n <= n_past + m (scores) resp. k <= n_past + m (KQV): tiles that are
// entirely masked are skipped (the softmax kernel never reads them, and wrote zeros where KQV reads).
// One wave per 32x32 output tile, 4 tiles (2x2) per 256-thread workgroup; operands come straight from
// L1/L2 (Q/K/V tiles of one head are a few hundred KiB).
// ------------------------------------------------------------------------------------------------
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void gemm_f32_abt_kernel(const float *__restrict__ A, int lda, int64_t sAz,
                                                           const float *__restrict__ B, int ldb, int64_t sBz,
                                                           float *__restrict__ C, int ldc, int64_t sCz, int M, int Nn,
                                                           int K, float alpha, int causal_mode, int n_past,
                                                           const int *__restrict__ dyn_past) {
    if (dyn_past) {                      // decode graph: P = n_past + M is read from device memory
        n_past = *dyn_past;
        if (causal_mode == 1) Nn = n_past + M;
        if (causal_mode == 2) K = n_past + M;
    }
    const int z = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    if (m0 >= M || n0 >= Nn) return;
    int kend = K;
    if (causal_mode == 1 && n0 > n_past + m0 + 31) return;             // scores: whole tile is masked
    if (causal_mode == 2) kend = min(K, n_past + m0 + 32);            // KQV: probabilities beyond are zero
    const int r = lane & 31, kk = lane >> 5;
    const float *pa = A + z * sAz + (int64_t)min(m0 + r, M - 1) * lda;
    const float *pb = B + z * sBz + (int64_t)min(n0 + r, Nn - 1) * ldb;
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // main loop, 8 k per step: lane (r, kk) loads ONE float4 = k {8j+4kk .. 8j+4kk+3}; MFMA step s pairs
    // k = 8j+s (lanes kk=0) with k = 8j+4+s (lanes kk=1).  A and B use the same pairing, so the sum over k is
    // complete; only the f32 order of the additions differs from a scalar loop.
    int k = 0;
    for (; k + 16 <= kend; k += 16) {
        const float4 a0 = *reinterpret_cast<const float4 *>(pa + k + 4 * kk);
        const float4 b0 = *reinterpret_cast<const float4 *>(pb + k + 4 * kk);
        const float4 a1 = *reinterpret_cast<const float4 *>(pa + k + 8 + 4 * kk);
        const float4 b1 = *reinterpret_cast<const float4 *>(pb + k + 8 + 4 * kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
    }
    for (; k + 8 <= kend; k += 8) {
        const float4 a0 = *reinterpret_cast<const float4 *>(pa + k + 4 * kk);
        const float4 b0 = *reinterpret_cast<const float4 *>(pb + k + 4 * kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    }
    for (; k < kend; k += 2) {   // tail: plain (k, k+1) pairing
        const float a = (k + kk < kend) ? pa[k + kk] : 0.f;
        const float b = (k + kk < kend) ? pb[k + kk] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *pc = C + z * sCz;
    const int col = n0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * kk;
        if (row < M && col < Nn) pc[(int64_t)row * ldc + col] = __fmul_rn(acc[i], alpha);
    }
}

hipError_t gemm_f32_abt(const float *A, int lda, int64_t sAz, const float *B, int ldb, int64_t sBz, float *C, int ldc,
                        int64_t sCz, int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past,
                        hipStream_t st, const int *dyn_past, int nn_max) {
    // with dyn_past the grid is sized for the largest possible Nn (scores: n_ctx columns); surplus tiles exit at once
    const dim3 grid(((dyn_past && causal_mode == 1 ? nn_max : Nn) + 63) / 64, (M + 63) / 64, batch);
    hipLaunchKernelGGL(gemm_f32_abt_kernel, grid, dim3(256), 0, st, A, lda, sAz, B, ldb, sBz, C, ldc, sCz, M, Nn, K, alpha,
                       causal_mode, n_past, dyn_past);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// prefill attention: KQ*scale -> diag_mask_inf + soft_max -> KQV with the score rows kept in LDS
// (lib/llama.cpp:364-398 without the [heads][N][n_ctx] round trip through HBM).
//
// One 512-thread workgroup per (head, PAIR of 32-row query blocks {x, nb-1-x}): causal attention gives block i
// (i+1) key tiles, so pairing the lightest with the heaviest makes every workgroup equally heavy -- with one
// workgroup per CU resident from the start there is no second scheduling round to even things out.
//   phase 0  the useful part of the fp16 exp table (arguments in [-17.4, -0]; everything below is 0) and the 64 Q rows -> LDS
//   phase 1  the workgroup's score tiles (block a's, then block b's) are dealt round-robin to the 8 waves (exact-f32
//            MFMA, the k pairing and order of gemm_f32_abt_kernel; A fragments from LDS, next K tile prefetched)
//   phase 2  soft_max of the 64 rows (the arithmetic of softmax_rows_kernel; table lookups from LDS, four rows'
//            lookups in flight together)
//   phase 3  wave w: block w/4, output columns [32(w%4), +32) of the head: A = probabilities (LDS), B = transposed V
// Per output the MFMA sequence is the one of gemm_f32_abt(KQ) -> softmax_rows -> gemm_f32_abt(KQV): bit-identical,
// and independent of how rows are grouped into blocks.
// ------------------------------------------------------------------------------------------------
// soft_max of FOUR score rows held in LDS by one wave: IT 64-column steps cover the longest of them.  Straight-line
// code (clamped addresses + selects, no branches) so that the LDS round trips of all columns overlap.  The f64 sum of
// fp16-valued terms is exact, so its order is free; everything else is the arithmetic of softmax_rows_kernel.
template <int IT>
__device__ __forceinline__ void pa_softmax4(float *const (&rowp)[4], const int (&Ls)[4], int ncol, const uint16_t *tab,
                                            int tab_n, int lane) {
    float x[4][IT];
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4)
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int i = lane + 64 * u;
            const float v = rowp[u4][min(i, ncol - 1)];                 // always a valid LDS address of this row
            x[u4][u] = i < Ls[u4] ? v : -INFINITY;
        }
    float mx[4];
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < IT; ++u) m = fmaxf(m, x[u4][u]);
        mx[u4] = wave_max_f32(m);
    }
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
        // table index from mx - x >= 0 (rounding is sign-symmetric: half(mx - x) is half(x - mx) without the sign bit):
        // +0 -> entry 0 = exp(-0) = 1; [0, tab_n): the table; up to +inf (0x7C00; masked columns, x = -inf): exp underflows
        // to 0 in fp16 (host-checked bound); NaNs and (impossible) negative differences: NaN like the full table.
        // The empty asm keeps the LDS reads unconditional (else: one branch per element around its lookup).
        double sum = 0.0;
        uint32_t idx[IT], t[IT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            idx[u] = __half_as_ushort(__float2half_rn(mx[u4] - x[u4][u]));
            t[u] = tab[min(idx[u], (uint32_t)(tab_n - 1))];
        }
#pragma unroll
        for (int u = 0; u < IT; ++u) asm volatile("" : "+v"(t[u]));
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const uint32_t e = idx[u] < (uint32_t)tab_n ? t[u] : idx[u] <= 0x7C00u ? 0u : 0x7E00u;
            const float v = __half2float(__ushort_as_half((uint16_t)e));
            x[u4][u] = v;
            sum += (double)v;
        }
        sum = wave_sum_f64(sum);
        const float inv = (float)(1.0 / sum);
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int i = lane + 64 * u;
            if (i < ncol) rowp[u4][i] = i < Ls[u4] ? __fmul_rn(x[u4][u], inv) : 0.f;
        }
    }
}

#ifdef PA_TIMING   // development build only: per-phase clocks of a few workgroups (scripts/attn_only.py)
__device__ long long pa_dbg[64 * 8];
#define PA_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y < 8) pa_dbg[blockIdx.y * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PA_STAMP(k) do {} while (0)
#endif
constexpr int PA_T = 512, PA_SM_IT = 15;                              // threads; 64-column steps of a row (P <= 960)

template <int NSTEP>   // head_dim / 8
__global__ __launch_bounds__(PA_T) void prefill_attention_kernel(const float *__restrict__ qkv, int ldq, int N, int n_past,
                                                                 int n_ctx, int E, const float *__restrict__ kc,
                                                                 const float *__restrict__ vc,
                                                                 const uint16_t *__restrict__ exp_tab, int tab_n,
                                                                 float scale, float *__restrict__ ao, int ldo, int pair,
                                                                 int8_t *__restrict__ oq, float *__restrict__ od,
                                                                 float *__restrict__ os) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    constexpr int D = NSTEP * 8;
    // grid = (heads, block pairs): consecutive workgroup ids go to consecutive XCDs, so all workgroups of a head share
    // one XCD's L2 (its K and V, 2 x n_ctx x 512 B, are read by every one of them)
    const int h = blockIdx.x, bx = blockIdx.y;
    const int nb = (N + 31) >> 5;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kk = lane >> 5;
    const int P = n_past + N;
    // the two blocks of this workgroup: mb0 (light), mb1 (heavy; -1 if none).  Scalars + selects, not arrays: a
    // runtime-indexed array would live in scratch.
    const int mb0 = bx;
    int mb1 = pair ? nb - 1 - bx : -1;
    if (pair && mb1 <= mb0) mb1 = -1;                                   // middle block of an odd count: alone
    const int kend0 = min(P, n_past + mb0 * 32 + 32);                  // keys any row of the block can see
    const int kend1 = mb1 >= 0 ? min(P, n_past + mb1 * 32 + 32) : 0;
    const int ntl0 = (kend0 + 31) >> 5, ntl1 = (kend1 + 31) >> 5;
    const int strd0 = ntl0 * 32 + 4, strd1 = ntl1 * 32 + 4;           // + 4 floats: rows land on different LDS banks
    float *const S0 = reinterpret_cast<float *>(psm);
    float *const S1 = S0 + 32 * strd0;
    float *base = S1 + 32 * strd1;
#define PA_SEL(q, a, b) ((q) ? (b) : (a))
    uint16_t *tab = reinterpret_cast<uint16_t *>(base);                  // exp table for fp16 arguments 0x8000 + [0, tab_n)

    PA_STAMP(0);
    // ---- phase 0: the roped Q rows of both blocks -> LDS (A operand of every score tile, read back per MFMA step), the
    //      exp table -> LDS, and this wave's first K tile -> registers: one memory round trip for all three ----
    constexpr int QLD = D + 4;                                        // + 4 floats: conflict-free ds_read_b128 of 32 rows
    float *const Qs = reinterpret_cast<float *>(tab + ((tab_n + 7) & ~7));   // [2][32][QLD]
    const int nunits = ntl0 + ntl1;                                  // score tiles of the workgroup: block a's, then block b's
    float4 kf[NSTEP], kn[NSTEP];
    auto load_k = [&](float4 (&dst)[NSTEP], int u) {                 // K tile of unit u
        const int kt = u < ntl0 ? u : u - ntl0;
        const float *pb = kc + (int64_t)min(kt * 32 + r, P - 1) * E + h * D;
#pragma unroll
        for (int j = 0; j < NSTEP; ++j) dst[j] = *reinterpret_cast<const float4 *>(pb + 8 * j + 4 * kk);
    };
#ifndef PA_ABL
#define PA_ABL 0
#endif
    if (wave < nunits && !(PA_ABL & 1)) load_k(kf, wave);
    for (int i = threadIdx.x; i < 2 * 32 * (D / 4); i += PA_T) {
        const int q = i / (32 * (D / 4)), row = (i / (D / 4)) & 31, c4 = i % (D / 4);
        const int mbq = PA_SEL(q, mb0, mb1);
        if (mbq < 0) continue;
        const float4 v = *reinterpret_cast<const float4 *>(qkv + (int64_t)min(mbq * 32 + row, N - 1) * ldq + h * D + c4 * 4);
        *reinterpret_cast<float4 *>(Qs + (q * 32 + row) * QLD + c4 * 4) = v;
    }
    for (int i = threadIdx.x; i * 8 < tab_n; i += PA_T)
        reinterpret_cast<uint4 *>(tab)[i] = reinterpret_cast<const uint4 *>(exp_tab + 0x8000)[i];
    __syncthreads();
    PA_STAMP(1);
    // ---- phase 1: scores ----   (PA_ABL: timing experiments only, never defined in the product)
    if (!(PA_ABL & 1)) {
        for (int u = wave; u < nunits; u += PA_T / 64) {
            if (u + PA_T / 64 < nunits) load_k(kn, u + PA_T / 64);    // next tile's K under this tile's MFMAs
            __builtin_amdgcn_sched_barrier(0);                        // (... and not, as the scheduler would have it, after them)
            const int q = u < ntl0 ? 0 : 1, kt = u < ntl0 ? u : u - ntl0;
            const float *qa = Qs + (q * 32 + r) * QLD + 4 * kk;
            v16f acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int j = 0; j < NSTEP; ++j) {                         // 8 k per step, lane half kk takes k+4kk..
                const float4 qf = *reinterpret_cast<const float4 *>(qa + 8 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.x, kf[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.y, kf[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.z, kf[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.w, kf[j].w, acc, 0, 0, 0);
            }
            float *const Sq = PA_SEL(q, S0, S1);
            const int strdq = PA_SEL(q, strd0, strd1), col = kt * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * kk;
                Sq[row * strdq + col] = __fmul_rn(acc[i], scale);
            }
#pragma unroll
            for (int j = 0; j < NSTEP; ++j) kf[j] = kn[j];
        }
    }
    PA_STAMP(2);
    __syncthreads();
    PA_STAMP(3);
    // ---- phase 2: soft_max; wave w owns rows w, w+8, ... of the 64.  Four rows at a time with every table lookup of
    //      the four rows in flight together; per lane the terms are added in increasing column order: the f64 sums of
    //      softmax_rows_kernel.
    for (int rb = 0; rb < ((PA_ABL & 2) ? 0 : 8); rb += 4) {          // rows wave + 8*(rb..rb+3): one block per batch
        const int q = rb >> 2;
        const int mbq = PA_SEL(q, mb0, mb1), ncol = PA_SEL(q, ntl0, ntl1) * 32;
        if (mbq < 0) continue;
        float *rowp[4];
        int Ls[4];
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const int row = wave + 8 * u4;                              // 0..31 inside the block
            rowp[u4] = PA_SEL(q, S0, S1) + row * PA_SEL(q, strd0, strd1);
            Ls[u4] = min(P, n_past + mbq * 32 + row + 1);              // rows >= N: clamped duplicates, never stored
        }
        const int nit = (ncol + 63) >> 6;                              // uniform: 64-column steps that hold any data
        if (nit <= 2) pa_softmax4<2>(rowp, Ls, ncol, tab, tab_n, lane);
        else if (nit <= 4) pa_softmax4<4>(rowp, Ls, ncol, tab, tab_n, lane);
        else if (nit <= 8) pa_softmax4<8>(rowp, Ls, ncol, tab, tab_n, lane);
        else pa_softmax4<PA_SM_IT>(rowp, Ls, ncol, tab, tab_n, lane);
    }
    PA_STAMP(4);
    __syncthreads();
    PA_STAMP(5);
    // ---- phase 3: KQV ----
    {
        const int q = wave >> 2, d0 = (wave & 3) * 32;
        const int mbq = PA_SEL(q, mb0, mb1);
        if (mbq >= 0 && d0 < D && !(PA_ABL & 4)) {
            const int m0 = mbq * 32, ke = PA_SEL(q, kend0, kend1);
            const float *pa = PA_SEL(q, S0, S1) + r * PA_SEL(q, strd0, strd1);   // probabilities of query row r
            const float *pb = vc + (int64_t)(h * D + d0 + r) * n_ctx;     // transposed V cache row (d0 + r)
            v16f acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            int k = 0;
#define FL_PV4(a, b)                                                      \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            // V (L2) is requested TWO 32-key steps ahead of its MFMAs: three register sets in rotation
            float4 vb[3][4];
            auto load_v = [&](float4 (&d)[4], int kb) __attribute__((always_inline)) {   // 32 k: four float4 per lane
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = *reinterpret_cast<const float4 *>(pb + kb + 8 * j + 4 * kk);
            };
            auto mm32 = [&](const float4 (&bv)[4], int kb) __attribute__((always_inline)) {
                const float4 a0 = *reinterpret_cast<const float4 *>(pa + kb + 4 * kk);
                const float4 a1 = *reinterpret_cast<const float4 *>(pa + kb + 8 + 4 * kk);
                const float4 a2 = *reinterpret_cast<const float4 *>(pa + kb + 16 + 4 * kk);
                const float4 a3 = *reinterpret_cast<const float4 *>(pa + kb + 24 + 4 * kk);
                FL_PV4(a0, bv[0]) FL_PV4(a1, bv[1]) FL_PV4(a2, bv[2]) FL_PV4(a3, bv[3])
            };
            if (32 <= ke) load_v(vb[0], 0);
            if (64 <= ke) load_v(vb[1], 32);
            for (; k + 96 <= ke; k += 96) {
                load_v(vb[2], k + 64);
                mm32(vb[0], k);
                if (k + 128 <= ke) load_v(vb[0], k + 96);
                mm32(vb[1], k + 32);
                if (k + 160 <= ke) load_v(vb[1], k + 128);
                mm32(vb[2], k + 64);
            }
            if (k + 32 <= ke) {
                mm32(vb[0], k);
                k += 32;
                if (k + 32 <= ke) {
                    mm32(vb[1], k);
                    k += 32;
                }
            }
            for (; k + 8 <= ke; k += 8) {
                const float4 a0 = *reinterpret_cast<const float4 *>(pa + k + 4 * kk);
                const float4 c0 = *reinterpret_cast<const float4 *>(pb + k + 4 * kk);
                FL_PV4(a0, c0)
            }
#undef FL_PV4
            for (; k < ke; k += 2) {   // tail: plain (k, k+1) pairing
                const float a = (k + kk < ke) ? pa[k + kk] : 0.f;
                const float b = (k + kk < ke) ? pb[k + kk] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (!oq) {
                const int col = h * D + d0 + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * kk;
                    if (row < N) ao[(int64_t)row * ldo + col] = acc[i];
                }
            } else {
                // the 32x32 tile goes to the (now free) Q area: [block][column tile][32 rows][33]
                float *T = Qs + (q * (D / 32) + (wave & 3)) * (32 * 33);
#pragma unroll
                for (int i = 0; i < 16; ++i) T[((i & 3) + 8 * (i >> 2) + 4 * kk) * 33 + r] = acc[i];
            }
        }
    }
    if (oq) {
        // ---- phase 4: quantize_row_q8_0 of the result (lib/ggml.c:1341-1403,1433-1440), straight into the QA16 operand of
        //      the wo matmul: one thread = one (token, 32 output columns = one quant block) ----
        __syncthreads();
        if (threadIdx.x < 256) {
            const int w = threadIdx.x >> 5, rr = threadIdx.x & 31;
            const int q = w >> 2, d0 = (w & 3) * 32, mbq = PA_SEL(q, mb0, mb1);
            const int n = mbq * 32 + rr, KBo = E >> 5, N16 = (N + 15) & ~15;
            if (mbq >= 0 && d0 < D && n < N16) {
                const float *T = Qs + (q * (D / 32) + (w & 3)) * (32 * 33) + rr * 33;
                float v[32], amax = 0.f;
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    v[e] = n < N ? T[e] : 0.f;                         // columns N..N16-1 of the operand are zero blocks
                    amax = fmaxf(amax, fabsf(v[e]));
                }
                const float dd = __fdiv_rn(amax, 127.0f);
                const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
                int qi[32], sum = 0;
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    qi[e] = (int)rintf(__fmul_rn(v[e], id));
                    sum += qi[e];
                }
                const int c = n & 15;
                const int64_t cb = ((int64_t)(n >> 4) * KBo + ((h * D + d0) >> 5)) * 16 + c;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    auto pk = [](int a, int b, int cc, int d) -> uint32_t {
                        return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(cc & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
                    };
                    *reinterpret_cast<uint2 *>(oq + cb * 32 + qw16_pos(c, g) * 8) =
                        make_uint2(pk(qi[8 * g], qi[8 * g + 2], qi[8 * g + 4], qi[8 * g + 6]), pk(qi[8 * g + 1], qi[8 * g + 3], qi[8 * g + 5], qi[8 * g + 7]));
                }
                od[cb] = dd;
                os[cb] = __fmul_rn(dd, (float)sum);
            }
        }
    }
    PA_STAMP(6);
}
#ifdef PA_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_pa_timing(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pa_dbg), sizeof(long long) * 64 * 8); }
extern "C" __attribute__((visibility("default"))) int fl_debug_pd_timing(long long *out);
#endif

#undef PA_SEL

// ------------------------------------------------------------------------------------------------
// prefill attention at depth: the same function when a block's score rows no longer fit LDS (a second 512-token chunk
// of a long prompt: 32 rows x 1024+ keys).  Key-tiled, one launch; the score rows take ONE trip through a scratch
// buffer that only this workgroup touches (L2 / MALL resident) instead of four trips through HBM between three kernels.
//
// One 256-thread workgroup per (head, 32-row query block), heaviest blocks first, two workgroups per CU.
//   phase 0  exp table and the block's roped Q rows -> LDS
//   phase A  score tiles dealt round-robin to the 4 waves (the MFMA sequence of gemm_f32_abt_kernel), scaled, written
//            to the scratch rows; every lane keeps the running max of its 16 rows -> row max over lanes and waves
//   phase B  per row: f64 sum of the fp16 exp-table values of (score - max) -> 1 / sum        (softmax_rows_kernel)
//   phase C  per chunk of 256 keys: probabilities p = rn(val * inv) of the 32 rows -> LDS, then wave w accumulates
//            output columns [32w, 32w + 32) over the chunk's keys: A = p (LDS), B = transposed V cache
//   phase D  Q8_0 of the result (or the f32 rows)
// Scores, max, the exact f64 sum, p and the k order of the KQV accumulation (8-key steps pairing k with k + 4, then
// (k, k+1) pairs for the tail: chunk boundaries are multiples of 8) are those of the three-kernel path: bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int PD_T = 256, PD_NW = PD_T / 64, PD_CH = 256, PD_LD = PD_CH + 4;
#ifdef PA_TIMING
__device__ long long pd_dbg[64 * 8];
#define PD_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y < 64) pd_dbg[blockIdx.y * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PD_STAMP(k) do {} while (0)
#endif
#ifndef PD_ABL
#define PD_ABL 0   // timing experiments only (scripts/attn_deep.py), never defined in the product
#endif

// fp16 exp-table values of x[j] - mx (masked x = -inf -> 0), NV lookups in flight together.  The table index is taken
// from mx - x >= 0: rounding is sign-symmetric, so half(mx - x) is half(x - mx) without its sign bit, and the cases
// softmax_rows_kernel tells apart collapse into one compare: +0 -> entry 0 = exp(-0) = 1; [0, tab_n) the table; up to +inf
// (0x7C00; masked columns) exp underflows to 0 in fp16 (host-checked bound); NaNs and impossible negative differences -> NaN.
// The empty asm pins every LDS read as unconditional: left alone the compiler turns the selects into one branch per
// element around its lookup, which serialises the LDS round trips.
template <int NV>
__device__ __forceinline__ void pa_exp_vals(const float (&x)[NV], float mx, const uint16_t *tab, int tab_n, float (&out)[NV]) {
    uint32_t t[NV], idx[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        idx[j] = __half_as_ushort(__float2half_rn(mx - x[j]));
        t[j] = tab[min(idx[j], (uint32_t)(tab_n - 1))];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(t[j]));
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const uint32_t e = idx[j] < (uint32_t)tab_n ? t[j] : idx[j] <= 0x7C00u ? 0u : 0x7E00u;
        out[j] = __half2float(__ushort_as_half((uint16_t)e));
    }
}

template <int NSTEP>   // head_dim / 8
__global__ __launch_bounds__(PD_T, 2) void prefill_attention_deep_kernel(const float *__restrict__ qkv, int ldq, int N, int n_past,
                                                                         int n_ctx, int E, const float *__restrict__ kc,
                                                                         const float *__restrict__ vc,
                                                                         const uint16_t *__restrict__ exp_tab, int tab_n,
                                                                         float scale, float *S, int lds_, int64_t s_head,
                                                                         float *__restrict__ ao, int ldo,
                                                                         int8_t *__restrict__ oq, float *__restrict__ od,
                                                                         float *__restrict__ os) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    PD_STAMP(0);
    constexpr int D = NSTEP * 8, QLD = D + 4, NT = D / 32;
    const int h = blockIdx.x, nb = (N + 31) >> 5, mb = nb - 1 - (int)blockIdx.y;   // heavy blocks first
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kk = lane >> 5;
    const int P = n_past + N, m0 = mb * 32;
    const int ke = min(P, n_past + m0 + 32);                          // keys any row of the block can see
    const int ntl = (ke + 31) >> 5;
    uint16_t *tab = reinterpret_cast<uint16_t *>(psm);
    float *const U = reinterpret_cast<float *>(tab + ((tab_n + 7) & ~7));        // Q rows | p chunk | output tile
    constexpr int U_FLOATS = 32 * QLD > 32 * PD_LD ? 32 * QLD : 32 * PD_LD;
    static_assert(NT * 32 * 33 <= U_FLOATS, "output staging fits the union area");
    float *const wmax = U + U_FLOATS;                                 // [PD_NW][32]
    float *const rmax = wmax + PD_NW * 32;                            // [32]
    float *const rinv = rmax + 32;                                    // [32]
    float *const Srow = S + h * s_head;                               // scratch rows of this head: [N][lds_]
    auto row_len = [&](int row) { return min(P, n_past + m0 + row + 1); };   // rows >= N: duplicates of row N - 1 (L = P)
    // the head's scratch rows as a buffer: rows >= N and columns >= ld fall outside and are dropped / read as 0 by the range check
    __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(Srow, 0, (int)((int64_t)N * lds_ * 4), 0x00020000);

    // ---- phase 0 ----
    float4 kf[NSTEP];
    {
        const float *pb = kc + (int64_t)min(min(wave, ntl - 1) * 32 + r, P - 1) * E + h * D + 4 * kk;
#pragma unroll
        for (int j = 0; j < NSTEP; ++j) kf[j] = *reinterpret_cast<const float4 *>(pb + 8 * j);
    }
    for (int i = threadIdx.x; i < 32 * (D / 4); i += PD_T) {
        const int row = i / (D / 4), c4 = i % (D / 4);
        const float4 v = *reinterpret_cast<const float4 *>(qkv + (int64_t)min(m0 + row, N - 1) * ldq + h * D + c4 * 4);
        *reinterpret_cast<float4 *>(U + row * QLD + c4 * 4) = v;
    }
    for (int i = threadIdx.x; i * 8 < tab_n; i += PD_T)
        reinterpret_cast<uint4 *>(tab)[i] = reinterpret_cast<const uint4 *>(exp_tab + 0x8000)[i];
    __syncthreads();
    PD_STAMP(1);

    // ---- phase A: scores -> scratch, running row max ----
    {
        float mx[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) mx[i] = -INFINITY;
        const uint32_t voffS = (uint32_t)(((m0 + 4 * kk) * lds_ + r) * 4);
        const int full_end = (n_past + m0 + 1) >> 5;                  // tiles below: every column visible to every row
        for (int kt = wave; kt < ntl; kt += PD_NW) {
            // fragment j of the NEXT tile is requested as soon as this tile's MFMAs have read fragment j: one K tile of
            // registers, a full tile of MFMAs (4096 cycles) to cover the latency.  Past the end: the last tile again.
            const float *pbn = kc + (int64_t)min(min(kt + PD_NW, ntl - 1) * 32 + r, P - 1) * E + h * D + 4 * kk;
            const float *qa = U + r * QLD + 4 * kk;
            v16f acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            float4 qf = *reinterpret_cast<const float4 *>(qa);
#pragma unroll
            for (int j = 0; j < NSTEP; ++j) {
                const float4 qn = *reinterpret_cast<const float4 *>(qa + 8 * (j + 1 < NSTEP ? j + 1 : j));   // next step's A fragment
#if !(PD_ABL & 1)
                __builtin_amdgcn_s_setprio(2);       // the wave that has its operands issues its four MFMAs ahead of the other workgroup's VALU (-3 %)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.x, kf[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.y, kf[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.z, kf[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.w, kf[j].w, acc, 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
#else
                acc[j & 15] += qf.x * kf[j].x;
#endif
                kf[j] = *reinterpret_cast<const float4 *>(pbn + 8 * j);
                qf = qn;
                __builtin_amdgcn_sched_barrier(0);                     // (left alone the scheduler gathers the 16 loads at the end of the tile:
            }                                                          //  no prefetch distance, an exposed L2 round trip per tile)
            const int sbase = __builtin_amdgcn_readfirstlane(kt * 128);
            float sc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sc[i] = __fmul_rn(acc[i], scale);
#if !(PD_ABL & 16)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc[i]), rS, voffS, sbase + ((i & 3) + 8 * (i >> 2)) * lds_ * 4, 0);
#endif
            }
            if (kt >= full_end) {                                     // a tile the diagonal crosses: masked columns do not count
                const int col = kt * 32 + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) sc[i] = col < row_len((i & 3) + 8 * (i >> 2) + 4 * kk) ? sc[i] : -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm("v_max_f32 %0, %1, %2" : "=v"(mx[i]) : "v"(sc[i]), "v"(mx[i]));   // no NaN quieting moves
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {                                // max over the 32 columns a half-wave holds
            float m = mx[i];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float t = __shfl_xor(m, o, 64);
                m = t > m ? t : m;
            }
            if (r == 0) wmax[wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * kk] = m;
        }
    }
    PD_STAMP(2);
    __syncthreads();   // scratch rows and wave maxima are complete (workgroup-scope release / acquire)
    PD_STAMP(3);

    // ---- phase B: 1 / sum of the row's exp values; wave w owns rows w, w + 4, ... (8 rows), all in flight together ----
    constexpr int RPW = 32 / PD_NW;
    int Lq[RPW];
    float mq[RPW], iq[RPW];
    uint32_t vrow[RPW];                                               // byte offset of the row inside the head's scratch
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const int row = wave + PD_NW * u;
        Lq[u] = row_len(row);
        vrow[u] = (uint32_t)(min(m0 + row, N - 1) * lds_) * 4u;
        float m = wmax[row];
#pragma unroll
        for (int w = 1; w < PD_NW; ++w) m = fmaxf(m, wmax[w * 32 + row]);
        mq[u] = m;
    }
    float x[RPW][4], xn[RPW][4];
    auto load_x = [&](float (&dst)[RPW][4], int c0) __attribute__((always_inline)) {   // scores of columns c0 + lane + 64 j; masked: -inf
#pragma unroll
        for (int u = 0; u < RPW; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = c0 + lane + 64 * j;
                const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, vrow[u] + (uint32_t)min(i, Lq[u] - 1) * 4u, 0, 0));
                dst[u][j] = i < Lq[u] ? v : -INFINITY;
            }
    };
    {
        double sum[RPW];
#pragma unroll
        for (int u = 0; u < RPW; ++u) sum[u] = 0.0;
        load_x(x, 0);
        for (int c0 = 0; c0 < ((PD_ABL & 2) ? 1 : ke); c0 += PD_CH) {
            if (c0 + PD_CH < ke) load_x(xn, c0 + PD_CH);
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
                float v[4];
                pa_exp_vals<4>(x[u], mq[u], tab, tab_n, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[u] += (double)v[j];
            }
#pragma unroll
            for (int u = 0; u < RPW; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[u][j] = xn[u][j];
        }
#pragma unroll
        for (int u = 0; u < RPW; ++u) iq[u] = (float)(1.0 / wave_sum_f64(sum[u]));
    }
    load_x(x, 0);      // chunk 0 again, for phase C (L2 hits; in flight across the barrier)
    PD_STAMP(4);
    __syncthreads();   // every wave is done with the Q rows: U becomes the p chunk

    // ---- phase C: KQV over chunks of PD_CH keys ----
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int d0 = wave * 32;
    const bool pv = wave < NT;
    const float *pb = vc + (int64_t)(h * D + (pv ? d0 : 0) + r) * n_ctx;   // transposed V cache row (d0 + r)
    float4 b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0;
    auto load_v = [&](int kb) {
        b0 = *reinterpret_cast<const float4 *>(pb + kb + 4 * kk);
        b1 = *reinterpret_cast<const float4 *>(pb + kb + 8 + 4 * kk);
        b2 = *reinterpret_cast<const float4 *>(pb + kb + 16 + 4 * kk);
        b3 = *reinterpret_cast<const float4 *>(pb + kb + 24 + 4 * kk);
    };
    if (pv && 32 <= ke) load_v(0);
    for (int c0 = 0; c0 < ke; c0 += PD_CH) {
        // p = rn(val * inv) of the chunk's columns, rows wave + 4u (masked: val = 0)
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            float v[4];
#if !(PD_ABL & 4)
            pa_exp_vals<4>(x[u], mq[u], tab, tab_n, v);
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = x[u][j];
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j) U[(wave + PD_NW * u) * PD_LD + lane + 64 * j] = __fmul_rn(v[j], iq[u]);
        }
        if (c0 + PD_CH < ke) load_x(x, c0 + PD_CH);                   // next chunk's scores under this chunk's MFMAs
        __syncthreads();
        if (pv) {
            const float *pa = U + r * PD_LD - c0;                     // pa[k]: probability of key k for query row r
            const int kc_end = min(ke, c0 + PD_CH);
            int k = c0;
            for (; k + 32 <= kc_end; k += 32) {
                const float4 c0v = b0, c1v = b1, c2v = b2, c3v = b3;
                if (k + 64 <= ke) load_v(k + 32);                     // runs ahead across the chunk boundary
                const float4 a0 = *reinterpret_cast<const float4 *>(pa + k + 4 * kk);
                const float4 a1 = *reinterpret_cast<const float4 *>(pa + k + 8 + 4 * kk);
                const float4 a2 = *reinterpret_cast<const float4 *>(pa + k + 16 + 4 * kk);
                const float4 a3 = *reinterpret_cast<const float4 *>(pa + k + 24 + 4 * kk);
#if !(PD_ABL & 8)
#define FL_PV4(a, b)                                                      \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
#else
#define FL_PV4(a, b) acc[0] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
#endif
                FL_PV4(a0, c0v) FL_PV4(a1, c1v) FL_PV4(a2, c2v) FL_PV4(a3, c3v)
            }
            if (kc_end == ke) {                                       // last chunk: the tail of gemm_f32_abt_kernel
                for (; k + 8 <= ke; k += 8) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(pa + k + 4 * kk);
                    const float4 c0v = *reinterpret_cast<const float4 *>(pb + k + 4 * kk);
                    FL_PV4(a0, c0v)
                }
#undef FL_PV4
                for (; k < ke; k += 2) {
                    const float a = (k + kk < ke) ? pa[k + kk] : 0.f;
                    const float b = (k + kk < ke) ? pb[k + kk] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    PD_STAMP(5);
    // ---- phase D ----
    if (!oq) {
        if (pv) {
            const int col = h * D + d0 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * kk;
                if (row < N) ao[(int64_t)row * ldo + col] = acc[i];
            }
        }
        return;
    }
    if (pv) {
        float *T = U + wave * (32 * 33);
#pragma unroll
        for (int i = 0; i < 16; ++i) T[((i & 3) + 8 * (i >> 2) + 4 * kk) * 33 + r] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 32 * NT) {                                      // quantize_row_q8_0 per (token, 32 columns), lib/ggml.c:1341-1440
        const int w = threadIdx.x >> 5, rr = threadIdx.x & 31;
        const int n = m0 + rr, KBo = E >> 5, N16 = (N + 15) & ~15;
        if (n < N16) {
            const float *T = U + w * (32 * 33) + rr * 33;
            float v[32], amax = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                v[e] = n < N ? T[e] : 0.f;
                amax = fmaxf(amax, fabsf(v[e]));
            }
            const float dd = __fdiv_rn(amax, 127.0f);
            const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
            int qi[32], sum = 0;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                qi[e] = (int)rintf(__fmul_rn(v[e], id));
                sum += qi[e];
            }
            const int c = n & 15;
            const int64_t cb = ((int64_t)(n >> 4) * KBo + ((h * D + w * 32) >> 5)) * 16 + c;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                auto pk = [](int a, int b, int cc, int d) -> uint32_t {
                    return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(cc & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
                };
                *reinterpret_cast<uint2 *>(oq + cb * 32 + qw16_pos(c, g) * 8) =
                    make_uint2(pk(qi[8 * g], qi[8 * g + 2], qi[8 * g + 4], qi[8 * g + 6]), pk(qi[8 * g + 1], qi[8 * g + 3], qi[8 * g + 5], qi[8 * g + 7]));
            }
            od[cb] = dd;
            os[cb] = __fmul_rn(dd, (float)sum);
        }
    }
    PD_STAMP(6);
}

#ifdef PA_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_pd_timing(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pd_dbg), sizeof(long long) * 64 * 8); }
#endif
hipError_t prefill_attention_deep(const float *qkv, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc,
                                  const float *vc, const uint16_t *exp_tab, int tab_n, float scale, float *scratch, int ld_s,
                                  int64_t s_head, float *ao, int ldo, hipStream_t st, const fl_qact *qout) {
    const int P = n_past + N, nb = (N + 31) / 32;
    if (!scratch || (qout && (E % 32 != 0))) return hipErrorInvalidValue;
    // rows of the scratch must not share a cache line with another workgroup's rows; ld_s >= the keys a row can see
    if (D % 32 != 0 || D > 128 || (n_ctx & 3) != 0 || (ld_s & 31) != 0 || ld_s < P || tab_n < 0 || tab_n > 24576 || (tab_n & 7)) return hipErrorInvalidValue;
    if ((int64_t)N * ld_s * 4 >= (int64_t)1 << 31) return hipErrorInvalidValue;          // 32-bit offsets into a head's scratch rows
    const int qld = D + 4, uf = 32 * qld > 32 * PD_LD ? 32 * qld : 32 * PD_LD;
    const size_t lds = (size_t)tab_n * 2 + (size_t)uf * 4 + (PD_NW * 32 + 64) * 4;
    const dim3 grid(H, nb);
#define FL_PD(NS)                                                                                                       \
    do {                                                                                                                \
        static bool attr_set[64] = {};                                                                                  \
        int dev = 0;                                                                                                    \
        (void)hipGetDevice(&dev);                                                                                       \
        if (!attr_set[dev & 63]) {                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(prefill_attention_deep_kernel<NS>),       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                  \
            if (e != hipSuccess) return e;                                                                              \
            attr_set[dev & 63] = true;                                                                                  \
        }                                                                                                               \
        hipLaunchKernelGGL(prefill_attention_deep_kernel<NS>, grid, dim3(PD_T), lds, st, qkv, ldq, N, n_past, n_ctx, E, kc, vc, \
                           exp_tab, tab_n, scale, scratch, ld_s, s_head, ao, ldo, qout ? qout->q : nullptr,             \
                           qout ? qout->d : nullptr, qout ? qout->s : nullptr);                                         \
    } while (0)
    if (D == 128) FL_PD(16);
    else if (D == 96) FL_PD(12);
    else if (D == 64) FL_PD(8);
    else FL_PD(4);
#undef FL_PD
    return hipGetLastError();
}

// tab_n: number of fp16 exp-table entries after 0x8000 that are kept in LDS; every entry in (0x8000 + tab_n, 0xFC00] must
// be 0 (the caller checks that on the host table).  Returns hipErrorInvalidValue when the shape does not fit
// (caller falls back to the three-kernel path).
hipError_t prefill_attention(const float *qkv, int ldq, int D, int H, int N, int n_past, int n_ctx, int E, const float *kc,
                             const float *vc, const uint16_t *exp_tab, int tab_n, float scale, float *ao, int ldo,
                             hipStream_t st, const fl_qact *qout, float *scratch, int ld_s, int64_t s_head, int force_deep) {
    const int P = n_past + N, nb = (N + 31) / 32;
    if (qout && (E % 32 != 0)) return hipErrorInvalidValue;
    // score rows in LDS when they fit; else the key-tiled form with one trip through `scratch` ([H][>= N rows][ld_s] floats)
    auto deep = [&]() {
        return prefill_attention_deep(qkv, ldq, D, H, N, n_past, n_ctx, E, kc, vc, exp_tab, tab_n, scale, scratch, ld_s, s_head, ao,
                                      ldo, st, qout);
    };
    if (force_deep) return deep();
    if (D % 32 != 0 || D > 128 || (n_ctx & 3) != 0 || tab_n < 0 || tab_n > 24576 || (tab_n & 7)) return hipErrorInvalidValue;
    if (P > 64 * PA_SM_IT) return deep();
    const size_t tab_bytes = (size_t)tab_n * 2, qblk = (size_t)32 * (D + 4) * 4;   // exp table; Q rows of one block
    // pair mode: score rows of block x and of block nb-1-x: 32 * (2 n_past + 32 (nb + 1) + 8) floats (+64: round-ups)
    const size_t lds_pair = (size_t)32 * (2 * (size_t)n_past + 32 * (size_t)(nb + 1) + 8 + 64) * 4 + tab_bytes + 2 * qblk;
    const size_t lds_single = (size_t)32 * (((size_t)P + 31) / 32 * 32 + 4 + 4) * 4 + tab_bytes + qblk;
    const size_t cap = 160 * 1024;
    int pair = nb >= 2 && lds_pair <= cap ? 1 : 0;
    if (!pair && lds_single > cap) return deep();
    const size_t lds = pair ? lds_pair : lds_single;
    const dim3 grid(H, pair ? (nb + 1) / 2 : nb);
#define FL_PA(NS)                                                                                                       \
    do {                                                                                                                \
        static bool attr_set[64] = {};                                                                                  \
        int dev = 0;                                                                                                    \
        (void)hipGetDevice(&dev);                                                                                       \
        if (!attr_set[dev & 63]) {                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(prefill_attention_kernel<NS>),            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);                   \
            if (e != hipSuccess) return e;                                                                              \
            attr_set[dev & 63] = true;                                                                                  \
        }                                                                                                               \
        hipLaunchKernelGGL(prefill_attention_kernel<NS>, grid, dim3(PA_T), lds, st, qkv, ldq, N, n_past, n_ctx, E, kc, vc, \
                           exp_tab, tab_n, scale, ao, ldo, pair, qout ? qout->q : nullptr, qout ? qout->d : nullptr,    \
                           qout ? qout->s : nullptr);                                                                   \
    } while (0)
    if (D == 128) FL_PA(16);
    else if (D == 96) FL_PA(12);
    else if (D == 64) FL_PA(8);
    else FL_PA(4);
#undef FL_PA
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// diag_mask_inf + soft_max over one score row [0, P):  valid length L = n_past + n + 1
//   max over the row; val = fp16->f32(exp_tab[fp32->fp16(s - max)]); sum in f64 (exact: fp16 terms);
//   p = val * (float)(1.0 / sum); masked entries become 0.          lib/ggml.c:8558-8580
// one wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ S, int ld, int64_t sz, int N, int P,
                                                           int n_past, const uint16_t *__restrict__ exp_tab,
                                                           int rows_total, const int *__restrict__ dyn_past) {
    if (dyn_past) {
        n_past = *dyn_past;
        P = n_past + N;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int lane = threadIdx.x & 63;
    const int z = row / N, n = row % N;
    float *p = S + z * sz + (int64_t)n * ld;
    const int L = min(P, n_past + n + 1);
    float mx = -INFINITY;
    for (int i = lane; i < L; i += 64) mx = fmaxf(mx, p[i]);
    mx = wave_max_f32(mx);
    double sum = 0.0;
    for (int i = lane; i < L; i += 64) {
        const float x = p[i];
        float val = 0.f;
        if (x != -INFINITY) {
            const uint16_t h = __half_as_ushort(__float2half_rn(x - mx));
            val = __half2float(__ushort_as_half(exp_tab[h]));
        }
        sum += (double)val;
        p[i] = val;
    }
    sum = wave_sum_f64(sum);
    const float inv = (float)(1.0 / sum);
    for (int i = lane; i < P; i += 64) p[i] = i < L ? __fmul_rn(p[i], inv) : 0.f;
}

// The same row held in registers (P <= 64 * IT): every load of the row and every table gather is issued before the first
// is waited for.  The loop form above pays a dependent memory round trip per 64 columns, three times over -- at 2048
// keys that is most of the three-kernel attention path.  Same arithmetic; the f64 sum of fp16 terms is exact in any order.
// COMPACT (round 6): a probability is rn(t * inv) with t an fp16 table value and inv one f32 per row -- the row is written as its P fp16 values t
// (zeros past the last visible key) in the first half of its own storage plus inv in the row's last float (S[ld - 1]); the V.P kernel
// (attn_pv_exact, compact = true) forms the same products while it stages the piece.  Half the bytes written here and half the bytes of every
// re-read there (V.P reads each row once per feature block).  The row is complete in registers before the first store: in place is safe.
template <int IT, bool COMPACT>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float *__restrict__ S, int ld, int64_t sz, int N, int P,
                                                               int n_past, const uint16_t *__restrict__ exp_tab,
                                                               int rows_total) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int lane = threadIdx.x & 63;
    const int z = row / N, n = row % N;
    float *p = S + z * sz + (int64_t)n * ld;
    const int L = min(P, n_past + n + 1);
    float x[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int i = lane + 64 * u;
        const float v = p[min(i, L - 1)];                  // always inside the valid part of the row
        x[u] = i < L ? v : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < IT; ++u) mx = fmaxf(mx, x[u]);
    mx = wave_max_f32(mx);
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {                         // (-inf - mx rounds to fp16 -inf: a valid table index, entry 0)
        const float t = __half2float(__ushort_as_half(exp_tab[__half_as_ushort(__float2half_rn(x[u] - mx))]));
        x[u] = x[u] != -INFINITY ? t : 0.f;
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) sum += (double)x[u];
    sum = wave_sum_f64(sum);
    const float inv = (float)(1.0 / sum);
    if constexpr (COMPACT) {
        __half *ph = reinterpret_cast<__half *>(p);
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int i = lane + 64 * u;
            if (i < P) ph[i] = __float2half_rn(i < L ? x[u] : 0.f);      // (exact: x[u] is an fp16 table value or zero)
        }
        if (lane == 0) p[ld - 1] = inv;
    } else {
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int i = lane + 64 * u;
            if (i < P) p[i] = i < L ? __fmul_rn(x[u], inv) : 0.f;
        }
    }
}

__global__ void set_decode_inputs_kernel(int *__restrict__ dst, int n_past, int token) {
    dst[0] = n_past;
    dst[1] = token;
}
hipError_t set_decode_inputs(int *dst, int n_past, int token, hipStream_t st) {
    hipLaunchKernelGGL(set_decode_inputs_kernel, dim3(1), dim3(1), 0, st, dst, n_past, token);
    return hipGetLastError();
}

// compact: see softmax_rows_reg_kernel; hipErrorInvalidValue where that form does not reach (rows of more than 2048 keys, a device-side n_past)
hipError_t softmax_rows(float *S, int ld, int64_t sz, int N, int P, int n_past, int batch, const uint16_t *exp_tab,
                        hipStream_t st, const int *dyn_past, bool compact) {
    const int rows = N * batch;
    const dim3 grid((rows + 3) / 4), block(256);
    if (compact && (dyn_past || P > 64 * 32 || P < 4 || ld < P)) return hipErrorInvalidValue;
    if (!dyn_past && P <= 64 * 32) {
#define FL_SM(IT) do { if (compact) hipLaunchKernelGGL((softmax_rows_reg_kernel<IT, true>), grid, block, 0, st, S, ld, sz, N, P, n_past, exp_tab, rows); \
                       else hipLaunchKernelGGL((softmax_rows_reg_kernel<IT, false>), grid, block, 0, st, S, ld, sz, N, P, n_past, exp_tab, rows); } while (0)
        if (P <= 64 * 4) FL_SM(4);
        else if (P <= 64 * 8) FL_SM(8);
        else if (P <= 64 * 16) FL_SM(16);
        else FL_SM(32);
#undef FL_SM
        return hipGetLastError();
    }
    hipLaunchKernelGGL(softmax_rows_kernel, grid, block, 0, st, S, ld, sz, N, P, n_past, exp_tab, rows, dyn_past);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// decode attention (N = 1), one workgroup per head (round 6: two per 128-feature head, see the kernel): rope(q, k) -> KV store -> KQ*scale -> soft_max (fp16 table,
// f64 sum) -> KQV -> quantize_row_q8_0 of the head's 128 outputs straight into the QA1 workspace of the wo matmul.
// Replaces five launches (rope_kv, 2 x gemm_f32_abt, softmax_rows, quantize_q8) of the generic path; same op
// semantics, f32 dots in plain k order.  The position is read from device memory when dyn_past != null.
// ------------------------------------------------------------------------------------------------
//
// 512 threads.  K.q: 8 lanes share one cached position (each 16 B x D/32 pieces of the row, coalesced 128 B), 64
// positions per pass; KQV: 8 lanes share one row of the transposed V cache (32 positions per 128 B piece), 64 rows
// per pass.  The first 256 positions of both caches are requested before anything else is computed, so the two HBM
// round trips overlap the rope / soft_max latency chains.
#ifdef PA_TIMING
__device__ long long da_dbg[8];
#define DA_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) da_dbg[k] = wall_clock64(); } while (0)
#define DA_COMMIT() do {} while (0)
#elif defined(LLC_TIMING)   // every workgroup of every launch since the last reset, on the wall clock the GEMV ring uses (scripts/dev/decode_timeline.py)
constexpr unsigned DA_TL_CAP = 1u << 14;
__device__ long long da_tl[DA_TL_CAP * 8];
__device__ unsigned da_tl_cur;
#define DA_T_DECL long long tl_[7] = {0, 0, 0, 0, 0, 0, 0}
#define DA_STAMP(k) do { tl_[k] = wall_clock64(); } while (0)
#define DA_COMMIT()                                                                            \
    do {                                                                                       \
        if (threadIdx.x == 0) {                                                                \
            const unsigned s_ = atomicAdd(&da_tl_cur, 1u);                                     \
            if (s_ < DA_TL_CAP) {                                                              \
                for (int k_ = 0; k_ < 7; ++k_) da_tl[(size_t)s_ * 8 + k_] = tl_[k_];           \
                da_tl[(size_t)s_ * 8 + 7] = (900ll << 32) | blockIdx.x;                        \
            }                                                                                  \
        }                                                                                      \
    } while (0)
#else
#define DA_STAMP(k) do {} while (0)
#define DA_COMMIT() do {} while (0)
#endif
#ifndef DA_T_DECL
#define DA_T_DECL do {} while (0)
#endif
// Dots of the decode attention: 8 lanes per row, lane l8 holds the float4 pieces at element offsets 32 j + 4 l8 (+ c).
// ORD = 0: the fast order -- one accumulator per lane, pieces ascending, group8_sum_f32.
// ORD = 1: ggml_vec_dot_f32's order (exact mode; lib/ggml.c:2295-2330, reduction :1921-1936): element 32 i + 8 jj + l belongs to
//   lane l of accumulator sum[jj]; offset 4 l8 + c is jj = l8 >> 1, l = 4 (l8 & 1) + c, so component c of this lane's float4
//   accumulator IS sum[l8 >> 1][4 (l8 & 1) + c], chained over the 32-element steps in order.  Reduction: sum0 += sum1,
//   sum2 += sum3 (lanes l8 ^ 2), sum0 += sum2 (lanes l8 ^ 4), lo128 + hi128 (lanes l8 ^ 1), hadd, hadd ((x + y) + (z + w)).
//   f32 addition is commutative, so all 8 lanes end with the same bits.
template <int ORD>
__device__ __forceinline__ void att_fma4(float4 &a, const float4 x, const float4 y) {
    if (ORD == 0) {
        a.x = __fmaf_rn(x.x, y.x, a.x);
        a.x = __fmaf_rn(x.y, y.y, a.x);
        a.x = __fmaf_rn(x.z, y.z, a.x);
        a.x = __fmaf_rn(x.w, y.w, a.x);
    } else {
        a.x = __fmaf_rn(x.x, y.x, a.x);
        a.y = __fmaf_rn(x.y, y.y, a.y);
        a.z = __fmaf_rn(x.z, y.z, a.z);
        a.w = __fmaf_rn(x.w, y.w, a.w);
    }
}
constexpr int DPP_QUAD_REV = 0x1B;     // quad_perm [3,2,1,0]: lane i <-> i ^ 3
template <int ORD>
__device__ __forceinline__ float att_reduce8(float4 a) {
    if (ORD == 0) return group8_sum_f32(a.x);
    auto x4 = [](float v) { return dpp_f32<DPP_QUAD_REV>(dpp_f32<DPP_HALF_MIRROR>(v)); };     // lane i ^ 7 ^ 3 = i ^ 4
    a.x = __fadd_rn(a.x, dpp_f32<DPP_XOR2>(a.x)); a.y = __fadd_rn(a.y, dpp_f32<DPP_XOR2>(a.y));
    a.z = __fadd_rn(a.z, dpp_f32<DPP_XOR2>(a.z)); a.w = __fadd_rn(a.w, dpp_f32<DPP_XOR2>(a.w));
    a.x = __fadd_rn(a.x, x4(a.x)); a.y = __fadd_rn(a.y, x4(a.y));
    a.z = __fadd_rn(a.z, x4(a.z)); a.w = __fadd_rn(a.w, x4(a.w));
    a.x = __fadd_rn(a.x, dpp_f32<DPP_XOR1>(a.x)); a.y = __fadd_rn(a.y, dpp_f32<DPP_XOR1>(a.y));
    a.z = __fadd_rn(a.z, dpp_f32<DPP_XOR1>(a.z)); a.w = __fadd_rn(a.w, dpp_f32<DPP_XOR1>(a.w));
    return __fadd_rn(__fadd_rn(a.x, a.y), __fadd_rn(a.z, a.w));
}
// the n % 32 leftovers of that dot as the reference's build compiled them: chunks of 8, then one of 4 elements as rounded
// products added in order, the last n % 4 as FMAs.  The 8 lanes of a row each hold four of them (p4, v4: elements 4 l8 .. 4 l8 + 3
// behind the 32-wide body; what lies past n is ignored); s: the body's sum, the same bits in all 8 lanes.  The running value walks up
// the lanes (DPP row_shr:1), lane j adding its four rounded products -- or, the lane holding the last n % 4, its FMAs -- in order;
// the result is the lane's that holds the last elements -- lane min(n >> 2, 7), att_leftovers_last -- and the walk stops there (round 6: it used to
// go on to lane 7 whatever n).  (Round 3 had lane 0 walk the elements itself, each one a dependent global load: 3.7 us of the reference-
// order decode attention's 11.4 at n = 1 .. 31, profiles/r04_decode_exact.md.)
constexpr int DPP_ROW_SHR1 = 0x111;
__device__ __forceinline__ int att_leftovers_last(int n) { return min(n >> 2, 7); }
__device__ __forceinline__ float att_leftovers_lanes(float s, const float4 p4, const float4 v4, int n, int l8) {
#pragma clang fp contract(off)     // plain operators under this pragma: hipcc contracts a * b + c even across __fmul_rn / __fadd_rn
    const int nfull = n >> 2, nt = n & 3;
    const float px = p4.x * v4.x, py = p4.y * v4.y, pz = p4.z * v4.z, pw = p4.w * v4.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = s;
        if (l8 < nfull) {
            t = t + px;
            t = t + py;
            t = t + pz;
            t = t + pw;
        } else if (l8 == nfull) {
            if (nt > 0) t = __fmaf_rn(p4.x, v4.x, t);
            if (nt > 1) t = __fmaf_rn(p4.y, v4.y, t);
            if (nt > 2) t = __fmaf_rn(p4.z, v4.z, t);
        }
        const float up = dpp_f32<DPP_ROW_SHR1>(t);         // lane j + 1 takes over from lane j
        s = l8 == j + 1 ? up : l8 == j ? t : s;
        if (j >= nfull) break;                             // (uniform: n is the same for every lane) lane j held the last elements
    }
    return s;
}

constexpr int DA_T = 512, DA_KPRE = 4, DA_VPRE = 8;
constexpr int DA_PARTS = 2;     // workgroups per 128-feature head (four: 1.564 against 1.570 ms per token, within the noise; 64 features are one whole V pass)

template <int ORD>
// (argument order: the position pointer and what the first requests need lead -- those 14 dwords are preloaded into SGPRs,
//  -mllvm -amdgpu-kernarg-preload-count in build.sh, so the kernel's first round trip is the position, not its own arguments)
__global__ __launch_bounds__(DA_T) void decode_attention_kernel(const int *__restrict__ dyn_past, const float *__restrict__ qkv,
                                                                const float2 *__restrict__ rope_tab, float *__restrict__ kc,
                                                                float *__restrict__ vc, int E, int D, int n_past, int n_ctx,
                                                                const uint16_t *__restrict__ exp_tab, float scale,
                                                                int8_t *__restrict__ oq, float *__restrict__ od,
                                                                float *__restrict__ os, const TpTail *__restrict__ tt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    DA_T_DECL;
    DA_STAMP(0);
    if (dyn_past) n_past = *dyn_past;
    const int h = blockIdx.x, tid = threadIdx.x, pos = n_past, P = n_past + 1;
    float *qs = reinterpret_cast<float *>(dsm);           // [D] roped q
    float *ks = qs + D;                                    // [D] roped k (position pos)
    float *vs = ks + D;                                    // [D] v (position pos)
    float *out = vs + D;                                   // [D]
    float *sc = out + D;                                   // [n_ctx + 4] scores / probabilities
    double *red = reinterpret_cast<double *>(sc + n_ctx + 4);  // [8] reduction scratch
    float *redf = reinterpret_cast<float *>(red + 8);      // [8]
    const int l8 = tid & 7, r64 = tid >> 3;
    const int J = D >> 5;                                  // float4 pieces per lane and K row (<= 4)
    const float *kbase = kc + h * D + l8 * 4;
    // Round 6: a head of 128 features is TWO workgroups (blockIdx.y: features [64 y, 64 y + 64)).  Both compute the head's scores and soft_max -- the K rows
    // of the second reader hit the L2 of the XCD both run on (workgroup (h, y) -> XCD h % 8) -- and each streams the V rows of its own half, the half of a
    // head's bytes a single CU could not pull any faster (~20 GB/s: profiles/r06_decode_exact.md).  The fresh k row is stored by half 0, the fresh v
    // values and the Q8_0 blocks by the half that owns the features.
    const int parts = (int)gridDim.y, part = (int)blockIdx.y, Dp = D / parts, d0p = part * Dp;
    const int nvr = (Dp + 63) >> 6;                        // row passes of the V phase (<= 2)
    const int nchunk = (P + 31) >> 5;                      // 32-position pieces

    // ---- requests first.  Loads return in order: the few bytes rope needs (this token's q, k, v and the rope table row)
    //      go out before the K / V history, or rope would wait for all of it ----
    // (round 6 tried the K rows and V pieces of the first 256 positions requested WHATEVER the position is -- nothing waits for the position's round
    //  trip, what lies beyond it is masked -- with the rope row last: 591-597 against 611-620 tok/s, the over-fetch and the rope row queued behind
    //  the history cost more than the position's round trip, profiles/r06_decode_exact.md)
    const float *q = qkv + h * D, *k = qkv + E + h * D, *v = qkv + 2 * E + h * D;
    float2 cs = {0.f, 0.f}, xq = cs, xk = cs;
    float vv = 0.f;
    if (tid < D / 2) {
        cs = rope_tab[(int64_t)pos * (D >> 1) + tid];
        xq = *reinterpret_cast<const float2 *>(q + 2 * tid);
        xk = *reinterpret_cast<const float2 *>(k + 2 * tid);
    } else if (tid >= 64 && tid < 64 + D) {
        vv = v[tid - 64];
    }
    // K rows [0, 256) and V pieces [0, 256) of this head
    float4 kreg[DA_KPRE][4];
#pragma unroll
    for (int u = 0; u < DA_KPRE; ++u) {
        const int p = u * 64 + r64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < J && p < pos) kreg[u][j] = *reinterpret_cast<const float4 *>(kbase + (int64_t)p * E + j * 32);
    }
    float4 vreg[2][DA_VPRE];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        const int dl = rp * 64 + r64, d = d0p + dl;
#pragma unroll
        for (int c = 0; c < DA_VPRE; ++c)
            if (rp < nvr && dl < Dp && c < nchunk && c * 32 + l8 * 4 < P)
                vreg[rp][c] = *reinterpret_cast<const float4 *>(vc + (int64_t)(h * D + d) * n_ctx + c * 32 + l8 * 4);
    }

    DA_STAMP(1);
    // ---- rope(q), rope(k) -> LDS + K cache; v -> LDS + V cache ----
    if (tid < D / 2) {
        qs[2 * tid] = __fmaf_rn(xq.x, cs.x, -__fmul_rn(xq.y, cs.y));
        qs[2 * tid + 1] = __fmaf_rn(xq.x, cs.y, __fmul_rn(xq.y, cs.x));
        const float k0 = __fmaf_rn(xk.x, cs.x, -__fmul_rn(xk.y, cs.y)), k1 = __fmaf_rn(xk.x, cs.y, __fmul_rn(xk.y, cs.x));
        ks[2 * tid] = k0;
        ks[2 * tid + 1] = k1;
        if (part == 0) *reinterpret_cast<float2 *>(kc + (int64_t)pos * E + h * D + 2 * tid) = make_float2(k0, k1);
    } else if (tid >= 64 && tid < 64 + D) {
        const int d = tid - 64;
        vs[d] = vv;
        if (d >= d0p && d < d0p + Dp) vc[(int64_t)(h * D + d) * n_ctx + pos] = vv;
    }
    if (tid < 4) sc[P + tid] = 0.f;                        // tail of the last float4 of probabilities
    __syncthreads();

    DA_STAMP(2);
    // ---- scores ----
    float4 q4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < J) q4[j] = *reinterpret_cast<const float4 *>(qs + l8 * 4 + j * 32);
    float mx = -INFINITY;
    auto kq = [&](int p, const float4 *kr) {
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= J) break;
            const float4 k4 = p == pos ? *reinterpret_cast<const float4 *>(ks + l8 * 4 + j * 32) : kr[j];
            att_fma4<ORD>(a4, k4, q4[j]);
        }
        float a = att_reduce8<ORD>(a4);
        a = __fmul_rn(a, scale);
        if (l8 == 0) sc[p] = a;
        mx = fmaxf(mx, a);
    };
#pragma unroll
    for (int u = 0; u < DA_KPRE; ++u) {
        const int p = u * 64 + r64;
        if (p < P) kq(p, kreg[u]);
    }
    for (int p = DA_KPRE * 64 + r64; p < P; p += 64) {     // long contexts: the rest, pass by pass
        float4 kr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < J && p < pos) kr[j] = *reinterpret_cast<const float4 *>(kbase + (int64_t)p * E + j * 32);
        kq(p, kr);
    }
    mx = wave_max_f32(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int i = 1; i < DA_T / 64; ++i) mx = fmaxf(mx, redf[i]);

    DA_STAMP(3);
    // ---- soft_max: fp16 exp table, f64 sum (ggml_compute_forward_soft_max_f32) ----
    double sum = 0.0;
    for (int p = tid; p < P; p += DA_T) {
        const uint16_t hb = __half_as_ushort(__float2half_rn(sc[p] - mx));
        const float val = __half2float(__ushort_as_half(exp_tab[hb]));
        sum += (double)val;
        sc[p] = val;
    }
    sum = block_sum_f64(sum, red);
    const float inv = (float)(1.0 / sum);
    for (int p = tid; p < P; p += DA_T) sc[p] = __fmul_rn(sc[p], inv);
    __syncthreads();

    DA_STAMP(4);
    // ---- KQV ----
    auto with_fresh = [&](int d, int p0, float4 v4) {      // piece holding the fresh position (and what lies beyond)
        if (p0 + 3 >= pos) {
            const float fresh = vs[d];
            v4.x = p0 == pos ? fresh : p0 > pos ? 0.f : v4.x;
            v4.y = p0 + 1 == pos ? fresh : p0 + 1 > pos ? 0.f : v4.y;
            v4.z = p0 + 2 == pos ? fresh : p0 + 2 > pos ? 0.f : v4.z;
            v4.w = p0 + 3 == pos ? fresh : p0 + 3 > pos ? 0.f : v4.w;
        }
        return v4;
    };
    auto pv = [&](float4 &a, int d, int c, float4 v4) {
        const int p0 = c * 32 + l8 * 4;
        const float4 p4 = *reinterpret_cast<const float4 *>(sc + p0);
        att_fma4<ORD>(a, p4, with_fresh(d, p0, v4));
    };
    // ORD = 1: only whole 32-position steps go through the lanes; the P % 32 positions behind them are the reference's leftover loop
    const int nlane = ORD ? (P >> 5) : nchunk;
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        if (rp >= nvr) break;
        const int dl = rp * 64 + r64, d = d0p + dl;
        const bool mine = dl < Dp;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine) {
#pragma unroll
            for (int c = 0; c < DA_VPRE; ++c)
                if (c < nlane && c * 32 + l8 * 4 < P) pv(a4, d, c, vreg[rp][c]);
            for (int c = DA_VPRE; c < nlane; ++c)
                if (c * 32 + l8 * 4 < P)
                    pv(a4, d, c, *reinterpret_cast<const float4 *>(vc + (int64_t)(h * D + d) * n_ctx + c * 32 + l8 * 4));
        }
        float a = att_reduce8<ORD>(a4);
        if (ORD && (P & 31)) {                             // the piece behind the body: already in the lanes' registers (or one load away)
            const int np = P & ~31, p0 = np + l8 * 4;
            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nlane < DA_VPRE) {
#pragma unroll
                for (int c = 0; c < DA_VPRE; ++c)
                    if (c == nlane) v4 = vreg[rp][c];
            } else if (mine && p0 < P) {
                v4 = *reinterpret_cast<const float4 *>(vc + (int64_t)(h * D + d) * n_ctx + p0);
            }
            if (mine) {
                const float4 p4 = *reinterpret_cast<const float4 *>(sc + (p0 < P ? p0 : np));     // (a lane past P: ignored)
                a = att_leftovers_lanes(a, p4, with_fresh(d, p0, v4), P - np, l8);
                if (l8 == att_leftovers_last(P - np)) out[d] = a;
            }
        } else if (mine && l8 == 0) {
            out[d] = a;
        }
    }
    __syncthreads();
    DA_STAMP(5);
    // ---- Q8_0 of the head's outputs: D/8 groups, 4 adjacent lanes per block ----
    if (tid < Dp / 8) {
        float o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] = out[d0p + tid * 8 + i];
        quantize_store_group(o8, 0, ((h * D + d0p) >> 3) + tid, E >> 5, 1, oq, od, os);
    }
    DA_STAMP(6);
    DA_COMMIT();
    if (tt) {                        // tensor parallel: this head's Q8_0 blocks -> every peer's copy of the planes, then the exchange's tail (tp_tail.h)
        __syncthreads();
        tp_push(tt, oq + h * D + d0p, Dp);
        tp_push(tt, od + ((h * D + d0p) >> 5), Dp >> 3);
        tp_push(tt, os + ((h * D + d0p) >> 5), Dp >> 3);
        tp_tail<false, false, true>(tt);
    }
}
#if defined(LLC_TIMING) && !defined(PA_TIMING)
// stamps: t0 entry, t1 requests issued, t2 rope + K/V stores, t3 scores + max, t4 soft_max, t5 K.Q.V, t6 Q8_0 stored
extern "C" __attribute__((visibility("default"))) int fl_debug_da_timeline(long long *out, int max_rec, int reset) {
    unsigned n = 0;
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(da_tl_cur), &n, sizeof n);
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(da_tl_cur), sizeof n) != hipSuccess) return -1;
    if (n > DA_TL_CAP) n = DA_TL_CAP;
    if ((int)n > max_rec) n = (unsigned)max_rec;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(da_tl), sizeof(long long) * 8 * (size_t)n) != hipSuccess) return -1;
    return (int)n;
}
#endif
#ifdef PA_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_da_timing(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(da_dbg), sizeof(long long) * 8); }
#endif

hipError_t decode_attention(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab, float *kc,
                            float *vc, const uint16_t *exp_tab, float scale, const fl_qact *out, hipStream_t st,
                            const int *dyn_past, bool exact) {
    if (D % 32 != 0 || D > 128 || n_ctx % 4 != 0 || E % 4 != 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)(4 * D + n_ctx + 4) * 4 + 8 * 8 + 8 * 4;
    const TpTail *tt = tp_take_tail();
    if (exact)
        hipLaunchKernelGGL(decode_attention_kernel<1>, dim3(H, D == 128 ? DA_PARTS : 1), dim3(DA_T), lds, st, dyn_past, qkv, reinterpret_cast<const float2 *>(rope_tab),
                           kc, vc, E, D, n_past, n_ctx, exp_tab, scale, out->q, out->d, out->s, tt);
    else
        hipLaunchKernelGGL(decode_attention_kernel<0>, dim3(H, D == 128 ? DA_PARTS : 1), dim3(DA_T), lds, st, dyn_past, qkv, reinterpret_cast<const float2 *>(rope_tab),
                           kc, vc, E, D, n_past, n_ctx, exp_tab, scale, out->q, out->d, out->s, tt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// decode attention for long contexts: the same arithmetic in TWO launches that spread one head over many CUs.
// One workgroup per head streams 8 KiB of K/V per cached position through a single CU -- beyond a few hundred
// positions that dominates the token (7B, n_past 1900: 95 us of a 133 us layer).
//   decode_scores_kernel  grid (heads, 64-position slices): rope(q) [+ rope(k), KV store in the slice that owns the
//                         fresh position], scaled K.q of the slice -> scores[head][p] in HBM
//   decode_pv_kernel      grid (heads, D/32): soft_max of the head's score row in LDS (every workgroup repeats it: P
//                         table lookups), KQV for 32 output features, Q8_0 block of those 32 features
// Every dot is accumulated in the order decode_attention_kernel uses (8 lanes per row, float4 pieces ascending), the
// max is exact and the f64 sum of fp16 table values is exact in any order: the result is bit-identical to the
// one-launch kernel (tests/test_eval_ops_gpu.py::test_decode_attention_split_equals_fused).
// ------------------------------------------------------------------------------------------------
// (round 6: slices of 64 positions -- 128: 547 -> 571 tok/s at 512 positions, 513 -> 530 at 1000; 32 positions of 256 threads: the same as 64)
constexpr int DS_T = 512, DS_POS = 64, DS_PP = DS_T / 8, DP_T = 256, DP_BATCH = 16;      // (DS_PP: positions per pass, eight lanes each)
static_assert(DS_POS % DS_PP == 0 && DS_T >= 256, "whole passes; the rope / v threads");

template <int ORD>
__global__ __launch_bounds__(DS_T) void decode_scores_kernel(const int *__restrict__ dyn_past, const float *__restrict__ qkv, int E, int D,
                                                             int n_past, int n_ctx, const float2 *__restrict__ rope_tab,
                                                             float *__restrict__ kc, float *__restrict__ vc, float scale,
                                                             float *__restrict__ scores) {      // (position pointer first: preloaded kernargs)
    __shared__ __attribute__((aligned(16))) float qs[128], ks[128];
    if (dyn_past) n_past = *dyn_past;
    const int h = blockIdx.x, tid = threadIdx.x, pos = n_past, P = n_past + 1;
    const int p0 = blockIdx.y * DS_POS;
    if (p0 >= P) return;                                   // the grid is sized for n_ctx when the position is dynamic
    const bool owner = pos < p0 + DS_POS;                  // this slice holds the fresh position
    const int l8 = tid & 7, r64 = tid >> 3, J = D >> 5;
    const float *q = qkv + h * D, *k = qkv + E + h * D, *v = qkv + 2 * E + h * D;
    float2 cs = {0.f, 0.f}, xq = cs, xk = cs;
    float vv = 0.f;
    if (tid < D / 2) {
        cs = rope_tab[(int64_t)pos * (D >> 1) + tid];
        xq = *reinterpret_cast<const float2 *>(q + 2 * tid);
        if (owner) xk = *reinterpret_cast<const float2 *>(k + 2 * tid);
    } else if (owner && tid >= 64 && tid < 64 + D) {
        vv = v[tid - 64];
    }
    const float *kbase = kc + h * D + l8 * 4;
    float4 kreg[DS_POS / DS_PP][4];
#pragma unroll
    for (int u = 0; u < DS_POS / DS_PP; ++u) {
        const int p = p0 + u * DS_PP + r64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < J && p < pos) kreg[u][j] = *reinterpret_cast<const float4 *>(kbase + (int64_t)p * E + j * 32);
    }
    if (tid < D / 2) {
        qs[2 * tid] = __fmaf_rn(xq.x, cs.x, -__fmul_rn(xq.y, cs.y));
        qs[2 * tid + 1] = __fmaf_rn(xq.x, cs.y, __fmul_rn(xq.y, cs.x));
        if (owner) {
            const float k0 = __fmaf_rn(xk.x, cs.x, -__fmul_rn(xk.y, cs.y)), k1 = __fmaf_rn(xk.x, cs.y, __fmul_rn(xk.y, cs.x));
            ks[2 * tid] = k0;
            ks[2 * tid + 1] = k1;
            *reinterpret_cast<float2 *>(kc + (int64_t)pos * E + h * D + 2 * tid) = make_float2(k0, k1);
        }
    } else if (owner && tid >= 64 && tid < 64 + D) {
        vc[(int64_t)(h * D + tid - 64) * n_ctx + pos] = vv;
    }
    __syncthreads();
    float4 q4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < J) q4[j] = *reinterpret_cast<const float4 *>(qs + l8 * 4 + j * 32);
#pragma unroll
    for (int u = 0; u < DS_POS / DS_PP; ++u) {
        const int p = p0 + u * DS_PP + r64;
        if (p >= P) continue;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= J) break;
            const float4 k4 = p == pos ? *reinterpret_cast<const float4 *>(ks + l8 * 4 + j * 32) : kreg[u][j];
            att_fma4<ORD>(a4, k4, q4[j]);
        }
        const float a = att_reduce8<ORD>(a4);
        if (l8 == 0) scores[(int64_t)h * n_ctx + p] = __fmul_rn(a, scale);
    }
}

template <int ORD>
__global__ __launch_bounds__(DP_T) void decode_pv_kernel(const int *__restrict__ dyn_past, const float *__restrict__ scores, int E, int D,
                                                         int n_past, int n_ctx, const float *__restrict__ vc,
                                                         const uint16_t *__restrict__ exp_tab, int8_t *__restrict__ oq,
                                                         float *__restrict__ od, float *__restrict__ os, const TpTail *__restrict__ tt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    if (dyn_past) n_past = *dyn_past;
    const int h = blockIdx.x, tid = threadIdx.x, pos = n_past, P = n_past + 1;
    double *red = reinterpret_cast<double *>(dsm);         // [4]
    float *redf = reinterpret_cast<float *>(red + 4);      // [4]
    float *out = redf + 4;                                 // [32]
    float *sc = out + 32;                                  // [roundup(n_ctx, 512) + 512] scores / probabilities
    const int l8 = tid & 7, d = blockIdx.y * 32 + (tid >> 3);
    const int nbatch = (P + DP_BATCH * 32 - 1) / (DP_BATCH * 32);   // batches of 16 pieces of 32 positions
    const float *vrow = vc + (int64_t)(h * D + d) * n_ctx;
    // Loads are unconditional (a piece past the position re-reads the row's first piece, which is cache-hot, and is
    // zeroed on use): a load under a lane- or even wave-dependent condition makes the compiler wait for ALL loads in
    // flight before it (s_waitcnt vmcnt(0) per load -- measured 33 us instead of 17 us at 1900 positions, 7B).
    float4 va[DP_BATCH], vb[DP_BATCH];
    auto load = [&](float4 *r, int b) {
#pragma unroll
        for (int c = 0; c < DP_BATCH; ++c) {
            const int pp = (b * DP_BATCH + c) * 32 + l8 * 4;
            r[c] = *reinterpret_cast<const float4 *>(vrow + (pp < P ? pp : l8 * 4));
        }
    };
    load(va, 0);                                           // V does not depend on the scores: in flight under the soft_max
    float4 vleft = make_float4(0.f, 0.f, 0.f, 0.f);        // ORD = 1: this lane's four of the P % 32 positions behind the body
    if (ORD) vleft = *reinterpret_cast<const float4 *>(vrow + ((P & ~31) + l8 * 4 < P ? (P & ~31) + l8 * 4 : l8 * 4));

    // ---- soft_max of the head's row: fp16 exp table, f64 sum (ggml_compute_forward_soft_max_f32) ----
    // (8 entries per thread and pass, their loads / table gathers issued together: with one entry per loop iteration
    //  every 256 positions cost a dependent memory round trip)
    const float *srow = scores + (int64_t)h * n_ctx;
    constexpr int SM_U = 8;
    float mx = -INFINITY;
    for (int p0 = tid; p0 < P; p0 += DP_T * SM_U) {
        float x[SM_U];
#pragma unroll
        for (int i = 0; i < SM_U; ++i) x[i] = srow[p0 + i * DP_T < P ? p0 + i * DP_T : 0];
#pragma unroll
        for (int i = 0; i < SM_U; ++i) {
            const int p = p0 + i * DP_T;
            if (p < P) {
                sc[p] = x[i];
                mx = fmaxf(mx, x[i]);
            }
        }
    }
    mx = wave_max_f32(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double sum = 0.0;
    for (int p0 = tid; p0 < P; p0 += DP_T * SM_U) {        // each thread revisits its own entries
        float val[SM_U];
#pragma unroll
        for (int i = 0; i < SM_U; ++i) {                   // past the row: exp_tab[half(-inf)] = 0, never stored
            const int p = p0 + i * DP_T;
            const float x = p < P ? sc[p] : -INFINITY;
            val[i] = __half2float(__ushort_as_half(exp_tab[__half_as_ushort(__float2half_rn(x - mx))]));
        }
#pragma unroll
        for (int i = 0; i < SM_U; ++i) {
            const int p = p0 + i * DP_T;
            if (p < P) {
                sum += (double)val[i];
                sc[p] = val[i];
            }
        }
    }
    sum = block_sum_f64(sum, red);
    const float inv = (float)(1.0 / sum);
    for (int p = tid; p < P; p += DP_T) sc[p] = __fmul_rn(sc[p], inv);
    for (int p = P + tid; p < (nbatch + 1) * DP_BATCH * 32; p += DP_T) sc[p] = 0.f;   // probabilities the batches over-read
    __syncthreads();

    // ---- KQV for feature d: lane l8 owns positions 32c + 4 l8 .. +3, pieces in ascending order ----
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int np = ORD ? (P & ~31) : INT_MAX;              // ORD = 1: positions from np on are the reference's leftover loop
    auto consume = [&](const float4 *r, int b) {
#pragma unroll
        for (int c = 0; c < DP_BATCH; ++c) {
            const int pp = (b * DP_BATCH + c) * 32 + l8 * 4;
            float4 v4 = r[c];
            const int lim = min(pos, np - 1);
            v4.x = pp > lim ? 0.f : v4.x;                  // beyond the fresh position: stale cache / clamped loads
            v4.y = pp + 1 > lim ? 0.f : v4.y;
            v4.z = pp + 2 > lim ? 0.f : v4.z;
            v4.w = pp + 3 > lim ? 0.f : v4.w;
            const float4 p4 = *reinterpret_cast<const float4 *>(sc + pp);
            att_fma4<ORD>(a4, p4, v4);
        }
    };
    for (int b = 0; b < nbatch; b += 2) {
        load(vb, b + 1);
        consume(va, b);
        load(va, b + 2);
        consume(vb, b + 1);
    }
    float a = att_reduce8<ORD>(a4);
    if (ORD && (P & 31)) {
        const float4 p4 = *reinterpret_cast<const float4 *>(sc + np + l8 * 4);
        a = att_leftovers_lanes(a, p4, vleft, P - np, l8);
        if (l8 == att_leftovers_last(P - np)) out[tid >> 3] = a;
    } else if (l8 == 0) {
        out[tid >> 3] = a;
    }
    __syncthreads();
    if (tid < 4) {                                         // one Q8_0 block: 4 adjacent lanes x 8 features
        float o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] = out[tid * 8 + i];
        quantize_store_group(o8, 0, ((h * D + blockIdx.y * 32) >> 3) + tid, E >> 5, 1, oq, od, os);
    }
    if (tt) {
        const int b = (h * D + blockIdx.y * 32) >> 5;
        __syncthreads();
        tp_push(tt, oq + b * 32, 32);
        tp_push(tt, od + b, 4);
        tp_push(tt, os + b, 4);
        tp_tail<false, false, true>(tt);
    }
}

hipError_t decode_attention_split(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab,
                                  float *kc, float *vc, const uint16_t *exp_tab, float scale, float *scores,
                                  const fl_qact *out, hipStream_t st, const int *dyn_past, bool exact) {
    if (D % 32 != 0 || D > 128 || n_ctx % 4 != 0 || E % 4 != 0) return hipErrorInvalidValue;
    const TpTail *tt = tp_take_tail();
    const int slices = dyn_past ? (n_ctx + DS_POS - 1) / DS_POS : (n_past + DS_POS) / DS_POS;
    if (exact)
        hipLaunchKernelGGL(decode_scores_kernel<1>, dim3(H, slices), dim3(DS_T), 0, st, dyn_past, qkv, E, D, n_past, n_ctx,
                           reinterpret_cast<const float2 *>(rope_tab), kc, vc, scale, scores);
    else
        hipLaunchKernelGGL(decode_scores_kernel<0>, dim3(H, slices), dim3(DS_T), 0, st, dyn_past, qkv, E, D, n_past, n_ctx,
                           reinterpret_cast<const float2 *>(rope_tab), kc, vc, scale, scores);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t lds = 4 * 8 + 4 * 4 + 32 * 4 + (size_t)((n_ctx + 511) / 512 * 512 + 512) * 4;
    if (exact)
        hipLaunchKernelGGL(decode_pv_kernel<1>, dim3(H, D / 32), dim3(DP_T), lds, st, dyn_past, scores, E, D, n_past, n_ctx, vc, exp_tab,
                           out->q, out->d, out->s, tt);
    else
        hipLaunchKernelGGL(decode_pv_kernel<0>, dim3(H, D / 32), dim3(DP_T), lds, st, dyn_past, scores, E, D, n_past, n_ctx, vc, exp_tab,
                           out->q, out->d, out->s, tt);
    return hipGetLastError();
}

// local (single-process) all-reduce of the tensor-parallel partial sums: rank-order sum written back to every shard
struct SumBufs { float *p[FL_COMM_MAX_LOCAL]; };
__global__ __launch_bounds__(256) void sum_buffers_kernel(SumBufs b, int world, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float s = b.p[0][i];
        for (int r = 1; r < world; ++r) s += b.p[r][i];
        for (int r = 0; r < world; ++r) b.p[r][i] = s;
    }
}
hipError_t sum_buffers_inplace(float *const *bufs, int world, size_t count, hipStream_t st) {
    if (world < 1 || world > FL_COMM_MAX_LOCAL) return hipErrorInvalidValue;
    SumBufs b;
    for (int r = 0; r < FL_COMM_MAX_LOCAL; ++r) b.p[r] = r < world ? bufs[r] : nullptr;
    const int grid = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    hipLaunchKernelGGL(sum_buffers_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, b, world, count);
    return hipGetLastError();
}

// One-shot all-reduce / all-gather of a SMALL message between the tensor-parallel ranks of one node (the 16-32 KB partial
// sums of a decode token): every rank owns an exchange buffer that its peers have mapped (hipIpc; xGMI between GPUs).
// Epoch e, slot e & 1: copy the local vector into the own slot (system-scope stores) -> release flag[slot] = e -> wait until
// every peer's flag says e -> read all slots and add them in RANK ORDER (the order sum_buffers_kernel uses, identical on every
// rank, so the ranks stay bit-identical).  Two slots: a rank can only start epoch e + 1 after every peer has published e,
// i.e. after it finished reading epoch e - 1's slot.  One kernel, no host involvement: it can be captured in the decode hipGraph
// (the epoch counter lives in device memory).  One workgroup: the message is a few pages and the cost is the round trip.
__global__ __launch_bounds__(1024) void p2p_exchange_kernel(P2PPeers a, float *data, unsigned count, float *gather_out) {
    __shared__ unsigned ep_s;
    const unsigned tid = threadIdx.x;
    if (tid == 0) ep_s = *a.epoch + 1;
    __syncthreads();
    const unsigned e = ep_s, slot = e & 1;
    float *mine = a.buf[a.rank] + (size_t)slot * a.cap;
    for (unsigned i = tid; i < count; i += 1024) __hip_atomic_store(mine + i, data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flag[a.rank] + slot, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < (unsigned)a.world && (int)tid != a.rank) {
        const unsigned *f = a.flag[tid] + slot;
        // bounded: a peer that died must not hang this GPU's queue for ever.  wall_clock64 ticks at 100 MHz; the bound is ~20 s (p2p_timeout_ticks), far beyond
        // any legitimate skew between ranks; the timeout is recorded next to the epoch (fl_comm_p2p_timeouts) and the sum of
        // whatever is in the slots goes on -- the host sees the count and fails the eval.
        const unsigned long long t0 = wall_clock64();
        // (relaxed polls and one acquire behind them: an acquire load invalidates the caches every time round the loop -- round 5, tp_tail.h)
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > a.timeout_ticks) {
                atomicAdd(a.epoch + 1, 1u);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (!gather_out) {
        for (unsigned i = tid; i < count; i += 1024) {
            float s = __hip_atomic_load(a.buf[0] + (size_t)slot * a.cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int r = 1; r < a.world; ++r)
                s += __hip_atomic_load(a.buf[r] + (size_t)slot * a.cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            data[i] = s;
        }
    } else {
        for (int r = 0; r < a.world; ++r)
            for (unsigned i = tid; i < count; i += 1024)
                gather_out[(size_t)r * count + i] = __hip_atomic_load(a.buf[r] + (size_t)slot * a.cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid == 0) *a.epoch = e;
}
hipError_t p2p_exchange(const P2PPeers &peers, float *data, size_t count, float *gather_out, hipStream_t st) {
    if (count == 0 || count > peers.cap || peers.world < 2 || peers.world > FL_COMM_MAX_LOCAL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(p2p_exchange_kernel, dim3(1), dim3(1024), 0, st, peers, data, (unsigned)count, gather_out);
    return hipGetLastError();
}

// the tensor-parallel exchange tail as a launch of its own (tp_tail.h): behind a producer whose kernel does not carry it
thread_local const TpTail *tp_pending_tail = nullptr;
__global__ __launch_bounds__(256) void tp_tail_kernel(const TpTail *__restrict__ tt) { tp_tail<false, true>(tt); }
hipError_t tp_tail_launch(const TpTail *tt_dev, hipStream_t st) {
    hipLaunchKernelGGL(tp_tail_kernel, dim3(1), dim3(256), 0, st, tt_dev);
    return hipGetLastError();
}

// The exchange's self-test AS THE DECODE PATH USES IT (comm.cpp p2p_selftest; ADVICE r5 / VERDICT r5 item 7).  One epoch = two launches:
//   produce: several workgroups; each first READS every rank's slice of the test area with plain loads -- the lines of the previous epoch are now in
//            this CU's L1 and this XCD's L2, as a decode launch's operands of the previous token are -- then writes its share of this rank's slice with
//            tp_put (own region written through + every peer's region, many workgroups at once), then the fused tail: ticket, publish, wait;
//   consume: the NEXT launch reads every rank's slice with PLAIN loads, as the next decode launch reads its operand, and counts words that are not
//            this epoch's pattern; its own tail (another exchange kind, no data) keeps any rank from producing epoch e + 1 into slots a peer is still reading.
// The same slots every epoch.  pattern(e, r, i) = e * 0x9E3779B9 ^ r << 20 ^ i * 2654435761.
__device__ __forceinline__ unsigned tp_selftest_pattern(unsigned e, unsigned r, unsigned i) { return (e * 0x9E3779B9u) ^ (r << 20) ^ (i * 2654435761u); }
__global__ __launch_bounds__(256) void tp_selftest_produce_kernel(const TpTail *__restrict__ tt, unsigned area_off, unsigned slice_words, unsigned epoch,
                                                                  unsigned *__restrict__ sink) {
    const int world = tt->world, rank = tt->rank;
    const unsigned *area = reinterpret_cast<const unsigned *>(tt->region[rank] + area_off);
    unsigned acc = 0;
    for (unsigned i = threadIdx.x; i < slice_words * (unsigned)world; i += 256) acc ^= area[i];        // plain loads: cache what the previous epoch left
    if (acc == 0x13572468u) sink[blockIdx.x] = acc;                                                    // (keeps the loads alive)
    __syncthreads();
    float *mine = reinterpret_cast<float *>(tt->region[rank] + area_off) + (size_t)rank * slice_words;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < slice_words; i += gridDim.x * 256)
        tp_put(tt, mine + i, __uint_as_float(tp_selftest_pattern(epoch, (unsigned)rank, i)));
    tp_tail<false, false, true>(tt);
}
__global__ __launch_bounds__(256) void tp_selftest_consume_kernel(const TpTail *__restrict__ barrier_tt, unsigned area_off, unsigned slice_words, unsigned epoch,
                                                                  unsigned *__restrict__ errors) {
    const int world = barrier_tt->world, rank = barrier_tt->rank;
    const unsigned *area = reinterpret_cast<const unsigned *>(barrier_tt->region[rank] + area_off);
    unsigned bad = 0;
    for (int r = 0; r < world; ++r)
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < slice_words; i += gridDim.x * 256)
            bad += area[(size_t)r * slice_words + i] != tp_selftest_pattern(epoch, (unsigned)r, i);     // plain loads, the launch after the exchange
    if (bad) atomicAdd(errors, bad);
    tp_tail<false, false, true>(barrier_tt);
}
hipError_t tp_selftest_epoch(const TpTail *exchange_dev, const TpTail *barrier_dev, unsigned area_off, unsigned slice_words, unsigned epoch,
                             unsigned *errors_dev, unsigned *sink_dev, hipStream_t st) {
    hipLaunchKernelGGL(tp_selftest_produce_kernel, dim3(8), dim3(256), 0, st, exchange_dev, area_off, slice_words, epoch, sink_dev);
    hipLaunchKernelGGL(tp_selftest_consume_kernel, dim3(8), dim3(256), 0, st, barrier_dev, area_off, slice_words, epoch, errors_dev);
    return hipGetLastError();
}

// out[n][e] = a[n][e] + b[n][e]   (ggml_add)
__global__ void add_rows_kernel(const float *__restrict__ a, int lda, const float *__restrict__ b, int ldb,
                                float *__restrict__ o, int ldo, int N, int E) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e4 = E >> 2;
    if (gid >= (int64_t)N * e4) return;
    const int n = (int)(gid / e4), e = (int)(gid % e4) * 4;
    const float4 x = *reinterpret_cast<const float4 *>(a + (int64_t)n * lda + e);
    const float4 y = *reinterpret_cast<const float4 *>(b + (int64_t)n * ldb + e);
    *reinterpret_cast<float4 *>(o + (int64_t)n * ldo + e) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

hipError_t add_rows(const float *a, int lda, const float *b, int ldb, float *o, int ldo, int N, int E, hipStream_t st) {
    const int64_t total = (int64_t)N * (E >> 2);
    hipLaunchKernelGGL(add_rows_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, a, lda, b, ldb, o, ldo, N, E);
    return hipGetLastError();
}


// -log softmax(logits[j])[next token of j] for rows j0 + blockIdx.x -- the inner loop of FastLlama::perplexity
// (/root/reference/lib/bridge.cpp:397-407: max, sum of expf(l - max), p = expf(l[t] - max) / sum, -log p) on the device,
// so that a perplexity run copies one double per row to the host instead of n_vocab floats.
__global__ __launch_bounds__(256) void logits_nll_kernel(const float *__restrict__ logits, int ld, int V, const int *__restrict__ next_tok,
                                                         int j0, double *__restrict__ out) {
    __shared__ float smx[4];
    __shared__ double ssum[4];
    const int j = j0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *l = logits + (int64_t)j * ld;
    float mx = -INFINITY;
    for (int k = tid; k < V; k += 256) mx = fmaxf(mx, l[k]);
    mx = wave_max_f32(mx);
    if (lane == 0) smx[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    double sum = 0.0;
    for (int k = tid; k < V; k += 256) sum += (double)expf(l[k] - mx);
    sum = wave_sum_f64(sum);
    if (lane == 0) ssum[w] = sum;
    __syncthreads();
    if (tid == 0) {
        const float tot = (float)((ssum[0] + ssum[1]) + (ssum[2] + ssum[3]));
        const float pr = expf(l[next_tok[blockIdx.x]] - mx) / tot;
        out[blockIdx.x] = (double)(-logf(pr));
    }
}

hipError_t logits_nll(const float *logits, int ld, int V, const int *next_tok_dev, int j0, int rows, double *out_dev, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(logits_nll_kernel, dim3(rows), dim3(256), 0, st, logits, ld, V, next_tok_dev, j0, out_dev);
    return hipGetLastError();
}


// out[n][r * Vl + j] = tmp[r][n][j]: the all-gathered logits slices of the row-split lm-head back into [N][n_vocab] rows
__global__ void gather_cols_kernel(const float *__restrict__ tmp, int G, int N, int Vl, int ldp, float *__restrict__ out, int ldo) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)G * N * Vl) return;
    const int j = (int)(gid % Vl), n = (int)((gid / Vl) % N), r = (int)(gid / ((int64_t)Vl * N));
    out[(int64_t)n * ldo + (int64_t)r * Vl + j] = tmp[((int64_t)r * N + n) * ldp + j];
}
hipError_t gather_cols(const float *tmp, int G, int N, int Vl, int ldp, float *out, int ldo, hipStream_t st) {
    const int64_t total = (int64_t)G * N * Vl;
    hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tmp, G, N, Vl, ldp, out, ldo);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Row-split tensor parallelism (reference-order mode): every matmul is split by OUTPUT rows -- the reference's own split across
// threads (/root/reference/lib/ggml.c:8127-8135) -- so each output is one rank's full-K dot in the reference's order and nothing is
// ever summed across ranks.  What travels instead: the Q8_0 operand of wo / w2 (each rank quantized the 32-element blocks of its
// own features) and the f32 output rows, both by all-gather.  The three kernels below are the byte work around those collectives.
// ------------------------------------------------------------------------------------------------
// dst <- seg0 | seg1 | seg2 (4-byte units): the used part of a Q8_0 workspace's q, d, s planes as ONE message
__global__ void pack3_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ s0, int64_t n0, const uint32_t *__restrict__ s1, int64_t n1,
                             const uint32_t *__restrict__ s2, int64_t n2) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n0) dst[gid] = s0[gid];
    else if (gid < n0 + n1) dst[gid] = s1[gid - n0];
    else if (gid < n0 + n1 + n2) dst[gid] = s2[gid - n0 - n1];
}
hipError_t pack3(void *dst, const void *s0, size_t b0, const void *s1, size_t b1, const void *s2, size_t b2, hipStream_t st) {
    if ((b0 | b1 | b2) & 3) return hipErrorInvalidValue;
    const int64_t total = (int64_t)((b0 + b1 + b2) >> 2);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pack3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, reinterpret_cast<uint32_t *>(dst),
                       reinterpret_cast<const uint32_t *>(s0), (int64_t)(b0 >> 2), reinterpret_cast<const uint32_t *>(s1), (int64_t)(b1 >> 2),
                       reinterpret_cast<const uint32_t *>(s2), (int64_t)(b2 >> 2));
    return hipGetLastError();
}

// The all-gathered messages (`stage`: G messages of `msg` bytes, rank order) back into full-K planes.  In every layout here a rank's
// K slice is a run of `chunk` bytes per row of the plane (QA16: row = column group, chunk = KB_local * 512 for q, KB_local * 64 for
// d / s; QA1: one row): out[row][r][chunk] = message r [seg offset + row * chunk ...].  4-byte units.
__global__ void unpack3_kernel(const uint32_t *__restrict__ stage, int64_t msg_w, int G, int rows, uint32_t *__restrict__ o0, int c0,
                               uint32_t *__restrict__ o1, int c1, uint32_t *__restrict__ o2, int c2) {
    const int64_t n0 = (int64_t)rows * c0, n1 = (int64_t)rows * c1, n2 = (int64_t)rows * c2, per = n0 + n1 + n2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per * G) return;
    const int r = (int)(gid / per);
    int64_t u = gid % per;
    uint32_t *o;
    int c;
    if (u < n0) { o = o0; c = c0; }
    else if (u < n0 + n1) { u -= n0; o = o1; c = c1; }
    else { u -= n0 + n1; o = o2; c = c2; }
    const int64_t row = u / c, i = u % c;
    o[(row * G + r) * c + i] = stage[(int64_t)r * msg_w + (gid % per)];
}
hipError_t unpack3(const void *stage, size_t msg_bytes, int G, int rows, void *o0, size_t chunk0, void *o1, size_t chunk1, void *o2,
                   size_t chunk2, hipStream_t st) {
    if ((msg_bytes | chunk0 | chunk1 | chunk2) & 3) return hipErrorInvalidValue;
    const int64_t total = (int64_t)rows * (int64_t)((chunk0 + chunk1 + chunk2) >> 2) * G;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(stage),
                       (int64_t)(msg_bytes >> 2), G, rows, reinterpret_cast<uint32_t *>(o0), (int)(chunk0 >> 2), reinterpret_cast<uint32_t *>(o1),
                       (int)(chunk1 >> 2), reinterpret_cast<uint32_t *>(o2), (int)(chunk2 >> 2));
    return hipGetLastError();
}

// The all-gathered Q8_0 operand of a row-split wo / w2 matmul (prefill): message r = rank r's QA16 planes of ITS KBl blocks, packed [q | d | s] for the
// N16 columns of this eval -> the consumer's operand of KB = G KBl blocks: the d / s planes (QA16), the XH16 copy the reference-order GEMM reads
// (q4_layout.h), and -- qout != null -- the q plane for consumers that read QA16.  One launch for what unpack3 + qa16_to_h16 did in two.
// One thread per (column group incl. the second half of the last 32-column tile, global block, column).
__global__ __launch_bounds__(256) void gathered_qa16_to_operand_kernel(const unsigned char *__restrict__ stage, int64_t msg, int G, int KBl, int NGT,
                                                                       int64_t n_threads, uint4 *__restrict__ qout, float *__restrict__ dout,
                                                                       float *__restrict__ sout, uint16_t *__restrict__ xh) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_threads) return;
    const int KB = G * KBl, c16 = (int)(u & 15);
    const int64_t gb = u >> 4;
    const int grp = (int)(gb / KB), B = (int)(gb % KB), r = B / KBl, b = B - r * KBl;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (grp < NGT) {
        const unsigned char *m = stage + (int64_t)r * msg;
        const int64_t cb = ((int64_t)grp * KBl + b) * 16 + c16, nq = (int64_t)NGT * 16 * KBl * 32, nd = (int64_t)NGT * 16 * KBl * 4;
        const uint4 *qp = reinterpret_cast<const uint4 *>(m + cb * 32);
        r0 = qp[0]; r1 = qp[1];
        const int64_t ob = ((int64_t)grp * KB + B) * 16 + c16;
        dout[ob] = *reinterpret_cast<const float *>(m + nq + cb * 4);
        sout[ob] = *reinterpret_cast<const float *>(m + nq + nd + cb * 4);
        if (qout) { qout[2 * ob] = r0; qout[2 * ob + 1] = r1; }
    }
    if (!xh) return;
    const uint32_t dw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const int tile = grp >> 1, i = (grp & 1) * 16 + c16;
    unsigned char *blk = reinterpret_cast<unsigned char *>(xh) + ((int64_t)tile * KB + B) * 2048;
    auto h16_of = [](int v) -> uint32_t { return (uint32_t)__half_as_ushort(__int2half_rn(v)); };
#pragma unroll
    for (int p = 0; p < 4; ++p) {                       // (the 8-byte position p of QA16 holds k-group p ^ (((col >> 3) & 1) << 1), bytes e0,e2,e4,e6,e1,e3,e5,e7)
        const int g = p ^ (((c16 >> 3) & 1) << 1);
        int el[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            el[2 * t] = (int)(int8_t)((dw[2 * p] >> (8 * t)) & 0xFF);
            el[2 * t + 1] = (int)(int8_t)((dw[2 * p + 1] >> (8 * t)) & 0xFF);
        }
        unsigned char *dst = blk + (g >> 1) * 1024 + (g & 1) * 8;
        *reinterpret_cast<uint2 *>(dst + i * 16) = make_uint2(h16_of(el[0]) | (h16_of(el[1]) << 16), h16_of(el[2]) | (h16_of(el[3]) << 16));
        *reinterpret_cast<uint2 *>(dst + (i + 32) * 16) = make_uint2(h16_of(el[4]) | (h16_of(el[5]) << 16), h16_of(el[6]) | (h16_of(el[7]) << 16));
    }
}
hipError_t gathered_qa16_to_operand(const void *stage, size_t msg_bytes, int G, int KBl, int N, const fl_qact &full, bool with_q, bool with_h16, hipStream_t st) {
    const int NGT = fl_roundup(N, 16) / 16, NG2 = (N + 31) / 32 * 2;
    if ((msg_bytes & 15) || (with_h16 && !full.h16) || msg_bytes != (size_t)NGT * 16 * KBl * 40) return hipErrorInvalidValue;
    const int64_t n = (int64_t)(with_h16 ? NG2 : NGT) * G * KBl * 16;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gathered_qa16_to_operand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned char *>(stage),
                       (int64_t)msg_bytes, G, KBl, NGT, n, with_q ? reinterpret_cast<uint4 *>(full.q) : nullptr, full.d, full.s, with_h16 ? full.h16 : nullptr);
    return hipGetLastError();
}

// out[n][r * Ml + j] = tmp[r][n][j] + resid[n][r * Ml + j]: the all-gathered output rows of a row-split matmul and the ggml_add
// that follows wo / w2 (lib/llama.cpp:407, :441) -- one plain f32 add per element, as in the single-GPU epilogue
__global__ void gather_rows_add_kernel(const float *__restrict__ tmp, int G, int N, int Ml, const float *__restrict__ resid, int ldr,
                                       float *__restrict__ out, int ldo) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m4 = Ml >> 2;
    if (gid >= (int64_t)G * N * m4) return;
    const int j = (int)(gid % m4) * 4, n = (int)((gid / m4) % N), r = (int)(gid / ((int64_t)m4 * N));
    const float4 v = *reinterpret_cast<const float4 *>(tmp + ((int64_t)r * N + n) * Ml + j);
    const float4 q = *reinterpret_cast<const float4 *>(resid + (int64_t)n * ldr + (int64_t)r * Ml + j);
    *reinterpret_cast<float4 *>(out + (int64_t)n * ldo + (int64_t)r * Ml + j) =
        make_float4(__fadd_rn(v.x, q.x), __fadd_rn(v.y, q.y), __fadd_rn(v.z, q.z), __fadd_rn(v.w, q.w));
}
hipError_t gather_rows_add(const float *tmp, int G, int N, int Ml, const float *resid, int ldr, float *out, int ldo, hipStream_t st) {
    if ((Ml | ldr | ldo) & 3) return hipErrorInvalidValue;
    const int64_t total = (int64_t)G * N * (Ml >> 2);
    hipLaunchKernelGGL(gather_rows_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tmp, G, N, Ml, resid, ldr, out, ldo);
    return hipGetLastError();
}

}  // namespace fl
