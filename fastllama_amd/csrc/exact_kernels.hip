// exact_kernels.hip -- the REFERENCE-ORDER ("exact") forms of the two matmul families of Model::eval.
//
// The fast kernels (gemm_q4_mfma32.hip, gemv_q4_kernel, the MFMA attention) compute every integer block dot exactly but
// add the per-block f32 terms in an order of their own.  Over 32 layers that 1e-7 freedom flips a handful of Q8_0 /
// fp16-table roundings and the logits end up 1e-2 away from the reference.  The kernels here keep the
// order of the reference's x86 build bit for bit (all file:line into /root/reference):
//
//   ggml_vec_dot_q4_0_q8_0 / _q4_1_q8_0, AVX2 branch (lib/ggml.c:2445-2487, :2639-2689)
//       8 f32 lane accumulators per output; lane j takes  acc_j = fma(d_w*d_x, float(sum of elements 4j..4j+3), acc_j)
//       block after block in K order; result = ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7))  [+ summs for Q4_1, a scalar
//       fma chain  summs = fma(m_w, s_x, summs)  in the same block order]
//   ggml_vec_dot_f32 as compiled into ggml_compute_forward_mul_mat_f32 (lib/ggml.c:2295-2330 at :7662): the attention
//       matmuls K.Q and V.P -- 4 x 8 FMA lanes over 32-element steps, the fixed reduction tree, and the n % 32 leftovers
//       as gcc vectorised them: chunks of 8, then one chunk of 4 elements as ROUNDED products added one by one in order,
//       and only the last n % 4 elements as scalar FMAs (disassembly of the reference build)
//
// v_dot4_i32_i8 IS the 4-adjacent-element sum of one AVX2 lane (maddubs + madd), so the lane structure maps 1:1 onto
// the hardware.  Work decomposition: a LANE owns (weight row, k-group of 8 elements) = two of the eight accumulators of
// every output of that row and walks the blocks of the row in order -- the fma chains never cross lanes; the final
// 8-term sum is three DPP adds inside a quad.
//
// Everything else of the eval (norms, rope, soft_max, SiLU table, Q8_0 quantization) was already bit-exact given equal
// inputs, so with these kernels the logits are the reference's, bit for bit (tests/test_parity_7b_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "eval_kernels.h"
#include "q4_device.h"
#include "q4_kernels.h"

#pragma clang fp contract(off)

namespace fl {

// bytes of the stored operand dwords: lo = elements (0,2,4,6), hi = elements (1,3,5,7) of a k-group (q4_layout.h).
// lane sum 2g   = elements 0..3 = (lo.b0, hi.b0, lo.b1, hi.b1);  lane sum 2g+1 = elements 4..7 = (lo.b2, hi.b2, lo.b3, hi.b3)
__device__ __forceinline__ uint32_t perm_a(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_perm(hi, lo, 0x05010400u); }
__device__ __forceinline__ uint32_t perm_b(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_perm(hi, lo, 0x07030602u); }

template <int TYPE>
__device__ __forceinline__ void unpack_lanes(uint32_t v, uint32_t &wa, uint32_t &wb) {
    uint32_t lo, hi;
    unpack_nibbles<TYPE>(v, lo, hi);     // Q4_0: 16*(nib-8) with d/16 stored -- fma(d/16, 16q, a) == fma(d, q, a) exactly
    wa = perm_a(hi, lo);
    wb = perm_b(hi, lo);
}

// ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) across the quad of lanes that holds a row: lane position p holds k-group
// g = p ^ sw (sw in {0, 2}, q4_layout.h) = accumulators (2g, 2g+1) in (e, o).  g ^ 2 <-> p ^ 2 and g ^ 1 <-> p ^ 1, and
// f32 addition is commutative, so every lane of the quad ends with the same bits.
__device__ __forceinline__ float hsum8_quad(float e, float o) {
    e = __fadd_rn(e, dpp_f32<DPP_XOR2>(e));      // a0+a4 | a2+a6
    o = __fadd_rn(o, dpp_f32<DPP_XOR2>(o));      // a1+a5 | a3+a7
    e = __fadd_rn(e, dpp_f32<DPP_XOR1>(e));      // (a0+a4)+(a2+a6)
    o = __fadd_rn(o, dpp_f32<DPP_XOR1>(o));      // (a1+a5)+(a3+a7)
    return __fadd_rn(e, o);
}

// ------------------------------------------------------------------------------------------------
// N <= 8: one wave per 16-row group, lane = 4 * row + dword position.  HBM-bound like gemv_q4_kernel: a wave-load is one
// block of the group = 256 contiguous bytes, U of them in flight per lane; the activation (QA1, a few KB) comes from L1/L2.
// ------------------------------------------------------------------------------------------------
template <int TYPE, int NC, int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemv_q4_exact_kernel(const uint32_t *__restrict__ qs, const float *__restrict__ dW,
                                                                const float *__restrict__ mW, const int8_t *__restrict__ xq,
                                                                const float *__restrict__ xd, const float *__restrict__ xs,
                                                                int N, int M, int groups, int KB, float *__restrict__ y, int ldy,
                                                                const float *__restrict__ resid, int ldr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = blockIdx.x * NW + wave;
    if (grp >= groups) return;
    const int r = lane >> 2, p = lane & 3, g = p ^ (((r >> 3) & 1) << 1);
    const uint32_t *wq = qs + (int64_t)grp * KB * 64 + lane;
    const float *wd = dW + (int64_t)grp * KB * 16 + r;
    const float *wm = TYPE == FL_TYPE_Q4_1 ? mW + (int64_t)grp * KB * 16 + r : nullptr;
    float acc[NC][2], summs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = summs[c] = 0.f;
    uint32_t w[U];
    float dw[U], mw[U];
    auto load = [&](int b0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = min(b0 + u, KB - 1);
            w[u] = wq[(int64_t)b * 64];
            dw[u] = wd[(int64_t)b * 16];
            mw[u] = TYPE == FL_TYPE_Q4_1 ? wm[(int64_t)b * 16] : 0.f;
        }
    };
    load(0);
    for (int b0 = 0; b0 < KB; b0 += U) {
        uint32_t wa[U], wb[U];
        float dcur[U], mcur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unpack_lanes<TYPE>(w[u], wa[u], wb[u]);
            dcur[u] = dw[u];
            mcur[u] = mw[u];
        }
        if (b0 + U < KB) load(b0 + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u;
            if (b >= KB) break;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int n = c < N ? c : 0;
                const int64_t xb = (int64_t)n * KB + b;
                const uint2 xv = *reinterpret_cast<const uint2 *>(xq + xb * 32 + g * 8);
                const float dd = __fmul_rn(dcur[u], xd[xb]);
                const int ia = __builtin_amdgcn_sdot4((int)wa[u], (int)perm_a(xv.y, xv.x), 0, false);
                const int ib = __builtin_amdgcn_sdot4((int)wb[u], (int)perm_b(xv.y, xv.x), 0, false);
                acc[c][0] = __fmaf_rn(dd, (float)ia, acc[c][0]);
                acc[c][1] = __fmaf_rn(dd, (float)ib, acc[c][1]);
                if (TYPE == FL_TYPE_Q4_1) summs[c] = __fmaf_rn(mcur[u], xs[xb], summs[c]);
            }
        }
    }
    const int row = grp * 16 + r;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float v = hsum8_quad(acc[c][0], acc[c][1]);
        if (TYPE == FL_TYPE_Q4_1) v = __fadd_rn(v, summs[c]);
        if (p == 0 && c < N && row < M) {
            if (resid) v = __fadd_rn(v, resid[(int64_t)c * ldr + row]);
            y[(int64_t)c * ldy + row] = v;
        }
    }
}

template <int TYPE>
static hipError_t launch_gemv_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                                    const float *resid, int ldr) {
    const int groups = W.M16 / 16;
    constexpr int NW = 2;
    const dim3 grid((groups + NW - 1) / NW);
#define FL_GE(NC, U)                                                                                                      \
    hipLaunchKernelGGL((gemv_q4_exact_kernel<TYPE, NC, NW, U>), grid, dim3(64 * NW), 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s, \
                       N, W.M, groups, W.KB, y, ldy, resid, ldr)
    if (N == 1) FL_GE(1, 16);
    else if (N == 2) FL_GE(2, 8);
    else if (N <= 4) FL_GE(4, 8);
    else FL_GE(8, 4);
#undef FL_GE
    return hipGetLastError();
}

hipError_t gemv_q4_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid,
                         int ldr) {
    if (N < 1 || N > 8) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_gemv_exact<FL_TYPE_Q4_0>(W, xq, N, y, ldy, st, resid, ldr)
                                  : launch_gemv_exact<FL_TYPE_Q4_1>(W, xq, N, y, ldy, st, resid, ldr);
}

// ------------------------------------------------------------------------------------------------
// N >= 9: a wave owns RG = 2 row groups (32 rows) x one QA16 column group (16 columns); a workgroup = 4 waves = 128 rows
// of the same column group.  The activation tile of a K-step (16 columns x KS blocks, bytes already in lane-sum order)
// is staged through LDS and read as broadcasts (the 16 rows of a group read the same 8 bytes); the weights stream from
// L2/HBM into registers one K-step ahead.  VALU-bound by construction: per (row, column, block) 8 v_dot4 + 8 cvt + 8
// fma are the reference's own operation count.
// ------------------------------------------------------------------------------------------------
template <int TYPE>
__global__ __launch_bounds__(256) void gemm_q4_exact_kernel(const uint32_t *__restrict__ qs, const float *__restrict__ dW,
                                                            const float *__restrict__ mW, const int8_t *__restrict__ xq,
                                                            const float *__restrict__ xd, const float *__restrict__ xs,
                                                            int N, int M, int groups, int KB, float *__restrict__ y, int ldy,
                                                            const float *__restrict__ resid, int ldr) {
    constexpr int KS = 4, RG = 2, TN = 16, TNP = TN + 2;                // +2: the four k-groups of a column fall into distinct LDS banks
    __shared__ __attribute__((aligned(16))) uint2 xt[KS][4][TNP];       // [block][k-group][column]: (lane-sum a, lane-sum b) dwords
    __shared__ __attribute__((aligned(16))) float dxs[KS][TN], sxs[KS][TN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = blockIdx.y;                                          // column group
    const int grp0 = (blockIdx.x * 4 + wave) * RG;
    const int r = lane >> 2, p = lane & 3, g = p ^ (((r >> 3) & 1) << 1);
    float acc[RG][TN][2], summs[RG][TN];
#pragma unroll
    for (int h = 0; h < RG; ++h)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[h][c][0] = acc[h][c][1] = summs[h][c] = 0.f;
    uint32_t w[RG][KS];
    float dw[RG][KS], mw[RG][KS];
    auto load_w = [&](int b0) {
#pragma unroll
        for (int h = 0; h < RG; ++h) {
            const int grp = min(grp0 + h, groups - 1);
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const int64_t gb = (int64_t)grp * KB + min(b0 + u, KB - 1);
                w[h][u] = qs[gb * 64 + lane];
                dw[h][u] = dW[gb * 16 + r];
                mw[h][u] = TYPE == FL_TYPE_Q4_1 ? mW[gb * 16 + r] : 0.f;
            }
        }
    };
    // staging: thread t < 256 = (block u = t >> 6, column c = (t >> 2) & 15, position pp = t & 3) moves one 8-byte k-group
    const int su = threadIdx.x >> 6, sc = (threadIdx.x >> 2) & 15, sp = threadIdx.x & 3;
    const int sg = sp ^ (((sc >> 3) & 1) << 1);                         // the k-group stored at position sp of column sc
    uint2 xnext;
    float dxnext = 0.f, sxnext = 0.f;
    auto load_x = [&](int b0) {
        const int64_t cb = ((int64_t)cg * KB + min(b0 + su, KB - 1)) * 16 + sc;
        xnext = *reinterpret_cast<const uint2 *>(xq + cb * 32 + sp * 8);
        if (sp == 0) dxnext = xd[cb];
        if (sp == 1 && TYPE == FL_TYPE_Q4_1) sxnext = xs[cb];
    };
    load_w(0);
    load_x(0);
    for (int b0 = 0; b0 < KB; b0 += KS) {
        __syncthreads();                                                // the previous step's reads are done
        xt[su][sg][sc] = make_uint2(perm_a(xnext.y, xnext.x), perm_b(xnext.y, xnext.x));
        if (sp == 0) dxs[su][sc] = dxnext;
        if (sp == 1 && TYPE == FL_TYPE_Q4_1) sxs[su][sc] = sxnext;
        __syncthreads();
        if (b0 + KS < KB) load_x(b0 + KS);
        uint32_t wa[RG][KS], wb[RG][KS];
        float dcur[RG][KS], mcur[RG][KS];
#pragma unroll
        for (int h = 0; h < RG; ++h)
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                unpack_lanes<TYPE>(w[h][u], wa[h][u], wb[h][u]);
                dcur[h][u] = dw[h][u];
                mcur[h][u] = mw[h][u];
            }
        if (b0 + KS < KB) load_w(b0 + KS);
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            if (b0 + u >= KB) break;
#pragma unroll
            for (int c4 = 0; c4 < TN; c4 += 4) {
                const float4 dx4 = *reinterpret_cast<const float4 *>(&dxs[u][c4]);
                float4 sx4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (TYPE == FL_TYPE_Q4_1) sx4 = *reinterpret_cast<const float4 *>(&sxs[u][c4]);
                const float dxa[4] = {dx4.x, dx4.y, dx4.z, dx4.w}, sxa[4] = {sx4.x, sx4.y, sx4.z, sx4.w};
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const int c = c4 + ci;
                    const uint2 xv = xt[u][g][c];
#pragma unroll
                    for (int h = 0; h < RG; ++h) {
                        const float dd = __fmul_rn(dcur[h][u], dxa[ci]);
                        const int ia = __builtin_amdgcn_sdot4((int)wa[h][u], (int)xv.x, 0, false);
                        const int ib = __builtin_amdgcn_sdot4((int)wb[h][u], (int)xv.y, 0, false);
                        acc[h][c][0] = __fmaf_rn(dd, (float)ia, acc[h][c][0]);
                        acc[h][c][1] = __fmaf_rn(dd, (float)ib, acc[h][c][1]);
                        if (TYPE == FL_TYPE_Q4_1) summs[h][c] = __fmaf_rn(mcur[h][u], sxa[ci], summs[h][c]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < RG; ++h) {
        const int row = (grp0 + h) * 16 + r;
#pragma unroll
        for (int c = 0; c < TN; ++c) {
            float v = hsum8_quad(acc[h][c][0], acc[h][c][1]);
            if (TYPE == FL_TYPE_Q4_1) v = __fadd_rn(v, summs[h][c]);
            const int n = cg * 16 + c;
            if (p == (c & 3) && grp0 + h < groups && row < M && n < N) {      // the quad's four lanes share the stores
                if (resid) v = __fadd_rn(v, resid[(int64_t)n * ldr + row]);
                y[(int64_t)n * ldy + row] = v;
            }
        }
    }
}

hipError_t gemm_q4_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid,
                         int ldr) {
    if (N < 1) return hipErrorInvalidValue;
    const int groups = W.M16 / 16;
    const dim3 grid((groups + 7) / 8, (N + 15) / 16);
    if (W.type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(gemm_q4_exact_kernel<FL_TYPE_Q4_0>, grid, dim3(256), 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s, N, W.M,
                           groups, W.KB, y, ldy, resid, ldr);
    else
        hipLaunchKernelGGL(gemm_q4_exact_kernel<FL_TYPE_Q4_1>, grid, dim3(256), 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s, N, W.M,
                           groups, W.KB, y, ldy, resid, ldr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// f32 x f32 matmuls of the attention in ggml_vec_dot_f32's order.  C[z][m][n] = alpha * dot(K, A[z] row m, B[z] row n),
// the interface of gemm_f32_abt (eval_kernels.hip).  A half-wave (32 lanes = the 4 x 8 AVX lanes: lane 8j + l is element l
// of sum[j]) computes one dot; the reduction tree is the macro's (lib/ggml.c:1921-1936):
//   sum0 += sum1; sum2 += sum3; sum0 += sum2;  t = lo128 + hi128;  hadd; hadd
// causal_mode 1 (scores): columns n > n_past + m are never read by soft_max -> skipped.
// causal_mode 2 (KQV): A = probabilities, zero beyond n_past + m: 32-element steps made of zeros only are skipped
//   (fma(0, v, s) == s for finite v; the accumulators start at +0 and can never become -0).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot_f32_ref_order(const float *__restrict__ x, const float *__restrict__ y, int n, int nbody,
                                                   int hl) {
    // hl = lane within the half-wave; every lane of the half-wave must call this (shuffles)
    const int np = n & ~31;
    float s = 0.f;
    for (int i = 0; i < min(np, nbody); i += 32) s = __fmaf_rn(x[i + hl], y[i + hl], s);
    s = __fadd_rn(s, __shfl_down(s, 8, 32));        // lanes 0-7: sum0+sum1, lanes 16-23: sum2+sum3
    s = __fadd_rn(s, __shfl_down(s, 16, 32));       // lanes 0-7: (sum0+sum1)+(sum2+sum3)
    s = __fadd_rn(s, __shfl_down(s, 4, 32));        // lanes 0-3: lo + hi
    s = __fadd_rn(s, __shfl_down(s, 1, 32));        // lane 0: t0+t1, lane 2: t2+t3
    s = __fadd_rn(s, __shfl_down(s, 2, 32));        // lane 0: (t0+t1)+(t2+t3)
    if (hl == 0 && np < n) {                         // leftovers, in order (gcc's vectorisation of the scalar loop)
        int i = np;
        for (; i + 8 <= n; i += 8)
            for (int l = 0; l < 8; ++l) s = __fadd_rn(s, __fmul_rn(x[i + l], y[i + l]));
        if (n - i >= 4) {
            for (int l = 0; l < 4; ++l) s = __fadd_rn(s, __fmul_rn(x[i + l], y[i + l]));
            i += 4;
        }
        for (; i < n; ++i) s = __fmaf_rn(x[i], y[i], s);
    }
    return s;   // valid on lane 0 of the half-wave
}

__global__ __launch_bounds__(256) void dot_f32_abt_exact_kernel(const float *__restrict__ A, int lda, int64_t sAz,
                                                                const float *__restrict__ B, int ldb, int64_t sBz,
                                                                float *__restrict__ C, int ldc, int64_t sCz, int M, int Nn,
                                                                int K, float alpha, int causal_mode, int n_past,
                                                                const int *__restrict__ dyn_past) {
    if (dyn_past) {
        n_past = *dyn_past;
        if (causal_mode == 1) Nn = n_past + M;
        if (causal_mode == 2) K = n_past + M;
    }
    const int z = blockIdx.z, m = blockIdx.y;
    const int hw = threadIdx.x >> 5, hl = threadIdx.x & 31;
    const int n = blockIdx.x * 8 + hw;
    if (n >= Nn) return;
    if (causal_mode == 1 && n > n_past + m) return;
    const float *pa = A + z * sAz + (int64_t)m * lda;
    const float *pb = B + z * sBz + (int64_t)n * ldb;
    const int nbody = causal_mode == 2 ? ((min(K, n_past + m + 1) + 31) & ~31) : K;
    const float s = dot_f32_ref_order(pa, pb, K, nbody, hl);
    if (hl == 0) C[z * sCz + (int64_t)m * ldc + n] = __fmul_rn(s, alpha);
}

hipError_t dot_f32_abt_exact(const float *A, int lda, int64_t sAz, const float *B, int ldb, int64_t sBz, float *C, int ldc,
                             int64_t sCz, int M, int Nn, int K, int batch, float alpha, int causal_mode, int n_past,
                             hipStream_t st, const int *dyn_past, int nn_max) {
    const dim3 grid(((dyn_past && causal_mode == 1 ? nn_max : Nn) + 7) / 8, M, batch);
    hipLaunchKernelGGL(dot_f32_abt_exact_kernel, grid, dim3(256), 0, st, A, lda, sAz, B, ldb, sBz, C, ldc, sCz, M, Nn, K, alpha,
                       causal_mode, n_past, dyn_past);
    return hipGetLastError();
}

}  // namespace fl
