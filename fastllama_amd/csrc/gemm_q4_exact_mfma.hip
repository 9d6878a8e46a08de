// gemm_q4_exact_mfma.hip -- the reference-order ("exact") Q4 x Q8_0 matmul for N >= 2 on the matrix cores.
//
// What must be reproduced (ggml_vec_dot_q4_{0,1}_q8_0, AVX2 branch, /root/reference/lib/ggml.c:2445-2487, :2639-2689): per output
// 8 f32 accumulators, accumulator j taking  acc_j = fma(d_w * d_x, float(sum of the products of elements 4j..4j+3), acc_j)
// block after block, then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ the scalar chain summs = fma(m_w, s_x, summs) for Q4_1].
// So the 8 four-element integer sums of every (output, block) are needed SEPARATELY, as floats -- and that is exactly what the
// K = 4 multi-block MFMA delivers: v_mfma_f32_32x32x4_2b_f16 multiplies two independent 32x4 by 4x32 pairs; with the weights'
// elements 8s..8s+3 / 8s+4..8s+7 in lanes 0-31 / 32-63 of A and the activations' in B, its 32 result registers ARE
// float(lane sum 2s) and float(lane sum 2s+1) of a 32x32 output tile (integers below 2^24: exact in f16 x f16 -> f32).  Four of
// them per block, then the 8 x 16 fma per lane that the reference's order fixes, with dd = rn(d_w * d_x) from the exact
// outer-product MFMA v_mfma_f32_32x32x1_2b_f32 (two blocks at a time).  Measured mix (profiles/r03_ubench_coexec4.txt):
// 210 ns per 32x32 tile and block against 690 ns for v_dot4 + v_cvt + v_fma on the VALU (exact_kernels.hip keeps that form as
// the on-device cross-check, fl_debug_mul_mat_q which = 4).
//
// Workgroup = 4 waves = 128 rows x 32 columns; a wave owns 32 rows (two QW16 row groups) and streams its weights from L2/HBM
// into registers one block pair ahead; the 32-column x 2-block activation tile of the next pair is converted int8 -> f16 once
// per workgroup while it is staged into LDS, in the order the B fragments are read (one ds_read_b128 each; the 8-byte staging stores are
// not conflict-free: SQ_LDS_BANK_CONFLICT ~ 1 cycle per LDS instruction, profiles/r03_gemm_exact_pmc.md).
// Round 4: superseded for N >= 9 by gemm_q4_exact_h16.hip (ready-made f16 fragments by LDS-DMA, 1.45x faster); this kernel remains the
// form for 2 <= N <= 8, the fallback when there is no memory for the H16 copies, and the independent cross-check (fl_debug_mul_mat_q 6).
// Q4_0: the unpacked weights are 16 (nib - 8) and the stored scale is d / 16: fma(rn((d/16) d_x), 16 q, a) rounds the same real
// number as the reference's fma(rn(d d_x), q, a).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "q4_device.h"
#include "q4_kernels.h"

#pragma clang fp contract(off)

namespace fl {

typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v32f __attribute__((ext_vector_type(32)));

// two nibbles of byte `sel`-selected from v -> two f16: Q4_0 (16 (n_lo - 8), 16 (n_hi - 8)) from the stored n ^ 8; Q4_1 (n_lo, n_hi)
template <int TYPE>
__device__ __forceinline__ uint32_t nib2_to_f16(uint32_t v, uint32_t sel) {
    const uint32_t w = __builtin_amdgcn_perm(v, v, sel);                 // (B, 0, B, 0)
    const uint32_t x = (w & 0x00F0000Fu) | 0x64006400u;                  // f16 (1024 + n_lo, 1024 + 16 n_hi)
    const v2h xh = __builtin_bit_cast(v2h, x);
    v2h r;
    if (TYPE == FL_TYPE_Q4_0) r = __builtin_elementwise_fma(xh, v2h{(_Float16)16.0f, (_Float16)1.0f}, v2h{(_Float16)-16512.0f, (_Float16)-1152.0f});
    else r = __builtin_elementwise_fma(xh, v2h{(_Float16)1.0f, (_Float16)0.0625f}, v2h{(_Float16)-1024.0f, (_Float16)-64.0f});
    return __builtin_bit_cast(uint32_t, r);
}

// (q_a, q_b) int8 taken from bytes of (lo, hi) by `sel` -> two f16
__device__ __forceinline__ uint32_t q2_to_f16(uint32_t lo, uint32_t hi, uint32_t sel) {
    const uint32_t w = __builtin_amdgcn_perm(hi, lo, sel) | 0x64006400u;   // (0x6400 | (q_a + 128), 0x6400 | (q_b + 128)): bytes were ^ 0x80
    const v2h r = __builtin_bit_cast(v2h, w) + v2h{(_Float16)-1152.0f, (_Float16)-1152.0f};
    return __builtin_bit_cast(uint32_t, r);
}

template <int TYPE>
__global__ __launch_bounds__(256, 2) void gemm_q4_exact_mfma_kernel(const uint32_t *__restrict__ qs, const float *__restrict__ dW,
                                                                    const float *__restrict__ mW, const int8_t *__restrict__ xq,
                                                                    const float *__restrict__ xd, const float *__restrict__ xs,
                                                                    int N, int M, int groups, int cgroups, int KB,
                                                                    float *__restrict__ y, int ldy, const float *__restrict__ resid,
                                                                    int ldr) {
    constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    // LDS, double buffered, one block PAIR per step: B fragments [2 blocks][h][part][32 cols][16 B] = 4 KB, d_x [2][32], s_x [2][32]
    __shared__ __attribute__((aligned(16))) uint4 bt[2][2][2][2][32];
    __shared__ __attribute__((aligned(16))) float dxs[2][2][32], sxs[2][2][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int grp0 = (blockIdx.x * 4 + wave) * 2;                           // this wave's two row groups
    const int cg0 = blockIdx.y * 2;                                         // the workgroup's two column groups
    const int npairs = (KB + 1) / 2;

    // ---- weights: lane (row i, half h) loads the row's 16 nibble bytes of a block; rows past the tensor re-read the last group
    const int grow = min(grp0 + (i >> 4), groups - 1), r16 = i & 15;
    const uint4 *wq = reinterpret_cast<const uint4 *>(qs) + ((int64_t)grow * KB) * 16 + r16;
    const float *wd = dW + ((int64_t)grow * KB) * 16 + r16;
    const float *wm = Q41 ? mW + ((int64_t)grow * KB) * 16 + r16 : nullptr;
    const bool sw = r16 >= 8;
    const uint32_t selA = h ? 0x0C020C02u : 0x0C000C00u, selB = h ? 0x0C030C03u : 0x0C010C01u;   // bytes 2h, 2h + 1 of a nibble dword
    struct WPair { uint4 w[2]; float d[2], m; };
    auto load_w = [&](int pair) {
        WPair r;
        const int b0 = min(2 * pair, KB - 1), b1 = min(2 * pair + 1, KB - 1);
        r.w[0] = wq[(int64_t)b0 * 16];
        r.w[1] = wq[(int64_t)b1 * 16];
        r.d[0] = h ? 0.f : wd[(int64_t)b0 * 16];                            // dd comes from a K = 2 MFMA whose second k is zero
        r.d[1] = h ? 0.f : wd[(int64_t)b1 * 16];
        r.m = Q41 ? wm[(int64_t)(h ? b1 : b0) * 16] : 0.f;                  // summs: half h takes block 2 pair + h (one chain, in order)
        return r;
    };
    // ---- activations: thread (block u = t >> 7 of the pair, column c = (t >> 2) & 31, position pos = t & 3) converts ONE stored
    //      8-byte k-group (QA16: bytes e0,e2,e4,e6,e1,e3,e5,e7) to f16 and writes it where lanes (c, 0) and (c, 1) read it
    const int su = threadIdx.x >> 7, sc = (threadIdx.x >> 2) & 31, spos = threadIdx.x & 3;
    const int scg = min(cg0 + (sc >> 4), cgroups - 1), sc16 = sc & 15;
    const int sg = spos ^ ((sc16 >> 3) << 1);                               // the k-group stored at this position = MFMA index s
    uint2 xraw;
    float dxr = 0.f, sxr = 0.f;
    auto load_x = [&](int pair) {
        const int bb = 2 * pair + su, b = min(bb, KB - 1);
        const int64_t cb = ((int64_t)scg * KB + b) * 16 + sc16;
        xraw = *reinterpret_cast<const uint2 *>(xq + cb * 32 + spos * 8);
        if (spos == 0) {
            dxr = bb < KB ? xd[cb] : 0.f;                                   // a block past K: dd = 0, nothing is added
            sxr = Q41 && bb < KB ? xs[cb] : 0.f;
        }
    };
    auto store_x = [&](int buf) {
        const uint32_t lo = xraw.x ^ 0x80808080u, hi = xraw.y ^ 0x80808080u;
        // elements 0..3 of the group -> half 0, elements 4..7 -> half 1; fragment slot: part = g >> 1, 8-byte half g & 1
        const uint2 f0 = make_uint2(q2_to_f16(lo, hi, 0x0C040C00u), q2_to_f16(lo, hi, 0x0C050C01u));
        const uint2 f1 = make_uint2(q2_to_f16(lo, hi, 0x0C060C02u), q2_to_f16(lo, hi, 0x0C070C03u));
        reinterpret_cast<uint2 *>(&bt[buf][su][0][sg >> 1][sc])[sg & 1] = f0;
        reinterpret_cast<uint2 *>(&bt[buf][su][1][sg >> 1][sc])[sg & 1] = f1;
        if (spos == 0) {
            dxs[buf][su][sc] = dxr;
            if (Q41) sxs[buf][su][sc] = sxr;
        }
    };

    v16f acc[8], summs;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) summs[e] = 0.f;
    const v32f zero32 = {};
    const v16f zero16 = {};

    WPair wc = load_w(0), w1 = load_w(npairs > 1 ? 1 : 0);                  // the weights run two pairs ahead of the MFMAs (MALL / HBM latency)
    load_x(0);
    store_x(0);
    __syncthreads();
#pragma unroll 1
    for (int pair = 0; pair < npairs; ++pair) {
        const int buf = pair & 1;
        const bool more = pair + 1 < npairs;
        const WPair wn = load_w(min(pair + 2, npairs - 1));                 // (unconditional: the compiler can count the loads in flight)
        load_x(more ? pair + 1 : pair);
        if (Q41) summs = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.m, sxs[buf][h][i], summs, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            // dd = rn(d_w * d_x) of this block as an exact outer product: k = 0 carries (d_w, d_x), k = 1 is (0, 0)
            const v16f P = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.d[ub], h ? 0.f : dxs[buf][ub][i], zero16, 0, 0, 0);
            uint4 wr = wc.w[ub];
            if (sw) wr = make_uint4(wr.z, wr.w, wr.x, wr.y);                // dword position p holds k-group p ^ 2 for rows 8..15
            if (TYPE == FL_TYPE_Q4_0) wr = make_uint4(wr.x ^ 0x88888888u, wr.y ^ 0x88888888u, wr.z ^ 0x88888888u, wr.w ^ 0x88888888u);   // stored nib ^ 8 -> nib
            uint32_t wdw[4] = {wr.x, wr.y, wr.z, wr.w};
            const uint4 b01 = bt[buf][ub][h][0][i], b23 = bt[buf][ub][h][1][i];
            const uint32_t bw[8] = {b01.x, b01.y, b01.z, b01.w, b23.x, b23.y, b23.z, b23.w};
            uint2 af = make_uint2(nib2_to_f16<TYPE>(wdw[0], selA), nib2_to_f16<TYPE>(wdw[0], selB));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint2 bf = make_uint2(bw[2 * s], bw[2 * s + 1]);
                const v32f D = __builtin_amdgcn_mfma_f32_32x32x4f16(__builtin_bit_cast(v4h, af), __builtin_bit_cast(v4h, bf), zero32, 0, 0, 0);
                // the next group's A fragment is unpacked in the shadow of this MFMA (its result cannot be read for ~20 cycles anyway)
                uint2 afn = af;
                if (s < 3) afn = make_uint2(nib2_to_f16<TYPE>(wdw[s + 1], selA), nib2_to_f16<TYPE>(wdw[s + 1], selB));
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[2 * s][e] = __builtin_fmaf(P[e], D[e], acc[2 * s][e]);
                    acc[2 * s + 1][e] = __builtin_fmaf(P[e], D[16 + e], acc[2 * s + 1][e]);
                }
                // Pin the order: the 32 fma above must be issued before the NEXT group's MFMA (whose A operand passes through this
                // statement) -- left alone, the compiler emits the four MFMAs first and keeps 4 x 32 result registers alive.
                asm volatile("" : "+v"(acc[2 * s]), "+v"(acc[2 * s + 1]), "+v"(afn));
                af = afn;
            }
        }
        if (more) store_x(buf ^ 1);
        __syncthreads();
        wc = w1;
        w1 = wn;
    }

    // ---- ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ summs] (+ resid): C layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h ----
    const int n = blockIdx.y * 32 + i;
    if (n >= N) return;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        float4 o;
        float *op = &o.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = 4 * e4 + k;
            float v = __fadd_rn(__fadd_rn(__fadd_rn(acc[0][e], acc[4][e]), __fadd_rn(acc[2][e], acc[6][e])),
                                __fadd_rn(__fadd_rn(acc[1][e], acc[5][e]), __fadd_rn(acc[3][e], acc[7][e])));
            if (Q41) v = __fadd_rn(v, summs[e]);
            op[k] = v;
        }
        const int row = (blockIdx.x * 4 + wave) * 32 + 8 * e4 + 4 * h;      // rows row .. row + 3
        if (row >= M) continue;
        float *yp = y + (int64_t)n * ldy + row;
        const float *rp = resid ? resid + (int64_t)n * ldr + row : nullptr;
        if (row + 3 < M && (ldy & 3) == 0 && (!resid || (ldr & 3) == 0)) {
            if (rp) {
                const float4 rr = *reinterpret_cast<const float4 *>(rp);
                o.x = __fadd_rn(o.x, rr.x); o.y = __fadd_rn(o.y, rr.y); o.z = __fadd_rn(o.z, rr.z); o.w = __fadd_rn(o.w, rr.w);
            }
            *reinterpret_cast<float4 *>(yp) = o;
        } else {
            for (int k = 0; k < 4 && row + k < M; ++k) yp[k] = rp ? __fadd_rn(op[k], rp[k]) : op[k];
        }
    }
}

hipError_t gemm_q4_exact_mfma(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid,
                              int ldr) {
    if (N < 1) return hipErrorInvalidValue;
    const int groups = W.M16 / 16, cgroups = (N + 15) / 16;
    const dim3 grid((groups + 7) / 8, (N + 31) / 32);
    if (W.type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(gemm_q4_exact_mfma_kernel<FL_TYPE_Q4_0>, grid, dim3(256), 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s, N, W.M,
                           groups, cgroups, W.KB, y, ldy, resid, ldr);
    else
        hipLaunchKernelGGL(gemm_q4_exact_mfma_kernel<FL_TYPE_Q4_1>, grid, dim3(256), 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s, N, W.M,
                           groups, cgroups, W.KB, y, ldy, resid, ldr);
    return hipGetLastError();
}

}  // namespace fl
