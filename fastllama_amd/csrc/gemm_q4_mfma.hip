// gemm_q4_mfma.hip -- prefill path (N >= 9) of ggml_compute_forward_mul_mat_q_f32
// (/root/reference/lib/ggml.c:7928-8176, COMPUTE phase :8127-8163): exact-integer MFMA GEMM for gfx950.
//
//   y[n][m] = sum_b  (d_w[m,b] * d_x[n,b]) * isum[m,n,b]   (+ m_w[m,b] * s_x[n,b] for Q4_1)
//   isum[m,n,b] = sum_{i<32} w_i * q_i  -- the int dot of ggml_vec_dot_q4_{0,1}_q8_0 (:2368, :2561)
//
// One MFMA K-step is exactly one 32-element quant block: v_mfma_i32_16x16x32_i8 (A = 16 W rows,
// B = 16 activation columns) produces the 16x16 block dots exactly.  The per-block scale product cannot
// be folded into integer operands, so the f32 scale-accumulate is a VALU epilogue of 1/32 of the MACs
// -- which on gfx950 is the co-critical resource (12 lane-ops per 16-cycle MFMA).  Design points:
//
//   * int -> float without v_cvt: the MFMA's C input is the constant 0x4B400000 (= 1.5*2^23 as f32
//     bits); D = magic + isum, reinterpreted as f32, IS 12582912 + isum exactly (|isum| < 2^22), so one
//     (packable) v_sub_f32 replaces v_cvt_f32_i32.  Epilogue = pk_add, pk_mul, pk_fma per 2 outputs.
//   * the MFMA of tile t+1 is issued before the epilogue of tile t (software pipeline in one wave).
//   * QW16 / QA16 make every LDS fill a linear 16-byte copy -> global_load_lds (no staging VGPRs),
//     two LDS stages; fragment reads are bank-conflict free by construction (q4_layout.h).
//   * workgroup tile (32*TM) x 128, 4 waves as 2 x 2, wave tile (16*TM) x 64; K-step = 2 blocks.
//   * XCD-aware bijective tile order: the N-tiles that share a W row panel run on one XCD's L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "q4_device.h"
#include "q4_kernels.h"

namespace fl {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ const uint4 fl_zero_chunk[1] = {{0u, 0u, 0u, 0u}};  // source of out-of-range LDS fills

constexpr int GM_KS = 2;  // quant blocks per K-step

template <int TYPE, int TM>
struct GemmCfg {
    static constexpr int MG = 2 * TM;                        // row groups per workgroup tile
    static constexpr int A_BYTES = MG * GM_KS * 256;         // packed nibbles
    static constexpr int B_BYTES = 8 * GM_KS * 512;          // int8 activations
    static constexpr int SW_BYTES = 1024;                    // d_w plane (MG*KS*64 <= 1024), padded to 1 KiB
    static constexpr int SX_BYTES = 1024;                    // d_x plane (8*KS*64 = 1024)
    static constexpr int OFF_B = A_BYTES;
    static constexpr int OFF_DW = OFF_B + B_BYTES;
    static constexpr int OFF_DX = OFF_DW + SW_BYTES;
    static constexpr int OFF_MW = OFF_DX + SX_BYTES;         // Q4_1 only
    static constexpr int OFF_SX = OFF_MW + SW_BYTES;
    static constexpr int STAGE = TYPE == FL_TYPE_Q4_1 ? OFF_SX + SX_BYTES : OFF_MW;
};

template <int TYPE, int TM>
__global__ __launch_bounds__(256, 2) void gemm_q4_mfma_kernel(
    const uint4 *__restrict__ qs, const float *__restrict__ dW, const float *__restrict__ mW,
    const int8_t *__restrict__ xq, const float *__restrict__ xd, const float *__restrict__ xs, int N, int M,
    int MGT /* row groups total */, int NGT /* col groups total */, int KB, float *__restrict__ y, int ldy) {
    using Cfg = GemmCfg<TYPE, TM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- XCD-aware bijective remap of the tile id (cdna guide T1): block b runs on XCD b%8 ----
    const int tiles_m = (MGT + Cfg::MG - 1) / Cfg::MG, tiles_n = (NGT + 7) >> 3;
    int bid = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int mg0 = tm * Cfg::MG, ng0 = tn * 8;

    // ---- LDS fill: every wave issues its share of 1-KiB global_load_lds pieces ----
    //   A : MG*KS*16 chunks of 16 B  (TM=4: 256 = 4 pieces, TM=2: 128 = 2 pieces)
    //   B : 8*KS*32 = 512 chunks = 8 pieces;  scale planes: 1 piece each (padded)
    auto fill = [&](int st, int kb0) {
        unsigned char *base = smem + st * Cfg::STAGE;
        const glb_void *zsrc = (glb_void *)fl_zero_chunk;
        // A pieces: piece p covers chunks [64p, 64p+64): group gi = c / (KS*16), e = c % (KS*16)
        constexpr int A_PIECES = Cfg::A_BYTES / 1024;
        for (int p = wave; p < A_PIECES; p += 4) {
            const int c = p * 64 + lane;
            const int gi = c / (GM_KS * 16), e = c % (GM_KS * 16);
            const int g = mg0 + gi, b = kb0 + e / 16;
            const glb_void *src = (g < MGT && b < KB) ? (glb_void *)(qs + ((int64_t)g * KB + kb0) * 16 + e) : zsrc;
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(base + p * 1024), 16, 0, 0);
        }
        // B pieces: one piece per column group (KS*512 = 1024 B)
        for (int p = wave * 2; p < wave * 2 + 2; ++p) {
            const int g = ng0 + p, b = kb0 + lane / 32;
            const glb_void *src = (g < NGT && b < KB)
                                      ? (glb_void *)(xq + (((int64_t)g * KB + kb0) * 16) * 32 + lane * 16)
                                      : zsrc;
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(base + Cfg::OFF_B + p * 1024), 16, 0, 0);
        }
        // scale planes: chunk c: group gi = c / (KS*4), e = c % (KS*4): block e/4, 4 rows each
        {
            const int gi = lane / (GM_KS * 4), e = lane % (GM_KS * 4);
            const int b = kb0 + e / 4;
            if (wave == 0 || (TYPE == FL_TYPE_Q4_1 && wave == 2)) {
                const int g = mg0 + gi;
                const float *pl = wave == 0 ? dW : mW;
                const glb_void *src =
                    (gi < Cfg::MG && g < MGT && b < KB) ? (glb_void *)(pl + ((int64_t)g * KB + kb0) * 16 + e * 4) : zsrc;
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(base + (wave == 0 ? Cfg::OFF_DW : Cfg::OFF_MW)), 16, 0, 0);
            } else if (wave == 1 || (TYPE == FL_TYPE_Q4_1 && wave == 3)) {
                const int g = ng0 + gi;
                const float *pl = wave == 1 ? xd : xs;
                const glb_void *src =
                    (g < NGT && b < KB) ? (glb_void *)(pl + ((int64_t)g * KB + kb0) * 16 + e * 4) : zsrc;
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(base + (wave == 1 ? Cfg::OFF_DX : Cfg::OFF_SX)), 16, 0, 0);
            }
        }
    };

    v4f acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    const v4i magic = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};  // 12582912.0f = 1.5 * 2^23
    const int apos = qw16_pos(l15, lg);

    const int nsteps = (KB + GM_KS - 1) / GM_KS;
    fill(0, 0);
    __syncthreads();

    for (int t = 0; t < nsteps; ++t) {
        const int cur = t & 1;
        if (t + 1 < nsteps) fill(cur ^ 1, (t + 1) * GM_KS);

        const unsigned char *base = smem + cur * Cfg::STAGE;
        const uint32_t *sa = reinterpret_cast<const uint32_t *>(base);
        const unsigned char *sb = base + Cfg::OFF_B;
        const float *sdw = reinterpret_cast<const float *>(base + Cfg::OFF_DW);
        const float *sdx = reinterpret_cast<const float *>(base + Cfg::OFF_DX);
        const float *smw = reinterpret_cast<const float *>(base + Cfg::OFF_MW);
        const float *ssx = reinterpret_cast<const float *>(base + Cfg::OFF_SX);

#pragma unroll
        for (int b = 0; b < GM_KS; ++b) {
            long afrag[TM], bfrag[4];
            v4f dwv[TM], mwv[TM];
            float dxv[4], sxv[4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int g = wm * TM + i;
                const uint32_t v = sa[((g * GM_KS + b) * 16 + l15) * 4 + apos];
                uint32_t lo, hi;
                unpack_nibbles<TYPE>(v, lo, hi);
                afrag[i] = (long)(((uint64_t)hi << 32) | lo);
                dwv[i] = *reinterpret_cast<const v4f *>(sdw + (g * GM_KS + b) * 16 + lg * 4);
                if (TYPE == FL_TYPE_Q4_1) mwv[i] = *reinterpret_cast<const v4f *>(smw + (g * GM_KS + b) * 16 + lg * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = wn * 4 + j;
                bfrag[j] = *reinterpret_cast<const long *>(sb + ((g * GM_KS + b) * 16 + l15) * 32 + apos * 8);
                dxv[j] = sdx[(g * GM_KS + b) * 16 + l15];
                if (TYPE == FL_TYPE_Q4_1) sxv[j] = ssx[(g * GM_KS + b) * 16 + l15];
            }
            // software pipeline inside one wave: MFMA(tile t+1) is issued, THEN the VALU scales tile t, so the
            // 6 packed VALU ops run under the 16-cycle MFMA.  hipcc's scheduler otherwise re-serialises this
            // (MFMA -> s_nop -> its own epilogue), hence the sched_barrier(0) pins between the two halves.
            v4i r0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(afrag[0], bfrag[0], magic, 0, 0, 0);
            v4i r1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < TM * 4; ++tt) {
                const int i = tt >> 2, j = tt & 3;
                if (tt + 1 < TM * 4) {
                    const v4i rn = __builtin_amdgcn_mfma_i32_16x16x32_i8(afrag[(tt + 1) >> 2], bfrag[(tt + 1) & 3], magic, 0, 0, 0);
                    if (tt & 1) r0 = rn; else r1 = rn;
                }
                __builtin_amdgcn_sched_barrier(0);
                const v4f f = __builtin_bit_cast(v4f, (tt & 1) ? r1 : r0) - 12582912.0f;  // exact: float(isum)
                const v4f p = dwv[i] * dxv[j];                                             // d_w*d_x (ggml.c:2452)
                acc[i][j] = __builtin_elementwise_fma(f, p, acc[i][j]);                    // fma(d, isum, acc) (:2478)
                if (TYPE == FL_TYPE_Q4_1) {
                    const v4f sv = {sxv[j], sxv[j], sxv[j], sxv[j]};
                    acc[i][j] = __builtin_elementwise_fma(mwv[i], sv, acc[i][j]);          // summs += m*s (:2651)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // hipcc drains the global_load_lds of the next stage here (vmcnt(0))
    }

    // ---- store: lane holds rows m = 16*g + 4*lg + {0..3} of column n = 16*h + l15 -> one 16-byte store
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row0 = (mg0 + wm * TM + i) * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = (ng0 + wn * 4 + j) * 16 + l15;
            if (n < N && row0 < M) {
                float *p = y + (int64_t)n * ldy + row0;
                if (row0 + 3 < M) {
                    *reinterpret_cast<v4f *>(p) = acc[i][j];
                } else {
                    for (int r = 0; r < 4 && row0 + r < M; ++r) p[r] = acc[i][j][r];
                }
            }
        }
    }
}

template <int TYPE, int TM>
static hipError_t launch_gemm(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st) {
    using Cfg = GemmCfg<TYPE, TM>;
    const int MGT = W.M16 / 16, NGT = fl_roundup(N, 16) / 16;
    const int tiles = ((MGT + Cfg::MG - 1) / Cfg::MG) * ((NGT + 7) / 8);
    const size_t lds = 2 * Cfg::STAGE;
    hipLaunchKernelGGL((gemm_q4_mfma_kernel<TYPE, TM>), dim3(tiles), dim3(256), lds, st,
                       reinterpret_cast<const uint4 *>(W.qs), W.d, W.m, xq.q, xq.d, xq.s, N, W.M, MGT, NGT, W.KB, y, ldy);
    return hipGetLastError();
}

hipError_t gemm_q4_mfma(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st) {
    if ((ldy & 3) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return hipErrorInvalidValue;
    const int MGT = W.M16 / 16, NGT = fl_roundup(N, 16) / 16;
    // 128-row tiles unless that leaves CUs idle: then 64-row tiles double the workgroup count
    const bool small = ((MGT + 7) / 8) * ((NGT + 7) / 8) < 384;
    if (W.type == FL_TYPE_Q4_0)
        return small ? launch_gemm<FL_TYPE_Q4_0, 2>(W, xq, N, y, ldy, st) : launch_gemm<FL_TYPE_Q4_0, 4>(W, xq, N, y, ldy, st);
    return small ? launch_gemm<FL_TYPE_Q4_1, 2>(W, xq, N, y, ldy, st) : launch_gemm<FL_TYPE_Q4_1, 4>(W, xq, N, y, ldy, st);
}

}  // namespace fl
