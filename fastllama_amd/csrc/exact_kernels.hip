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
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <type_traits>
#include "eval_kernels.h"
#include "runtime.h"
#include "q4_device.h"
#include "gemv_prologue.h"
#include "q4_kernels.h"
#include "tp_tail.h"

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

hipError_t gemv1_q4_exact(const fl_qtensor &W, const fl_qact &xq, float *y, hipStream_t st, const float *resid);
static inline bool resid_has_ld(int) { return false; }    // N == 1: one row of y / resid, strides do not matter
hipError_t gemv_q4_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid,
                         int ldr) {
    if (N < 1 || N > 8) return hipErrorInvalidValue;
    if (N == 1 && !resid_has_ld(ldr)) {
        const hipError_t e = gemv1_q4_exact(W, xq, y, st, resid);
        if (e != hipErrorInvalidValue) return e;
        (void)hipGetLastError();
    }
    return W.type == FL_TYPE_Q4_0 ? launch_gemv_exact<FL_TYPE_Q4_0>(W, xq, N, y, ldy, st, resid, ldr)
                                  : launch_gemv_exact<FL_TYPE_Q4_1>(W, xq, N, y, ldy, st, resid, ldr);
}

// ------------------------------------------------------------------------------------------------
// N = 1, the decode kernel.  A matrix of 4096 rows has only 256 row groups, so one wave per group (above) would have to pull
// 40-108 KB through its own registers, round trip after round trip.  Here ONE workgroup per CU streams a sequence of row groups
// (group = blockIdx.x, + gridDim.x, ...) and splits the work by what depends on the running sums and what does not:
//   * NWG PRODUCER waves stream the blocks (a chunk of KC = 8 * NWG blocks at a time, D chunks in flight -- across the
//     boundaries between row groups, so the stream never drains) and do everything order-independent -- unpack, v_dot4,
//     d_w * d_x -- leaving per block and row the 8 integer lane sums and the f32 scale in a double-buffered LDS chunk;
//   * 2 CHAIN waves (lane = row x lane-sum j, one accumulator each) run  acc_j = fma(dd_b, float(q_bj), acc_j)  block after
//     block over the chunk the producers finished last, while those fill the next one: the fma chain is hidden under the weight
//     stream; at the end of a row group they add the 8 lane sums in the reference's order and store.
// Q4_0 bookkeeping: the unpacked nibbles are 16 * (nib - 8) and the stored scale is d / 16 (q4_layout.h); the producers store
// 16 isum and rn((d/16) d_x) = rn(d d_x) / 16: fma(dd/16, 16 q, a) and the reference's fma(dd, q, a) round the same real number.
// The prologue (rms_norm / silu*mul / plain Q8_0 of the f32 activation, built in LDS by all waves, gemv_prologue.h) runs once
// per workgroup.  PAIR (woven w1|w3 matrix): groups 2u, 2u+1 are the w1 / w3 rows of the same 16 features; the chain lane that
// finishes row r of both stores silu(w1 x) * (w3 x) -- ggml_silu + ggml_mul of lib/llama.cpp:428-431.
// ------------------------------------------------------------------------------------------------
template <int TYPE, int NWG, int PRO, int PAIR>
__global__ __launch_bounds__(64 * (NWG + 2)) void gemv1_q4_exact_kernel(
    int M, int units, int KB, int woven,                      // (the leading 12-14 dwords are preloaded into SGPRs: build.sh)
    const uint32_t *__restrict__ qs, const float *__restrict__ dW, const float *__restrict__ xf, const void *__restrict__ aux,
    const float *__restrict__ mW, const int8_t *__restrict__ xq, const float *__restrict__ xd, const float *__restrict__ xs,
    float *__restrict__ y, const float *__restrict__ resid, float *__restrict__ ynorm, const uint16_t *__restrict__ aux2,
    const TpTail *__restrict__ tt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    constexpr int G2 = PAIR ? 2 : 1, NT = 64 * (NWG + 2);
    constexpr int BPW = 8, KC = BPW * NWG, D = 6;                   // blocks per producer wave and chunk; chunk; chunks in flight
    constexpr int CHUNK_BYTES = KC * (512 + 64 + (TYPE == FL_TYPE_Q4_1 ? 64 + 4 : 0));
    const int lane = threadIdx.x & 63, wg = threadIdx.x >> 6;
    const bool producer = wg < NWG;
    const int nchunks = (KB + KC - 1) / KC;
    const int my_units = blockIdx.x < units ? (units - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int T = my_units * G2 * nchunks;                          // chunks this workgroup streams
    auto group_of = [&](int gidx) { return (blockIdx.x + (gidx / G2) * (int)gridDim.x) * G2 + gidx % G2; };

    // LDS: [Q8_0 activation (PRO)] [chunk buffers 0, 1]
    int8_t *lq = reinterpret_cast<int8_t *>(gsm);                   // [KB][32]
    float *ld_ = reinterpret_cast<float *>(gsm + (size_t)KB * 32);  // [KB] d
    float *ls_ = ld_ + KB;                                          // [KB] s
    unsigned char *cbase = gsm + (size_t)KB * 40;
    auto P32 = [&](int buf) { return reinterpret_cast<int *>(cbase + (size_t)buf * CHUNK_BYTES); };                // [KC][16][8]
    auto DDp = [&](int buf) { return reinterpret_cast<float *>(cbase + (size_t)buf * CHUNK_BYTES + KC * 512); };   // [KC][16]
    auto MWp = [&](int buf) { return DDp(buf) + KC * 16; };                                                          // [KC][16]
    auto SXp = [&](int buf) { return DDp(buf) + KC * 32; };                                                          // [KC]
    __shared__ double sh[4];

    GP_DECL(PRO);
    GemvPrologue<PRO, NT>::issue(pv, pw, psl, psb, xf, aux, KB, woven);

    // producer state.  A chunk gives a producer wave 8 blocks x 16 rows = 128 (block, row) items, two per lane: item i of lane L
    // is flat = 64 i + L -> block u = flat / 16, row = flat % 16 -- its 16 nibble bytes are one uint4 and a wave-load is one
    // contiguous KiB.  The scales of the 8 blocks (128 floats, contiguous) are taken two per lane, independently of the items.
    uint4 w[D][2];
    float2 dw[D], mw[D];
    auto load_chunk = [&](int t, int slot) {                       // t >= T: a cache-hot dummy (the tensor's first blocks), never used
        const bool live = t < T;
        const int gidx = live ? t / nchunks : 0, c = live ? t - gidx * nchunks : 0;
        const int64_t gb0 = live ? (int64_t)group_of(gidx) * KB + c * KC + wg * BPW : 0;   // first block of this wave's share
        const int64_t gbl = live ? (int64_t)group_of(gidx) * KB + KB - 1 : KB - 1;         // last valid block of the row group
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int flat = i * 64 + lane;
            const int64_t gb = min(gb0 + (flat >> 4), gbl);
            typedef unsigned int nt_v4u __attribute__((ext_vector_type(4)));     // (nontemporal: every weight byte is read once, by one CU)
            const nt_v4u tq = __builtin_nontemporal_load(reinterpret_cast<const nt_v4u *>(qs) + gb * 16 + (flat & 15));
            w[slot][i] = make_uint4(tq.x, tq.y, tq.z, tq.w);
        }
        const int64_t gs = min(gb0 + (lane >> 3), gbl);
        typedef float nt_v2f __attribute__((ext_vector_type(2)));
        const nt_v2f td = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f *>(dW + gs * 16 + 2 * (lane & 7)));
        dw[slot] = make_float2(td.x, td.y);
        if (TYPE == FL_TYPE_Q4_1) {
            const nt_v2f tm = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f *>(mW + gs * 16 + 2 * (lane & 7)));
            mw[slot] = make_float2(tm.x, tm.y);
        }
    };
    if (T == 0) return;                                             // (more workgroups than row groups: never launched that way)
    // Every load of the stream is issued unconditionally -- past the end of the stream and in the chain waves it reads a
    // cache-hot dummy: the compiler can then COUNT the loads in flight and wait for exactly the chunk (or the prologue
    // operands, which were requested before) it needs, s_waitcnt vmcnt(n); under conditions it waits for everything.
#pragma unroll
    for (int dd = 0; dd < D; ++dd) load_chunk(producer ? dd : T, dd);
    if constexpr (PRO != 0) {
        GemvPrologue<PRO, NT>::finish(pv, pw, psl, psb, xf, aux, KB, woven, lq, ld_, ls_, sh, ynorm, blockIdx.x == 0);
    } else {                                          // the activation is already Q8_0 (QA1 in HBM): copy it next to the chunks
        for (int i = threadIdx.x; i < KB * 2; i += NT) reinterpret_cast<uint4 *>(lq)[i] = reinterpret_cast<const uint4 *>(xq)[i];
        for (int i = threadIdx.x; i < KB; i += NT) {
            ld_[i] = xd[i];
            ls_[i] = TYPE == FL_TYPE_Q4_1 ? xs[i] : 0.f;
        }
        __syncthreads();
    }
    // the activation's k-groups, once per workgroup, in the byte order the lane sums need: (lo, hi) = elements (0,2,4,6 | 1,3,5,7)
    // -> (elements 0..3 | elements 4..7), i.e. what the producers would otherwise v_perm again for every row of every block
    for (int i = threadIdx.x; i < KB * 4; i += NT) {
        uint2 *px2 = reinterpret_cast<uint2 *>(lq) + i;
        const uint2 lh = *px2;
        *px2 = make_uint2(perm_a(lh.y, lh.x), perm_b(lh.y, lh.x));
    }
    __syncthreads();

    // The two roles run separate loops (their register sets must not be live together); every wave of the workgroup passes
    // the same number of barriers -- one per chunk -- and the branch is wave-uniform.
    if (producer) {
#pragma unroll 1
        for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const int t = t0 + dd;
                if (t >= T) break;
                const int c = t % nchunks, buf = t & 1;
                int *pq = P32(buf);
                float *pd = DDp(buf), *pm = MWp(buf), *px = SXp(buf);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int flat = i * 64 + lane, u = flat >> 4, row = flat & 15;
                    const int bl = wg * BPW + u, b = c * KC + bl;           // block inside the chunk / of the row
                    const int bc = min(b, KB - 1);
                    // dword position p of the row holds k-group p ^ 2 sw (sw = rows 8..15, constant per lane): the two 16-byte halves
                    // of the activation block are read swapped and the two halves of the result stored swapped -- no selects
                    const int sw = (row >> 3) & 1;
                    const uint4 x0 = *reinterpret_cast<const uint4 *>(lq + bc * 32 + sw * 16), x1 = *reinterpret_cast<const uint4 *>(lq + bc * 32 + (sw ^ 1) * 16);
                    const uint32_t xw[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};     // position p: (elements 0..3, 4..7) = xw[2p], xw[2p+1]
                    const uint32_t wv[4] = {w[dd][i].x, w[dd][i].y, w[dd][i].z, w[dd][i].w};
                    int out[8];                                             // the lane sums j = 2 (p ^ 2 sw) + {0, 1} at out[2p], out[2p+1]
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4) {
                        uint32_t wa, wb;
                        unpack_lanes<TYPE>(wv[p4], wa, wb);
                        out[2 * p4] = __builtin_amdgcn_sdot4((int)wa, (int)xw[2 * p4], 0, false);
                        out[2 * p4 + 1] = __builtin_amdgcn_sdot4((int)wb, (int)xw[2 * p4 + 1], 0, false);
                    }
                    if (b < KB) {
                        uint4 *dst = reinterpret_cast<uint4 *>(pq + (bl * 16 + row) * 8);
                        dst[sw] = make_uint4(out[0], out[1], out[2], out[3]);
                        dst[sw ^ 1] = make_uint4(out[4], out[5], out[6], out[7]);
                    }
                }
                {   // scales of the wave's 8 blocks: lane L holds rows 2 (L & 7), +1 of block L >> 3
                    const int bl = wg * BPW + (lane >> 3), b = c * KC + bl, bc = min(b, KB - 1);
                    const float dx = ld_[bc];
                    const float2 ddv = make_float2(__fmul_rn(dw[dd].x, dx), __fmul_rn(dw[dd].y, dx));
                    if (b < KB) {
                        *reinterpret_cast<float2 *>(pd + bl * 16 + 2 * (lane & 7)) = ddv;
                        if (TYPE == FL_TYPE_Q4_1) {
                            *reinterpret_cast<float2 *>(pm + bl * 16 + 2 * (lane & 7)) = mw[dd];
                            if ((lane & 7) == 0) px[bl] = ls_[bc];
                        }
                    }
                }
                load_chunk(t + D, dd);
                __syncthreads();            // chunk t is complete in LDS; the chain waves are done with chunk t - 1
            }
        }
    } else {
    // chain waves: lane = 8 * (row & 7) + j, chain wave cw holds rows 8 cw .. 8 cw + 7
    const int cw = wg - NWG, crow = cw * 8 + (lane >> 3), cj = lane & 7;
    float acc = 0.f, summs = 0.f, y1 = 0.f;
    // (with a tail: written through to memory, and to every peer's region -- tp_put, tp_tail.h)
    auto put_y = [&](float *p, float v) {
        if (tt) tp_put(tt, p, v);
        else *p = v;
    };
    // (all LDS reads of a batch of CB blocks are issued before the first fma: one round trip per batch instead of one per block)
    constexpr int CB = TYPE == FL_TYPE_Q4_1 ? 16 : 32;
    auto chain_chunk = [&](int t) {
        const int gidx = t / nchunks, c = t - gidx * nchunks;
        const int buf = t & 1, nb = min(KC, KB - c * KC);
        const int *pq = P32(buf) + crow * 8 + cj;
        const float *pd = DDp(buf) + crow, *pm = MWp(buf) + crow, *px = SXp(buf);
#pragma unroll
        for (int b0 = 0; b0 < KC; b0 += CB) {
            if (b0 >= nb) break;
            float dv[CB], mv[TYPE == FL_TYPE_Q4_1 ? CB : 1], sv[TYPE == FL_TYPE_Q4_1 ? CB : 1];
            int qv[CB];
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                dv[b] = pd[(b0 + b) * 16];
                qv[b] = pq[(b0 + b) * 128];
                if (TYPE == FL_TYPE_Q4_1) {
                    mv[b] = pm[(b0 + b) * 16];
                    sv[b] = px[b0 + b];
                }
            }
            if (b0 + CB <= nb) {
#pragma unroll
                for (int b = 0; b < CB; ++b) {
                    acc = __fmaf_rn(dv[b], (float)qv[b], acc);
                    if (TYPE == FL_TYPE_Q4_1) summs = __fmaf_rn(mv[b], sv[b], summs);
                }
            } else {
#pragma unroll
                for (int b = 0; b < CB; ++b) {
                    if (b0 + b < nb) {
                        acc = __fmaf_rn(dv[b], (float)qv[b], acc);
                        if (TYPE == FL_TYPE_Q4_1) summs = __fmaf_rn(mv[b], sv[b], summs);
                    }
                }
            }
        }
        if (c != nchunks - 1) return;
        // the row group is complete: ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) over the 8 lanes of a row (all end with the same bits)
        float v = acc;
        v = __fadd_rn(v, __shfl_xor(v, 4));
        v = __fadd_rn(v, __shfl_xor(v, 2));
        v = __fadd_rn(v, __shfl_xor(v, 1));
        if (TYPE == FL_TYPE_Q4_1) v = __fadd_rn(v, summs);
        acc = 0.f;
        summs = 0.f;
        const int grp = group_of(gidx), row = grp * 16 + crow;
        if constexpr (PAIR) {
            if ((gidx & 1) == 0) {
                y1 = v;                                                    // w1 . x of this feature; its w3 row comes next
            } else if (cj == 0 && row < M) {
                const uint16_t hx = __half_as_ushort(__float2half_rn(y1));                // GGML_FP32_TO_FP16
                const float sl = __half2float(__ushort_as_half(aux2[hx]));               // table_silu_f16
                put_y(y + (grp >> 1) * 16 + crow, __fmul_rn(sl, v));                      // ggml_mul(silu, tmp)
            }
        } else if (cj == 0 && row < M) {
            if (resid) v = __fadd_rn(v, resid[row]);
            put_y(y + row, v);
        }
    };
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (t > 0) chain_chunk(t - 1);
        __syncthreads();
    }
    if (T > 0) chain_chunk(T - 1);
    }
    if (tt) tp_tail<false, false, true>(tt);      // tensor parallel: this launch's rows -> every peer (tp_tail.h); all waves of every workgroup arrive here
}

// false: the activation does not fit LDS next to the chunk buffers (K > ~100 000) -> the caller takes the per-op sequence
template <int TYPE, int PRO, int PAIR>
static bool launch_gemv1_exact(const fl_qtensor &W, const fl_qact *xq, float *y, hipStream_t st, const float *resid, const float *xf,
                               const void *aux, float *ynorm, int woven, const uint16_t *aux2) {
    constexpr int NWG = 8, G2 = PAIR ? 2 : 1, KC = 8 * NWG;
    const int KB = W.KB, units = W.M16 / 16 / G2;
    const size_t lds = (size_t)KB * 40 + (size_t)2 * KC * (512 + 64 + (TYPE == FL_TYPE_Q4_1 ? 64 + 4 : 0));
    if (lds > 150 * 1024) return false;
    // dynamic LDS beyond 64 KB must be asked for once per instantiation AND device (several GPUs in one process)
    static bool attr_set[64] = {false};
    static int n_cu_dev[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemv1_q4_exact_kernel<TYPE, NWG, PRO, PAIR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return false;
        attr_set[dev] = true;
    }
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t pr;
        n_cu_dev[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    const int n_cu = n_cu_dev[dev];
    const int grid = units < n_cu ? units : n_cu;                   // one resident workgroup per CU, each streaming its share of the rows
    hipLaunchKernelGGL((gemv1_q4_exact_kernel<TYPE, NWG, PRO, PAIR>), dim3(grid), dim3(64 * (NWG + 2)), lds, st, W.M, units, KB, woven,
                       W.qs, W.d, xf, aux, W.m, xq ? xq->q : nullptr, xq ? xq->d : nullptr, xq ? xq->s : nullptr, y, resid, ynorm, aux2,
                       tp_take_tail());
    return true;
}

#define FL_TYPED(CALL0, CALL1) (W.type == FL_TYPE_Q4_0 ? (CALL0) : (CALL1))
// the exact forms of gemv_q4 (N = 1) / gemv_q4_norm / gemv_q4_silu / gemv_q4_norm_silu / gemv_q4_quant (q4_kernels.h);
// hipErrorInvalidValue: shape outside this kernel's reach -> the caller takes the per-op sequence
// (each entry point tries the round-4 lane-local-chain kernel first: it needs the tensor's QWD copy, gemv1_q4_exact_llc.hip)
hipError_t gemv1_q4_exact(const fl_qtensor &W, const fl_qact &xq, float *y, hipStream_t st, const float *resid) {
    if ((gemv1_stream(W, xq, y, st, resid) || gemv1_llc(W, xq, y, st, resid))) return hipGetLastError();
    const bool ok = FL_TYPED((launch_gemv1_exact<FL_TYPE_Q4_0, 0, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr, 0, nullptr)),
                             (launch_gemv1_exact<FL_TYPE_Q4_1, 0, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr, 0, nullptr)));
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}
hipError_t gemv_q4_norm_exact(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st) {
    if (W.K % 32 != 0 || W.K > 8192) return hipErrorInvalidValue;
    if ((gemv1_stream_norm(W, x, norm_w, ynorm, y, st) || gemv1_llc_norm(W, x, norm_w, ynorm, y, st))) return hipGetLastError();
    const bool ok = FL_TYPED((launch_gemv1_exact<FL_TYPE_Q4_0, 1, 0>(W, nullptr, y, st, nullptr, x, norm_w, ynorm, 0, nullptr)),
                             (launch_gemv1_exact<FL_TYPE_Q4_1, 1, 0>(W, nullptr, y, st, nullptr, x, norm_w, ynorm, 0, nullptr)));
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}
hipError_t gemv_q4_silu_exact(const fl_qtensor &W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid,
                              hipStream_t st, bool woven) {
    if (W.K % 32 != 0) return hipErrorInvalidValue;
    if (gemv1_llc_silu(W, h13, silu_tab, y, resid, st, woven)) return hipGetLastError();
    const bool ok = FL_TYPED((launch_gemv1_exact<FL_TYPE_Q4_0, 2, 0>(W, nullptr, y, st, resid, h13, silu_tab, nullptr, woven ? 1 : 0, nullptr)),
                             (launch_gemv1_exact<FL_TYPE_Q4_1, 2, 0>(W, nullptr, y, st, resid, h13, silu_tab, nullptr, woven ? 1 : 0, nullptr)));
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}
hipError_t gemv_q4_norm_silu_exact(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act,
                                   hipStream_t st, float *pair_ws, int form) {
    if (W.K % 32 != 0 || W.K > 8192 || W.M % 32 != 0) return hipErrorInvalidValue;
    if (((form == 0 && gemv1_stream_norm_silu(W, x, norm_w, silu_tab, act, st)) || gemv1_llc_norm_silu(W, x, norm_w, silu_tab, act, st, pair_ws, form)))
        return hipGetLastError();
    const bool ok = FL_TYPED((launch_gemv1_exact<FL_TYPE_Q4_0, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab)),
                             (launch_gemv1_exact<FL_TYPE_Q4_1, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab)));
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}
hipError_t gemv_q4_norm_silu_q8_exact(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, const fl_qact &out,
                                      hipStream_t st) {
    if (W.K % 32 != 0 || W.K > 8192 || W.M % 64 != 0) return hipErrorInvalidValue;
    return gemv1_stream_norm_silu_q8(W, x, norm_w, silu_tab, out, st) ? hipGetLastError() : hipErrorInvalidValue;
}
hipError_t gemv_q4_quant_exact(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st) {
    if (W.K % 32 != 0) return hipErrorInvalidValue;
    if ((gemv1_stream_quant(W, x, y, resid, st) || gemv1_llc_quant(W, x, y, resid, st))) return hipGetLastError();
    const bool ok = FL_TYPED((launch_gemv1_exact<FL_TYPE_Q4_0, 3, 0>(W, nullptr, y, st, resid, x, nullptr, nullptr, 0, nullptr)),
                             (launch_gemv1_exact<FL_TYPE_Q4_1, 3, 0>(W, nullptr, y, st, resid, x, nullptr, nullptr, 0, nullptr)));
    return ok ? hipGetLastError() : hipErrorInvalidValue;
}
#undef FL_TYPED

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

// the VALU form (v_dot4 + v_cvt + v_fma): the on-device cross-check of gemm_q4_exact_mfma.hip, fl_debug_mul_mat_q which = 4
hipError_t gemm_q4_exact_valu(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid,
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
            for (int l = 0; l < 8; ++l) {
                const float pr = x[i + l] * y[i + l];     // (plain operators: this file is compiled with fp contract(off))
                s = s + pr;
            }
        if (n - i >= 4) {
            for (int l = 0; l < 4; ++l) {
                const float pr = x[i + l] * y[i + l];
                s = s + pr;
            }
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

namespace fl {
hipError_t gemm_q4_exact(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid,
                         int ldr) {
    return gemm_q4_exact_mfma(W, xq, N, y, ldy, st, resid, ldr);
}
}  // namespace fl

namespace fl {
// ------------------------------------------------------------------------------------------------
// Prefill attention in exact mode: the two f32 matmuls (K.Q and V.P, lib/llama.cpp:364,389) on the f32-input MFMA.
// v_mfma_f32_32x32x2_f32 is bitwise a chain of two fmaf per output, k = 0 then k = 1, on top of C (MI355X_MICROARCH.md) -- and
// ggml_vec_dot_f32's 32 partial sums are exactly such chains: sum[jj][l] takes the elements 32 st + 8 jj + l, st = 0, 1, 2, ...
// in order.  So for one (jj, l) a 32x32 tile of partial sums is a chain of MFMAs whose k = 0 / 1 operands (lanes 0-31 / 32-63)
// are the elements of two consecutive 32-element steps; the fixed reduction tree over the 32 partial sums is 31 vector adds
// on result tiles.  Same arithmetic, bit for bit, as dot_f32_abt_exact_kernel (one half-wave per dot), at the MFMA's f32 rate.
//   scores  grid (query blocks of 32, heads): Q tile and the waves' K tiles (32 keys each, dealt round-robin, causal tiles
//           only) in LDS; per tile 32 x 2 MFMAs + the tree -> scale -> att[head][q][key]
//   pv      grid (query blocks of 32, heads): probability rows and V rows pass through LDS in 512-key pieces, feature block by
//           feature block; wave w owns the partial sums l = w and w + 4 -- 8 chains that run over ALL 32-key steps, i.e. 8
//           accumulator tiles (128 VGPRs) carried across the pieces; the waves' results meet in LDS for the last two tree
//           levels; leftover keys (P % 32) in order.  Any context length (round 3; contexts beyond 512 keys used to fall back
//           to dot_f32_abt_exact, one half-wave per dot).
// ------------------------------------------------------------------------------------------------
constexpr int XA_KT = 512;      // keys the pv kernel holds in LDS (a context of one piece)
constexpr int XA_LD = XA_KT + 1;
constexpr int XD_KT = 256;      // ... per piece and buffer behind a deeper context: two buffers, one under the MFMAs while the other is filled
constexpr int XD_LD = XD_KT + 1;

// lane (i, h) of an MFMA supplies, for the 32-element step st = 2 m + h, element 8 jj + l of row i.  The key row of a tile is kept in
// registers (the 32 elements of each of the lane's MS = ceil(NST / 2) steps, zeros for a step past the row; straight from HBM/L2
// as 128-byte pieces), the 32 query rows of the workgroup in LDS.
// SM (round 5): soft_max of the workgroup's 32 score rows in the same launch -- the workgroup computes every key tile of its query block, so
// the complete rows can wait in LDS ([32][PLD] floats, contexts of up to 1024 keys) instead of making the trip through HBM to a separate
// kernel and back: ggml_soft_max's arithmetic as softmax_rows_reg_kernel (eval_kernels.hip) has it -- row max, fp16 exp table, f64 sum (exact in
// any order), rn(t * rn_f32(1 / sum)), zeros past each row's last visible key.
template <int NST, bool SM>
__global__ __launch_bounds__(512) void attn_scores_exact_kernel(const float *__restrict__ qkv, int ldq, int N, int n_past,
                                                                const float *__restrict__ kc, int ldk, float scale,
                                                                float *__restrict__ att, int ld_att, int64_t head_stride,
                                                                const uint16_t *__restrict__ exp_tab, int PLD, int compact) {
    extern __shared__ float sm_rows[];                        // SM: [32][PLD] scaled scores
    constexpr int D = 32 * NST, MS = (NST + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // heaviest query blocks first (block i sees (i + 1) 32-key steps): with one or two workgroups resident per CU the launch runs in
    // rounds, and the CU that finishes a light block of the first round takes the heaviest one left -- every CU ends up with about
    // the same number of key steps instead of two heavy blocks on some and two light ones on others (99 -> 80 us, 35 -> 24 us)
    const int qb = (int)gridDim.y - 1 - (int)blockIdx.y, hd = blockIdx.x, q0 = qb * 32;
    const int i = lane & 31, h = lane >> 5;
    auto load_row = [&](const float *row, float (&dst)[MS][32]) {
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            const int st = 2 * m + h;
            const bool ok = st < NST;
            const float4 *p4 = reinterpret_cast<const float4 *>(row + 32 * (ok ? st : 0));
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 v = p4[q4];
                dst[m][4 * q4 + 0] = ok ? v.x : 0.f;
                dst[m][4 * q4 + 1] = ok ? v.y : 0.f;
                dst[m][4 * q4 + 2] = ok ? v.z : 0.f;
                dst[m][4 * q4 + 3] = ok ? v.w : 0.f;
            }
        }
    };
    // the query rows sit in LDS (one read per MFMA, requested ahead by the compiler); the key rows in registers
    __shared__ float Qs[32][D + 1];
    for (int idx = threadIdx.x; idx < 32 * (D / 4); idx += 512) {
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        const float4 v = *reinterpret_cast<const float4 *>(qkv + (int64_t)min(q0 + r, N - 1) * ldq + hd * D + c4 * 4);   // rows past N: the last one, never stored
        Qs[r][c4 * 4 + 0] = v.x; Qs[r][c4 * 4 + 1] = v.y; Qs[r][c4 * 4 + 2] = v.z; Qs[r][c4 * 4 + 3] = v.w;
    }
    __syncthreads();
    float kv[MS][32];
    const int last_key = n_past + min(q0 + 31, N - 1);       // keys this query block can see: [0, last_key]
    const int ntiles = last_key / 32 + 1;
    // (the NEXT tile's rows in a second register set, requested before this tile's MFMAs: 64 more registers than two waves per SIMD have -- 90 spilled)
    for (int kt = wave; kt < ntiles; kt += 8) {
        const int k0 = kt * 32;
        load_row(kc + (int64_t)min(k0 + i, last_key) * ldk + hd * D, kv);
        // t_l = v_l + v_{l+4} (lo128 + hi128), (t0 + t1) and (t2 + t3) (hadd), their sum (hadd); v_l = (s0l + s1l) + (s2l + s3l)
        v16f total, u, t;
        v16f z = {};                                         // the chains' zero start -- and the token that orders the groups (below)
#pragma unroll
        for (int lo = 0; lo < 8; ++lo) {                     // l in the order 0, 4, 1, 5, 2, 6, 3, 7
            const int l = (lo >> 1) + 4 * (lo & 1);
            v16f p01, vs;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                v16f c = z;
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    const float qa = 2 * m + h < NST ? Qs[i][32 * min(2 * m + h, NST - 1) + 8 * jj + l] : 0.f;
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, kv[m][8 * jj + l], c, 0, 0, 0);
                }
                if (jj == 0) p01 = c;
                // (whole-tile adds: two floats per v_pk_add_f32 -- the same IEEE add, half the instructions; nothing here that could contract to an fma)
                else if (jj == 1) p01 = p01 + c;             // s0 + s1
                else if (jj == 2) vs = c;
                else vs = p01 + (vs + c);                    // (s0 + s1) + (s2 + s3)
                if (NST > 2 && jj < 3) asm volatile("" : "+v"(z), "+v"(p01));   // (one chain at a time: 128 operand registers leave room for few tiles)
            }
            if ((lo & 1) == 0) t = vs;                       // v_l, waits for v_{l+4}
            else {
                t = t + vs;                                  // t_l
                if (lo == 1 || lo == 5) u = t;               // t0 / t2
                else {
                    u = u + t;                               // t0 + t1 / t2 + t3
                    if (lo == 3) total = u;
                    else total = total + u;
                }
            }
            // every MFMA of the next group starts from z, which passes through this statement together with the group's result:
            // left alone the compiler issues all 32 x MS (pure) MFMAs first and keeps their result tiles alive (1.6 KB of scratch)
            asm volatile("" : "+v"(z), "+v"(t));
        }
        // C layout: col = lane & 31 (key), row = (e & 3) + 8 (e >> 2) + 4 h (query)
        float *out = att + hd * head_stride;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ql = (e & 3) + 8 * (e >> 2) + 4 * h, q = q0 + ql, key = k0 + i;
            if (q < N && key <= n_past + q) {
                if (SM) sm_rows[ql * PLD + key] = __fmul_rn(total[e], scale);
                else out[(int64_t)q * ld_att + key] = __fmul_rn(total[e], scale);
            }
        }
    }
    if constexpr (SM) {
        __syncthreads();
        const int P = n_past + N;
        constexpr int IT = 16;                                // 64 x 16 = 1024 keys at most (the launcher checks)
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int ql = wave * 4 + rr, q = q0 + ql;
            if (q >= N) break;                                // (wave-uniform)
            const int L = min(P, n_past + q + 1);
            const float *srow = sm_rows + ql * PLD;
            float x[IT];
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                const int c = lane + 64 * u;
                x[u] = c < L ? srow[c] : -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < IT; ++u) mx = fmaxf(mx, x[u]);
            mx = wave_max_f32(mx);
#pragma unroll
            for (int u = 0; u < IT; ++u) {                    // (-inf - mx rounds to fp16 -inf: a valid table index, entry 0)
                const float t = __half2float(__ushort_as_half(exp_tab[__half_as_ushort(__float2half_rn(x[u] - mx))]));
                x[u] = x[u] != -INFINITY ? t : 0.f;
            }
            double sum = 0.0;
#pragma unroll
            for (int u = 0; u < IT; ++u) sum += (double)x[u];
            sum = wave_sum_f64(sum);
            const float inv = (float)(1.0 / sum);
            float *prow = att + hd * head_stride + (int64_t)q * ld_att;
            if (compact) {                                    // fp16 table values + the row's factor (softmax_rows_reg_kernel, eval_kernels.hip)
                __half *ph = reinterpret_cast<__half *>(prow);
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    const int c = lane + 64 * u;
                    if (c < P) ph[c] = __float2half_rn(c < L ? x[u] : 0.f);
                }
                if (lane == 0) prow[ld_att - 1] = inv;
            } else {
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    const int c = lane + 64 * u;
                    if (c < P) prow[c] = c < L ? __fmul_rn(x[u], inv) : 0.f;
                }
            }
        }
    }
}

#ifdef XA_TIMING   // development build only: per-workgroup clocks of the V.P launch (scripts/dev/xa_timeline.py)
__device__ long long xa_dbg[2 * 1024 * 16];      // [0]: the 100 MHz wall clock, [1]: the shader clock's counter (their ratio = the clock the launch ran at)
#define XA_STAMP(k) do { if (threadIdx.x == 0) { const int xi_ = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) % 1024 * 16 + (k); xa_dbg[xi_] = wall_clock64(); xa_dbg[1024 * 16 + xi_] = clock64(); } } while (0)
#else
#define XA_STAMP(k) do {} while (0)
#endif
// a barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store of the wave (s_waitcnt vmcnt(0)) -- here
// that is the V block requested ahead and the Q8_0 / f32 stores of the previous feature block, a round trip each
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// ONE: the whole context is one piece for every query block (n_past + N <= 512): its own kernel, because the two forms want different
// registers (a V block in flight + four chain tiles here; eight chain tiles carried across pieces there)
// HP (round 6, several pieces only): the probabilities arrive COMPACT -- fp16 table values t in the first half of each row and the row's factor inv
// in its last float (softmax_rows_reg_kernel, eval_kernels.hip) -- and p = rn(t * inv), the product soft_max itself would have stored, is formed while
// the piece goes to LDS.  Every probability row is read once per feature block: half the bytes each time (profiles/r06_attn_exact.md).
template <bool ONE, bool HP = false>
__global__ __launch_bounds__(256) void attn_pv_exact_kernel(const float *__restrict__ att, int ld_att, int64_t head_stride, int D, int N,
                                                            int n_past, const float *__restrict__ vc, int n_ctx, float *__restrict__ ao,
                                                            int ldo, int8_t *__restrict__ oq, float *__restrict__ od,
                                                            float *__restrict__ os, uint16_t *__restrict__ oh) {
    extern __shared__ __attribute__((aligned(16))) float xs_[];
    constexpr int LD = ONE ? XA_LD : XD_LD;
    float *Ps = xs_;                                         // [32 queries][LD]   probabilities, zero past each query's last key
    float *Vs = xs_ + 32 * LD;                               // [32 features][LD]     (several pieces: buffer 0; buffer 1 = the next 64 rows)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // heaviest query blocks first (block i sees (i + 1) 32-key steps): with one or two workgroups resident per CU the launch runs in
    // rounds, and the CU that finishes a light block of the first round takes the heaviest one left -- every CU ends up with about
    // the same number of key steps instead of two heavy blocks on some and two light ones on others (99 -> 80 us, 35 -> 24 us)
    const int qb = (int)gridDim.y - 1 - (int)blockIdx.y, hd = blockIdx.x, q0 = qb * 32;
    const int i = lane & 31, h = lane >> 5;
    const int P = n_past + N;                                // the dot runs over all P keys (soft_max wrote zeros past the diagonal)
    const int kend = min(P, n_past + min(q0 + 31, N - 1) + 1);     // ... but past this block's last visible key every term is +0
    const int np = P & ~31, nbody = min(np, (kend + 31) & ~31);     // whole 32-key steps that can hold a non-zero probability
    const int npc = (nbody + XD_KT - 1) / XD_KT;             // 256-key pieces of the body: the 32 chains of an output run through all of them
    const float *prow = att + hd * head_stride;
    // staging of keys [k0, k0 + kt): float4 pieces, sixteen loads in flight per thread (a one-load-at-a-time loop is a chain of L2
    // round trips); keys >= kend and rows >= rows_valid become +0
    auto stage = [&](float *dst, const float *src, int64_t row_stride, int rows_valid, int k0, int kt) {
        const int q4n = kt / 4, total4 = 32 * q4n;
        for (int base = threadIdx.x; base < total4; base += 16 * 256) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = min(base + u * 256, total4 - 1), r = idx / q4n, k4 = idx % q4n;
                v[u] = *reinterpret_cast<const float4 *>(src + (int64_t)min(r, rows_valid - 1) * row_stride + min(k0 + k4 * 4, n_ctx - 4));
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = base + u * 256;
                if (idx < total4) {
                    const int r = idx / q4n, k = (idx % q4n) * 4;
                    float *d = dst + r * LD + k;
                    const bool rv = r < rows_valid;
                    d[0] = rv && k0 + k < kend ? v[u].x : 0.f;
                    d[1] = rv && k0 + k + 1 < kend ? v[u].y : 0.f;
                    d[2] = rv && k0 + k + 2 < kend ? v[u].z : 0.f;
                    d[3] = rv && k0 + k + 3 < kend ? v[u].w : 0.f;
                }
            }
        }
    };
    // The whole context in one piece (<= 512 keys): load and store are separate steps, so that the next feature block's V rows are
    // requested BEFORE the current block's MFMAs and written to LDS after them -- their round trip used to stand between every two
    // blocks (profiles/r04_attn_exact.md).  Thread -> 16 float4: lane (rr = lane >> 3, kk = lane & 7) of wave w takes, for u = 0 .. 15,
    // row 8 (u & 3) + rr, keys 32 (4 w + (u >> 2)) + 4 kk .. + 3: a wave-load is 8 rows x one 128-byte line, and a wave-store of one
    // component lands on bank (rr + 4 kk + const) % 32 -- all 32 lanes of a half wave on different banks (keys along the lanes put a
    // whole wave on 8 banks: 4-way conflicts on 64 stores per thread and block).
    const int s_rr = lane >> 3, s_kk = lane & 7;
    // (k0: first key of the piece -- 0 when the context is one piece)
    // Buffer loads (round 5): the wave-uniform part of the address (the piece's first key) travels in the scalar offset, the lane's part is
    // one 32-bit register per load, and bytes past the tile's rows read as zero (descriptor bounds; keys past the row's end are the next
    // row's -- every such value is past the piece's columns or belongs to a step past the body, which the chains skip).  As flat loads the sixteen 64-bit addresses of a tile lived in
    // registers (or, with a second tile in flight, in scratch memory).
    auto tile_rsrc = [&](const float *base, int rows, int row_stride) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)((unsigned)rows * (unsigned)row_stride * 4u), 0x00020000);
    };
    // (part: all sixteen loads, or -- tag 0 .. 3 -- a quarter of them: a tile requested under a piece's chains goes out a quarter per 128-key trip,
    // between the MFMAs.  All at once, the 96-128 KB of a piece queue at the CU's one address unit for ~1500 cycles, and a wave issues in order:
    // it stood there before its first MFMA.)
    constexpr std::integral_constant<int, -1> whole{};
    auto wide_load = [&](float4 (&v)[16], __amdgpu_buffer_rsrc_t rs, int row_stride, int rows_valid, int k0, auto part) __attribute__((always_inline)) {
        constexpr int T = decltype(part)::value;
#pragma unroll
        for (int u = T < 0 ? 0 : 4 * T; u < (T < 0 ? 16 : 4 * T + 4); ++u) {
            const int row = 8 * (u & 3) + s_rr, col = 32 * (4 * wave + (u >> 2)) + 4 * s_kk;
            const int voff = (min(row, rows_valid - 1) * row_stride + col) * 4;
            typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
            const v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k0 * 4, 0);
            v[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        }
    };
    // (No masks, round 6: every key of a body piece is < P, where soft_max wrote +0 past a query's last visible key and the cache holds real V rows --
    // p = +0 times a finite v adds a zero to a chain that never holds -0; rows past the batch are copies of its last row and are never stored.  The
    // per-value compares and selects were a third of the 5000 cycles a piece spent between its two barriers.)
    // valid: the piece's body keys (a multiple of 32); columns [valid, its round-up to whole 128-key trips of the chains) are ZEROS, so that the chains
    // run without a compare or select of their own (a ragged last piece only: a whole piece takes the branch without the selects)
    auto wide_store = [&](float *dst, const float4 (&v)[16], int valid) {
        const int kt = (valid + 127) & ~127;
        if (kt == valid) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int row = 8 * (u & 3) + s_rr, col = 32 * (4 * wave + (u >> 2)) + 4 * s_kk;
                if (col < kt) {                              // (wave-uniform: a wave-store covers 32 keys)
                    float *d = dst + row * XA_LD + col;
                    d[0] = v[u].x;
                    d[1] = v[u].y;
                    d[2] = v[u].z;
                    d[3] = v[u].w;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int row = 8 * (u & 3) + s_rr, col = 32 * (4 * wave + (u >> 2)) + 4 * s_kk;
                if (col < kt) {
                    float *d = dst + row * XA_LD + col;
                    const bool in = col < valid;             // (the lane's four keys are inside or outside together)
                    d[0] = in ? v[u].x : 0.f;
                    d[1] = in ? v[u].y : 0.f;
                    d[2] = in ? v[u].z : 0.f;
                    d[3] = in ? v[u].w : 0.f;
                }
            }
        }
    };
    auto chunk_len = [&](int c) { return min(XA_KT, nbody - c * XA_KT); };                  // body keys of piece c
    const int rows_q = min(32, N - q0);
    // ---- several pieces (ONE = false; round 6) ----
    // Two LDS buffers of 256 keys (P tile + V tile each): the MFMA chains of piece g run from buffer g & 1 while piece g + 1 goes from registers into the
    // other buffer and piece g + 2 is requested -- both AMONG the MFMAs (sched_group_barrier), so the staging costs issue slots, not time.  (Round 5: one
    // 512-key buffer; per piece 2.6 us of stores between two barriers, then 4.1 us of chains that waited out an LDS round trip per pair of MFMAs and had
    // queued the next piece's 24 loads per thread before the first one -- 1.85 us of MFMA work; profiles/r06_attn_exact.md.)  A piece is (feature block,
    // 256 keys); the sequence runs through the feature blocks, so the first piece of the next block is under way during the last of this one.
    // thread -> f32 tile: 8 float4, lane (rr, kk) of wave w, u = 0 .. 7: row 8 (u & 3) + rr, keys 32 (2 w + (u >> 2)) + 4 kk .. + 3;
    //        -> compact P tile: 4 x 16 bytes, u = 0 .. 3: row 8 u + rr, keys 64 w + 8 kk .. + 7 (a wave-load: 8 rows x one 128-byte line either way)
    typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
    struct PieceRegs {
        v4u_ hp[HP ? 4 : 1];
        float4 fp[HP || ONE ? 1 : 8];
        float4 v[ONE ? 1 : 8];
    };
    float inv4[4];                                           // compact P: the factors of rows 8 r + rr
    if constexpr (HP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) inv4[r] = prow[(int64_t)(q0 + min(8 * r + s_rr, rows_q - 1)) * ld_att + ld_att - 1];
    }
    auto piece_len = [&](int c) { return min(XD_KT, nbody - c * XD_KT); };                  // body keys of piece c (a multiple of 32)
    // (part: the whole piece (-1) or one of its twelve wave-loads, numbered as store_piece's parts)
    auto load_piece = [&](PieceRegs &R, __amdgpu_buffer_rsrc_t rsp, __amdgpu_buffer_rsrc_t rsv, int k0, auto part) __attribute__((always_inline)) {
        constexpr int PT = decltype(part)::value;
        if constexpr (HP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (PT < 0 || PT == u)
                    R.hp[u] = __builtin_amdgcn_raw_buffer_load_b128(rsp, min(8 * u + s_rr, rows_q - 1) * ld_att * 4 + (64 * wave + 8 * s_kk) * 2, k0 * 2, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!(PT < 0 || PT == (HP ? 4 + u : u))) continue;
            const int row = 8 * (u & 3) + s_rr, col = 32 * (2 * wave + (u >> 2)) + 4 * s_kk;
            if constexpr (!HP && !ONE) {
                const v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rsp, (min(row, rows_q - 1) * ld_att + col) * 4, k0 * 4, 0);
                R.fp[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
            if constexpr (!ONE) {
                const v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rsv, (row * n_ctx + col) * 4, k0 * 4, 0);
                R.v[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
        }
    };
    // valid body keys; columns [valid, whole 128-key trips) become zeros (FULL: valid = 256, no compare at all); columns past the trips are not written
    // (pd / vd: the LANE's first element of the buffer's P / V tile -- st_base below -- as an offset into the LDS array that the compiler cannot fold
    // with the other buffers': the four tiles span 128 KB, a ds instruction's immediate reaches 64 KB, and folded into one base every address of
    // every buffer became a register of its own, computed ahead of the loop and spilled: 1.1 KB of scratch)
    // valid body keys; the columns behind them become zeros: every piece runs its two 128-key trips.  part: all of the piece (-1) or one of its
    // twelve wave-stores, in the order the loads were issued (compact P: four P then eight V; f32 P: eight P + V pairs)
    auto store_piece = [&](const PieceRegs &R, int pd_, int vd_, int valid, auto part) __attribute__((always_inline)) {
        constexpr int PT = decltype(part)::value;
        float *const pd = xs_ + pd_, *const vd = xs_ + vd_;
        if constexpr (HP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (PT < 0 || PT == u) {
                    float *d = pd + 8 * u * LD;
                    const bool in = 64 * wave + 8 * s_kk < valid;     // (the lane's eight keys together; a select, not a factor: those halves are anything)
                    const unsigned w4[4] = {in ? R.hp[u].x : 0u, in ? R.hp[u].y : 0u, in ? R.hp[u].z : 0u, in ? R.hp[u].w : 0u};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        d[j] = __fmul_rn(__half2float(__ushort_as_half((unsigned short)(w4[j >> 1] >> (16 * (j & 1))))), inv4[u]);
                }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (PT < 0 || PT == (HP ? 4 + u : u)) {
                const bool in = 32 * (2 * wave + (u >> 2)) + 4 * s_kk < valid;
                if constexpr (!HP && !ONE) {
                    float *d = pd + 8 * (u & 3) * LD + 32 * (u >> 2);
                    d[0] = in ? R.fp[u].x : 0.f; d[1] = in ? R.fp[u].y : 0.f; d[2] = in ? R.fp[u].z : 0.f; d[3] = in ? R.fp[u].w : 0.f;
                }
                if constexpr (!ONE) {
                    float *d = vd + 8 * (u & 3) * LD + 32 * (u >> 2);
                    d[0] = in ? R.v[u].x : 0.f; d[1] = in ? R.v[u].y : 0.f; d[2] = in ? R.v[u].z : 0.f; d[3] = in ? R.v[u].w : 0.f;
                }
            }
    };
    constexpr bool one_piece = ONE;                          // short contexts: P staged once, V blocks requested one ahead
    const __amdgpu_buffer_rsrc_t rsP = tile_rsrc(prow + (int64_t)q0 * ld_att, rows_q, ld_att);
    float4 vnext[16];
    XA_STAMP(0);
    if constexpr (one_piece) {
        float4 pfirst[16];
        wide_load(pfirst, rsP, ld_att, rows_q, 0, whole);
        wide_load(vnext, tile_rsrc(vc + (int64_t)(hd * D) * n_ctx, 32, n_ctx), n_ctx, 32, 0, whole);
        wide_store(Ps, pfirst, chunk_len(0));
        wide_store(Vs, vnext, chunk_len(0));
    }
    XA_STAMP(1);
    float *Ts = xs_ + (ONE ? 64 * XA_LD : 128 * XD_LD);      // the waves' t tiles meet here: [4 waves][16 e][64 lanes]
    float *Os = Ts + 4 * 64 * 16;                            // oq: the finished [32 queries][33] tile of a feature block, on its way to Q8_0
    const int nleft = P - np;                                // < 32 keys behind the body, taken in order
    const bool left_visible = nleft > 0 && kend > np;
    // several pieces: what lives across pieces AND feature blocks -- the registers of the piece under way, the first trip's operands of the next
    // piece (read right behind the barrier that completed its buffer), the 8 chain tiles of the feature block
    PieceRegs R;
    // what follows a feature block's chains: the waves' t tiles meet, the last two tree levels, the leftover keys, the stores (f32 rows or Q8_0 blocks)
    auto finish_block = [&](int d0, const v16f &tw) __attribute__((always_inline)) {
        // (the thread's indices taken afresh: derived from the kernel's own, every address below is loop-invariant, and the compiler computes all of
        // them ahead of the piece loop -- where the chains need the registers -- and spills them)
        int tid_ = threadIdx.x;
        if constexpr (!ONE) asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
        if (d0 == 0) XA_STAMP(2); else if (d0 == 32) XA_STAMP(6);
        lds_barrier();                                        // every wave is done with Ps / Vs
        if (left_visible) {                                  // the leftover keys [np, P) to columns 0.. of both tiles
            if constexpr (HP) {
                for (int idx = tid; idx < 32 * 32; idx += 256) {
                    const int r = idx >> 5, k = idx & 31, rc = min(r, rows_q - 1);
                    const float *row = prow + (int64_t)(q0 + rc) * ld_att;
                    const float t = __half2float(reinterpret_cast<const __half *>(row)[min(np + k, P - 1)]);
                    Ps[r * LD + k] = r < rows_q && np + k < kend ? __fmul_rn(t, row[ld_att - 1]) : 0.f;
                }
            } else
                stage(Ps, prow + (int64_t)q0 * ld_att, ld_att, rows_q, np, 32);
            stage(Vs, vc + (int64_t)(hd * D + d0) * n_ctx, n_ctx, 32, np, 32);
        }
        {
            float *dst = Ts + wave * 1024 + lane;            // [wave][e][lane]: a wave's store / load of one e is 64 consecutive floats
#pragma unroll                                               // ([wave][lane][e] put every lane of a load on two banks: 32-way conflicts)
            for (int e = 0; e < 16; ++e) dst[e * 64] = tw[e];
        }
        lds_barrier();   
        {
            // every wave finishes four of the sixteen query rows a lane holds (e = 4 wave .. 4 wave + 3): the four t tiles are read first
            // (one LDS round trip), then -- contexts with P % 32 != 0 only -- the leftover keys in the reference's order, then the stores.
            // (Wave 0 alone, row by row, each behind the branches of the leftover code: 2.2 us per feature block, profiles/r04_attn_exact.md.)
            const float *t0 = Ts + lane, *t1 = Ts + 1024 + lane, *t2 = Ts + 2048 + lane, *t3 = Ts + 3072 + lane;
            // C layout: col = lane & 31 = feature, row = (e & 3) + 8 (e >> 2) + 4 h = query
            const float *vr = Vs + i * LD;
            float sv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * wave + j;
                sv[j] = __fadd_rn(__fadd_rn(t0[e * 64], t1[e * 64]), __fadd_rn(t2[e * 64], t3[e * 64]));   // (t0+t1) + (t2+t3) (hadd, hadd)
            }
            if (nleft > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ql = j + 8 * wave + 4 * h;             // (e & 3) + 8 (e >> 2) + 4 h with e = 4 wave + j
                    const float *pr = Ps + ql * LD;
                    float s = sv[j];
                    // the n % 32 leftovers as the reference's build compiled them (chunks of 8, one of 4: rounded products added in
                    // order; the last n % 4: FMAs); keys past this block's last visible one carry p = +0 and change nothing
                    int k = 0;
                    for (; k + 8 <= nleft; k += 8)
                        for (int u = 0; u < 8; ++u)
                            if (np + k + u < kend) {
                                const float prd = pr[k + u] * vr[k + u];
                                s = s + prd;
                            }
                    if (nleft - k >= 4) {
                        for (int u = 0; u < 4; ++u)
                            if (np + k + u < kend) {
                                const float prd = pr[k + u] * vr[k + u];
                                s = s + prd;
                            }
                        k += 4;
                    }
                    for (; k < nleft; ++k)
                        if (np + k < kend) s = __fmaf_rn(pr[k], vr[k], s);
                    sv[j] = s;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ql = j + 8 * wave + 4 * h, q = q0 + ql;
                if (oq) Os[ql * 33 + i] = sv[j];
                else if (q < N) ao[(int64_t)q * ldo + hd * D + d0 + i] = sv[j];
            }
        }
        if (d0 == 0) XA_STAMP(3);
        if (oq) {
            // the Q8_0 operand of the wo matmul, written here (quantize_row_q8_0 of the merged [N, n_embd] rows, lib/ggml.c:8063-8075: a
            // block = 32 consecutive features of one query = one row of this tile): QA16 (+ its XH16 copy), 4 adjacent lanes per block,
            // the arithmetic of quantize_q8_kernel; queries past N are the layout's padding: zero blocks
            lds_barrier();   
            if (tid < 128) {
                const int ql = tid >> 2, part = tid & 3, q = q0 + ql, KBo = ldo >> 5, kg = ((hd * D + d0) >> 3) + part;
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = q < N ? Os[ql * 33 + part * 8 + u] : 0.f;
                if (q < ((N + 15) & ~15)) {
                    quantize_store_group(v, q, kg, KBo, 16, oq, od, os, oh);
                } else if (oh) {                                 // (columns [N16, N32) exist in the XH16 copy only)
                    unsigned char *dst = reinterpret_cast<unsigned char *>(oh) + ((((int64_t)(q >> 5) * KBo + (kg >> 2)) * 2 + (part >> 1)) * 64 + (q & 31)) * 16 + (part & 1) * 8;
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(0, 0);
                    *reinterpret_cast<uint2 *>(dst + 512) = make_uint2(0, 0);
                }
            }
        }
        if (d0 == 0) XA_STAMP(4);
    };
    // wave w: partial sums l = w and l = w + 4, all four jj: chains over ALL 32-key steps of the body, two steps per MFMA.
    // chain(acc, NL, l, valid): the four jj chains of the NL partial sums l, l + 4 over the staged piece (valid body keys + zeros to a whole trip).
    // A trip = 128 keys = 8 NL MFMAs; its 16 NL operand reads are issued one trip AHEAD, before the previous trip's MFMAs (round 6): a wave is alone
    // on its SIMD, nothing else covers the LDS round trip -- read-then-multiply per pair of MFMAs ran them at 150 cycles apiece instead of 64
    // (profiles/r06_attn_exact.md).
    // between(tag t), t = 0 .. 3: what the caller wants issued among the MFMAs of trip t (a quarter of the next tiles' loads); called for every t.
    auto chain = [&](auto &acc, auto nl_tag, int l, int valid, auto &&between) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_tag)::value;
        const int kt = (valid + 127) & ~127;              // (whole trips: the tiles hold zeros behind the body)
        auto fetch = [&](float (&a)[NL][2][4], float (&b)[NL][2][4], int cs) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int n = 0; n < NL; ++n)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int e = cs + 64 * u + 32 * h + 8 * jj + l + 4 * n;
                        a[n][u][jj] = Ps[i * XA_LD + e];
                        b[n][u][jj] = Vs[i * XA_LD + e];
                    }
        };
        auto mac = [&](const float (&a)[NL][2][4], const float (&b)[NL][2][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int n = 0; n < NL; ++n)
                        acc[n][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n][u][jj], b[n][u][jj], acc[n][jj], 0, 0, 0);
        };
        float a0[NL][2][4], b0[NL][2][4], a1[NL][2][4], b1[NL][2][4];
        if (kt > 0) fetch(a0, b0, 0);
        static_assert(XA_KT == 512, "four 128-key trips per piece");
        auto two_trips = [&](auto t0) __attribute__((always_inline)) {
            constexpr int T0 = decltype(t0)::value, cs = 128 * T0;
            const bool first = cs < kt, second = cs + 128 < kt;      // (wave-uniform)
            if (second) fetch(a1, b1, cs + 128);
            between(std::integral_constant<int, T0>{});
            if (first) mac(a0, b0);
            if (cs + 256 < kt) fetch(a0, b0, cs + 256);
            between(std::integral_constant<int, T0 + 1>{});
            if (second) mac(a1, b1);
        };
        two_trips(std::integral_constant<int, 0>{});
        two_trips(std::integral_constant<int, 2>{});
    };
    auto nothing = [](auto) {};
    constexpr std::integral_constant<int, 1> one_sum{};
    if constexpr (one_piece) {
    for (int d0 = 0; d0 < D; d0 += 32) {
        v16f tw;
        {
            // one partial sum at a time: its four chain tiles (64 registers) are folded to v_l = (s0+s1)+(s2+s3) before the other starts
            // -- with both alive (128) next to the 64 registers of the V block in flight the kernel shuffled ~600 values between the
            // two register files per feature block (profiles/r04_attn_exact.md)
            lds_barrier();                                    // P, V staged / the previous block is done with Ps, Vs and Ts
            const int kt = chunk_len(0);
            const bool more = d0 + 32 < D;
            const __amdgpu_buffer_rsrc_t rsV = tile_rsrc(vc + (int64_t)(hd * D + d0 + 32) * n_ctx, more ? 32 : 0, n_ctx);
            auto next_v = [&](auto part) __attribute__((always_inline)) { if (more) wide_load(vnext, rsV, n_ctx, 32, 0, part); };
            v16f v0;
#pragma unroll
            for (int li = 0; li < 2; ++li) {
                v16f t4[1][4] = {{{}, {}, {}, {}}};
                if (li == 0) chain(t4, one_sum, wave + 4 * li, nbody, next_v);
                else chain(t4, one_sum, wave + 4 * li, nbody, nothing);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = __fadd_rn(__fadd_rn(t4[0][0][e], t4[0][1][e]), __fadd_rn(t4[0][2][e], t4[0][3][e]));   // (s0+s1)+(s2+s3)
                    if (li == 0) v0[e] = v;
                    else tw[e] = __fadd_rn(v0[e], v);                                                            // t_w = v_w + v_{w+4} (lo128 + hi128)
                }
                if (li == 0) asm volatile("" : "+v"(v0));     // (folded before the second partial sum's MFMAs are issued)
            }
        }
        finish_block(d0, tw);
        if (one_piece && d0 + 32 < D) {
            lds_barrier();                                    // wave 0 / the quantizing threads are done with Ps, Vs and Os
            if (left_visible)                                // the single staged P piece was overwritten by the leftovers: stage it again
                stage(Ps, prow + (int64_t)q0 * ld_att, ld_att, rows_q, 0, chunk_len(0));
            wide_store(Vs, vnext, chunk_len(0));
        }
        if (d0 == 0) XA_STAMP(5);
    }
    } else {
        // the lane's first element of buffer 0's tiles, for the chains' reads (row i, key 32 h + wave) and for the stores (store_piece); buffer 1: + 64 LD
        auto own = [](int v) { asm volatile("" : "+v"(v)); return v; };
        const int rd_lane = i * LD + 32 * h + wave, st_lane = s_rr * LD + 64 * wave + (HP ? 8 : 4) * s_kk, stv_lane = 32 * LD + s_rr * LD + 64 * wave + 4 * s_kk;
        // wave w: partial sums l = w and w + 4; a trip = 128 keys = 16 MFMAs, its 32 operand reads issued one trip ahead
        auto fetchd = [&](float (&a)[2][2][4], float (&b)[2][2][4], int pb, int cs) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int e = cs + 64 * u + 8 * jj + 4 * n;
                        a[n][u][jj] = xs_[pb + e];
                        b[n][u][jj] = xs_[pb + 32 * LD + e];
                    }
        };
        auto macd = [&](v16f (&tl)[2][4], const float (&a)[2][2][4], const float (&b)[2][2][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        tl[n][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n][u][jj], b[n][u][jj], tl[n][jj], 0, 0, 0);
        };
        auto rsv = [&](int dd) { return tile_rsrc(vc + (int64_t)(hd * D + dd) * n_ctx, 32, n_ctx); };
        constexpr std::integral_constant<int, -1> whole_piece{};
        // The sequence of pieces: (feature block d0, 256 keys c), c fastest; the buffers alternate.  Under the 32 MFMAs of a piece: its second trip's operand
        // reads (trip 0), then the NEXT piece of the sequence from the registers into the other buffer (the stores among the last ten MFMAs of trip 1); the
        // registers free again, the piece after that one is requested -- it has a piece's time (~0.9 us) until its own stores.  One straight-line body:
        // with the MFMAs in several branches (ragged pieces, a last piece without a successor) the compiler gave each branch its own accumulator registers
        // and copied all 128 at every join.  So a ragged piece runs both trips (its buffer holds zeros behind the body) and the last piece of all stores zeros.
        // (A second register set -- the piece stored during piece g requested while piece g - 2 ran, two pieces' time for its loads instead of the 0.6 us
        // between a request and its first store here -- measured SLOWER, 159 against 143 us per launch: 48 more registers than the 256 the operands can
        // live in, so the sets travelled through the accumulator file, and the compiler's count of outstanding loads turned conservative across the
        // loop's two bodies: half the pieces waited for the newest request anyway.)
        auto seq_next = [&](int &c, int &d0) { if (++c == npc) { c = 0; d0 += 32; } };
        int c = 0, d0 = 0, cur = 0;                           // the piece about to run, its buffer
        int cl = 0, dl = 0;                                   // the piece requested last
        load_piece(R, rsP, rsv(0), 0, whole_piece);
        store_piece(R, own(st_lane), own(stv_lane), piece_len(0), whole_piece);
        // (a request past the end of the sequence goes out all the same, for the last block's rows: one straight-line body)
        seq_next(cl, dl);
        load_piece(R, rsP, rsv(min(dl, D - 32)), cl * XD_KT, whole_piece);
        lds_barrier();
        v16f tl[2][4] = {{{}, {}, {}, {}}, {{}, {}, {}, {}}};
        float a0[2][2][4], b0[2][2][4];
        fetchd(a0, b0, own(rd_lane), 0);
        XA_STAMP(1);
        auto piece = [&](PieceRegs &RS) __attribute__((always_inline)) {      // RS: holds the next piece; takes the one after it
            int c1 = c, d1 = d0;
            seq_next(c1, d1);
            const bool has1 = d1 < D;
            const int rd = own(rd_lane + cur * 64 * LD), rdn = own(rd_lane + (cur ^ 1) * 64 * LD);
            const int stp = own(st_lane + (cur ^ 1) * 64 * LD), stv = own(stv_lane + (cur ^ 1) * 64 * LD);
            float a1[2][2][4], b1[2][2][4];
            fetchd(a1, b1, rd, 128);
            macd(tl, a0, b0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // two LDS reads
            }
            __builtin_amdgcn_sched_barrier(0);                // (the stores stay behind trip 0)
            // trip 1: from its fifth MFMA on, one of the next piece's twelve wave-stores in front of each (the order pinned: left alone, the scheduler
            // puts every store -- and its wait for the load -- in front of the first MFMA)
            // ... and right behind each store, the same part of the piece after that one is requested into the registers just freed: every load has
            // one whole piece's time (requested all together behind the last store, the first had 0.35 us less)
            const int v1 = has1 ? piece_len(c1) : 0;
            seq_next(cl, dl);
            const __amdgpu_buffer_rsrc_t rsvn = rsv(min(dl, D - 32));
            auto trip1 = [&](auto... K) __attribute__((always_inline)) {
                auto one = [&](auto k) __attribute__((always_inline)) {
                    constexpr int KK = decltype(k)::value, u = KK >> 3, jj = (KK >> 1) & 3, n = KK & 1;
                    if constexpr (KK >= 4) {
                        store_piece(RS, stp, stv, v1, std::integral_constant<int, KK - 4>{});
                        load_piece(RS, rsP, rsvn, cl * XD_KT, std::integral_constant<int, KK - 4>{});
                    }
                    tl[n][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[n][u][jj], b1[n][u][jj], tl[n][jj], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                (one(K), ...);
            };
            trip1(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
                  std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{},
                  std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{},
                  std::integral_constant<int, 12>{}, std::integral_constant<int, 13>{}, std::integral_constant<int, 14>{}, std::integral_constant<int, 15>{});
            lds_barrier();                                    // buffer cur is free, buffer cur ^ 1 complete
            fetchd(a0, b0, rdn, 0);
            if (c1 == 0) {                                    // that was the feature block's last piece
                Ps = xs_ + cur * 64 * LD;                     // (the leftovers go through the buffer just left; the other one holds the next block's first piece)
                Vs = Ps + 32 * LD;
                v16f tw;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v0 = __fadd_rn(__fadd_rn(tl[0][0][e], tl[0][1][e]), __fadd_rn(tl[0][2][e], tl[0][3][e]));   // (s0+s1)+(s2+s3)
                    const float v1_ = __fadd_rn(__fadd_rn(tl[1][0][e], tl[1][1][e]), __fadd_rn(tl[1][2][e], tl[1][3][e]));
                    tw[e] = __fadd_rn(v0, v1_);                                                                             // t_w = v_w + v_{w+4} (lo128 + hi128)
                }
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) tl[n][jj] = v16f{};
                finish_block(d0, tw);
                if (has1) lds_barrier();                      // (the next piece's stores go into the buffer the leftovers were read from)
            }
            c = c1;
            d0 = d1;
            cur ^= 1;
        };
        while (d0 < D) piece(R);
    }
    XA_STAMP(7);
}
#ifdef XA_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_xa_timing(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xa_dbg), sizeof(long long) * 1024 * 16); }
extern "C" __attribute__((visibility("default"))) int fl_debug_xa_cycles(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xa_dbg), sizeof(long long) * 1024 * 16, sizeof(long long) * 1024 * 16); }
#endif

// ------------------------------------------------------------------------------------------------
// V.P behind a deep context with EIGHT waves (round 6, second form): wave w owns partial sum l = w (its four jj chains: 64 accumulator registers), two
// waves share a SIMD -- so one wave's stores of the next piece (LDS-write bound: 64 KB per 256-key piece, ~1100 cycles that a wave alone on its SIMD
// could not put under its own MFMAs, profiles/r06_attn_exact.md) run beside the other wave's MFMAs.  The sequence of pieces, the two 256-key buffers,
// the compact probabilities, the zero-filled ragged pieces and the leftover keys are attn_pv_exact_kernel<false, .>'s; the tree behind the chains is
// the same one, the eight v_l tiles meeting in LDS in two steps (t_w = v_w + v_{w+4}, then (t0 + t1) + (t2 + t3)).
// ------------------------------------------------------------------------------------------------
template <bool HP>
__global__ __launch_bounds__(512) void attn_pv_exact8_kernel(const float *__restrict__ att, int ld_att, int64_t head_stride, int D, int N,
                                                             int n_past, const float *__restrict__ vc, int n_ctx, float *__restrict__ ao,
                                                             int ldo, int8_t *__restrict__ oq, float *__restrict__ od,
                                                             float *__restrict__ os, uint16_t *__restrict__ oh) {
    extern __shared__ __attribute__((aligned(16))) float xs_[];
    constexpr int LD = XD_LD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qb = (int)gridDim.y - 1 - (int)blockIdx.y, hd = blockIdx.x, q0 = qb * 32;        // (heaviest query blocks first)
    const int i = lane & 31, h = lane >> 5;
    const int P = n_past + N;
    const int kend = min(P, n_past + min(q0 + 31, N - 1) + 1);
    const int np = P & ~31, nbody = min(np, (kend + 31) & ~31);
    const int npc = (nbody + XD_KT - 1) / XD_KT;
    const float *prow = att + hd * head_stride;
    const int rows_q = min(32, N - q0);
    const int s_rr = lane >> 3, s_kk = lane & 7;
    float *Ts = xs_ + 128 * LD;                              // [4 tiles][16 e][64 lanes]
    float *Os = Ts + 4 * 64 * 16;                            // [32 queries][33]
    auto own = [](int v) { asm volatile("" : "+v"(v)); return v; };
    typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
    // thread -> f32 tile: 4 float4, u = 0 .. 3: row 8 u + rr, keys 32 w + 4 kk .. + 3; -> compact P tile: 2 x 16 bytes, u = 0, 1: row 16 (w >> 2) + 8 u + rr,
    // keys 64 (w & 3) + 8 kk .. + 7 (a wave-load: 8 rows x one 128-byte line either way)
    struct Regs { v4u_ hp[HP ? 2 : 1]; float4 fp[HP ? 1 : 4]; float4 v[4]; } R;
    const int prow0 = 16 * (wave >> 2) + s_rr;               // compact: this lane's first P row
    float inv2[2];
    if constexpr (HP) {
#pragma unroll
        for (int u = 0; u < 2; ++u) inv2[u] = prow[(int64_t)(q0 + min(prow0 + 8 * u, rows_q - 1)) * ld_att + ld_att - 1];
    }
    auto rsrc = [&](const float *base, int rows, int row_stride) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)((unsigned)rows * (unsigned)row_stride * 4u), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsP = rsrc(prow + (int64_t)q0 * ld_att, rows_q, ld_att);
    auto rsv = [&](int dd) { return rsrc(vc + (int64_t)(hd * D + dd) * n_ctx, 32, n_ctx); };
    auto piece_len = [&](int c) { return min(XD_KT, nbody - c * XD_KT); };
    // parts 0 .. 5 of a piece (compact P: two P then four V; f32 P: four P + V pairs, parts 4 and 5 empty), -1: all
    auto load_piece = [&](__amdgpu_buffer_rsrc_t rsvv, int k0, auto part) __attribute__((always_inline)) {
        constexpr int PT = decltype(part)::value;
        if constexpr (HP) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (PT < 0 || PT == u)
                    R.hp[u] = __builtin_amdgcn_raw_buffer_load_b128(rsP, min(prow0 + 8 * u, rows_q - 1) * ld_att * 4 + (64 * (wave & 3) + 8 * s_kk) * 2, k0 * 2, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!(PT < 0 || PT == (HP ? 2 + u : u))) continue;
            const int row = 8 * u + s_rr, col = 32 * wave + 4 * s_kk;
            if constexpr (!HP) {
                const v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rsP, (min(row, rows_q - 1) * ld_att + col) * 4, k0 * 4, 0);
                R.fp[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
            const v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rsvv, (row * n_ctx + col) * 4, k0 * 4, 0);
            R.v[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        }
    };
    // pb: the buffer's first float (0 or 64 LD); valid body keys, zeros behind them up to the piece's 256
    const int st_lane = s_rr * LD + 32 * wave + 4 * s_kk, sth_lane = prow0 * LD + 64 * (wave & 3) + 8 * s_kk;
    auto store_piece = [&](int pb, int valid, auto part) __attribute__((always_inline)) {
        constexpr int PT = decltype(part)::value;
        if constexpr (HP) {
            float *const pd = xs_ + own(pb + sth_lane);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (PT < 0 || PT == u) {
                    float *d = pd + 8 * u * LD;
                    const bool in = 64 * (wave & 3) + 8 * s_kk < valid;
                    const unsigned w4[4] = {in ? R.hp[u].x : 0u, in ? R.hp[u].y : 0u, in ? R.hp[u].z : 0u, in ? R.hp[u].w : 0u};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        d[j] = __fmul_rn(__half2float(__ushort_as_half((unsigned short)(w4[j >> 1] >> (16 * (j & 1))))), inv2[u]);
                }
        }
        float *const fd = xs_ + own(pb + st_lane);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (PT < 0 || PT == (HP ? 2 + u : u)) {
                const bool in = 32 * wave + 4 * s_kk < valid;
                if constexpr (!HP) {
                    float *d = fd + 8 * u * LD;
                    d[0] = in ? R.fp[u].x : 0.f; d[1] = in ? R.fp[u].y : 0.f; d[2] = in ? R.fp[u].z : 0.f; d[3] = in ? R.fp[u].w : 0.f;
                }
                float *d = fd + 32 * LD + 8 * u * LD;
                d[0] = in ? R.v[u].x : 0.f; d[1] = in ? R.v[u].y : 0.f; d[2] = in ? R.v[u].z : 0.f; d[3] = in ? R.v[u].w : 0.f;
            }
    };
    // the chains: lane (i, h) of wave l reads row i, keys cs + 64 u + 32 h + 8 jj + l of the P tile and of the V tile (32 LD behind it)
    const int rd_lane = i * LD + 32 * h + wave;
    auto fetch = [&](float (&a)[2][4], float (&b)[2][4], int pb, int cs) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                a[u][jj] = xs_[pb + cs + 64 * u + 8 * jj];
                b[u][jj] = xs_[pb + 32 * LD + cs + 64 * u + 8 * jj];
            }
    };
    constexpr std::integral_constant<int, -1> whole_piece{};
    auto seq_next = [&](int &c, int &d0) { if (++c == npc) { c = 0; d0 += 32; } };
    int c = 0, d0 = 0, cur = 0;                               // the piece about to run, its buffer
    int cl = 0, dl = 0;                                       // the piece requested last
    load_piece(rsv(0), 0, whole_piece);
    store_piece(0, piece_len(0), whole_piece);
    seq_next(cl, dl);
    load_piece(rsv(min(dl, D - 32)), cl * XD_KT, whole_piece);
    lds_barrier();
    v16f acc[4] = {{}, {}, {}, {}};
    float a0[2][4], b0[2][4];
    fetch(a0, b0, own(rd_lane), 0);
    const int nleft = P - np;
    const bool left_visible = nleft > 0 && kend > np;
    while (d0 < D) {
        int c1 = c, d1 = d0;
        seq_next(c1, d1);
        const bool has1 = d1 < D;
        const int rd = own(rd_lane + cur * 64 * LD), rdn = own(rd_lane + (cur ^ 1) * 64 * LD);
        const int stb = (cur ^ 1) * 64 * LD;
        float a1[2][4], b1[2][4];
        fetch(a1, b1, rd, 128);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][jj], b0[u][jj], acc[jj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                    // (the stores stay behind trip 0)
        const int v1 = has1 ? piece_len(c1) : 0;
        seq_next(cl, dl);
        const __amdgpu_buffer_rsrc_t rsvn = rsv(min(dl, D - 32));
        auto trip1 = [&](auto... K) __attribute__((always_inline)) {
            auto one = [&](auto k) __attribute__((always_inline)) {
                constexpr int KK = decltype(k)::value, u = KK >> 2, jj = KK & 3;
                if constexpr (KK >= 2) {
                    store_piece(stb, v1, std::integral_constant<int, KK - 2>{});
                    load_piece(rsvn, cl * XD_KT, std::integral_constant<int, KK - 2>{});
                }
                acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][jj], b1[u][jj], acc[jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            (one(K), ...);
        };
        trip1(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
              std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{});
        lds_barrier();                                        // buffer cur is free, buffer cur ^ 1 complete
        fetch(a0, b0, rdn, 0);
        if (c1 == 0) {                                        // that was the feature block's last piece
            float *Ps = xs_ + cur * 64 * LD, *Vs = Ps + 32 * LD;      // (the leftovers go through the buffer just left)
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));                    // (the thread's indices taken afresh: nothing below is hoisted above the piece loop)
            const int tid = tid_, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
            v16f vt;                                          // v_l = (s0 + s1) + (s2 + s3)
#pragma unroll
            for (int e = 0; e < 16; ++e) vt[e] = __fadd_rn(__fadd_rn(acc[0][e], acc[1][e]), __fadd_rn(acc[2][e], acc[3][e]));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = v16f{};
            if (left_visible) {                              // the leftover keys [np, P) to columns 0.. of both tiles
                for (int idx = tid; idx < 32 * 32; idx += 512) {
                    const int r = idx >> 5, k = idx & 31, rc = min(r, rows_q - 1);
                    const float *row = prow + (int64_t)(q0 + rc) * ld_att;
                    float pv;
                    if constexpr (HP) pv = __fmul_rn(__half2float(reinterpret_cast<const __half *>(row)[min(np + k, P - 1)]), row[ld_att - 1]);
                    else pv = row[min(np + k, P - 1)];
                    Ps[r * LD + k] = r < rows_q && np + k < kend ? pv : 0.f;
                    const float vv = vc[(int64_t)(hd * D + d0 + r) * n_ctx + min(np + k, P - 1)];
                    Vs[r * LD + k] = np + k < kend ? vv : 0.f;
                }
            }
            // t_w = v_w + v_{w+4} (lo128 + hi128): waves 4 .. 7 hand their tiles over, waves 0 .. 3 add and publish t_w
            if (wave >= 4) {
                float *dst = Ts + (wave - 4) * 1024 + lane;
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[e * 64] = vt[e];
            }
            lds_barrier();
            if (wave < 4) {
                float *dst = Ts + wave * 1024 + lane;
#pragma unroll
                for (int e = 0; e < 16; ++e) vt[e] = __fadd_rn(vt[e], dst[e * 64]);
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[e * 64] = vt[e];     // (each lane rewrites the slots it just read: no barrier between)
            }
            lds_barrier();
            {
                // every wave finishes two of the sixteen query rows a lane holds (e = 2 wave, 2 wave + 1): (t0 + t1) + (t2 + t3), then the leftover keys
                const float *t0 = Ts + lane, *t1 = Ts + 1024 + lane, *t2 = Ts + 2048 + lane, *t3 = Ts + 3072 + lane;
                const float *vr = Vs + i * LD;
                float sv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = 2 * wave + j;
                    sv[j] = __fadd_rn(__fadd_rn(t0[e * 64], t1[e * 64]), __fadd_rn(t2[e * 64], t3[e * 64]));
                }
                if (nleft > 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int e = 2 * wave + j, ql = (e & 3) + 8 * (e >> 2) + 4 * h;
                        const float *pr = Ps + ql * LD;
                        float s = sv[j];
                        int k = 0;
                        for (; k + 8 <= nleft; k += 8)
                            for (int u = 0; u < 8; ++u)
                                if (np + k + u < kend) {
                                    const float prd = pr[k + u] * vr[k + u];
                                    s = s + prd;
                                }
                        if (nleft - k >= 4) {
                            for (int u = 0; u < 4; ++u)
                                if (np + k + u < kend) {
                                    const float prd = pr[k + u] * vr[k + u];
                                    s = s + prd;
                                }
                            k += 4;
                        }
                        for (; k < nleft; ++k)
                            if (np + k < kend) s = __fmaf_rn(pr[k], vr[k], s);
                        sv[j] = s;
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = 2 * wave + j, ql = (e & 3) + 8 * (e >> 2) + 4 * h, q = q0 + ql;
                    if (oq) Os[ql * 33 + i] = sv[j];
                    else if (q < N) ao[(int64_t)q * ldo + hd * D + d0 + i] = sv[j];
                }
            }
            if (oq) {
                lds_barrier();
                if (tid < 128) {
                    const int ql = tid >> 2, part = tid & 3, q = q0 + ql, KBo = ldo >> 5, kg = ((hd * D + d0) >> 3) + part;
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = q < N ? Os[ql * 33 + part * 8 + u] : 0.f;
                    if (q < ((N + 15) & ~15)) {
                        quantize_store_group(v, q, kg, KBo, 16, oq, od, os, oh);
                    } else if (oh) {                         // (columns [N16, N32) exist in the XH16 copy only)
                        unsigned char *dst = reinterpret_cast<unsigned char *>(oh) + ((((int64_t)(q >> 5) * KBo + (kg >> 2)) * 2 + (part >> 1)) * 64 + (q & 31)) * 16 + (part & 1) * 8;
                        *reinterpret_cast<uint2 *>(dst) = make_uint2(0, 0);
                        *reinterpret_cast<uint2 *>(dst + 512) = make_uint2(0, 0);
                    }
                }
            }
            if (has1) lds_barrier();                          // (the next piece's stores go into the buffer the leftovers were read from)
        }
        c = c1;
        d0 = d1;
        cur ^= 1;
    }
}

// hipErrorInvalidValue: shape outside these kernels' reach (head_dim not a multiple of 32 or > 128, unaligned rows): the caller
// (run_eval_kernels, model.cpp) then takes dot_f32_abt_exact
hipError_t attn_scores_exact(const float *qkv, int ldq, int D, int H, int N, int n_past, const float *kc, int ldk, float scale,
                             float *att, int ld_att, int64_t head_stride, hipStream_t st) {
    if (D % 32 != 0 || D > 128 || N < 1 || (ldq & 3) || (ldk & 3)) return hipErrorInvalidValue;
    const dim3 grid(H, (N + 31) / 32);
#define FL_XS(NST) hipLaunchKernelGGL((attn_scores_exact_kernel<NST, false>), grid, dim3(512), 0, st, qkv, ldq, N, n_past, kc, ldk, scale, att, ld_att, head_stride, nullptr, 0, 0)
    if (D == 32) FL_XS(1);
    else if (D == 64) FL_XS(2);
    else if (D == 96) FL_XS(3);
    else FL_XS(4);
#undef FL_XS
    return hipGetLastError();
}
// K.Q, scale, mask AND soft_max in one launch (contexts of up to 1024 keys: the 32 score rows of a workgroup wait in LDS); hipErrorInvalidValue:
// outside its reach -- the caller runs attn_scores_exact + softmax_rows
hipError_t attn_scores_softmax_exact(const float *qkv, int ldq, int D, int H, int N, int n_past, const float *kc, int ldk, float scale,
                                     float *att, int ld_att, int64_t head_stride, const uint16_t *exp_tab, hipStream_t st, bool compact) {
    const int P = n_past + N;
    if (!exp_tab || D % 32 != 0 || D > 128 || N < 1 || (ldq & 3) || (ldk & 3) || P > 1024 || (compact && (P < 4 || ld_att < P))) return hipErrorInvalidValue;
    const int PLD = ((P + 31) & ~31) + 1;
    const size_t lds = (size_t)32 * PLD * 4;
    // 131 KB of dynamic LDS must be asked for once per device: 0 = not tried, 1 = granted, -1 = refused (remembered and said ONCE: the caller's
    // two-launch path gives the same bits, and a refusal retried on every layer of every eval would be a silent cliff -- ADVICE r5)
    static std::atomic<int> attr_state[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidValue;
    int stt = attr_state[dev].load(std::memory_order_acquire);
    if (stt == 0) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        stt = attr_state[dev].load(std::memory_order_relaxed);
        if (stt == 0) {
            const size_t mx = (size_t)32 * 1025 * 4;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_scores_exact_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_scores_exact_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_scores_exact_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_scores_exact_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx);
            stt = e == hipSuccess ? 1 : -1;
            if (stt < 0) {
                (void)hipGetLastError();
                warn("device %d refuses %zu bytes of dynamic LDS (%s): the reference-order prefill attention runs K.Q and soft_max as two launches (same results)",
                     dev, mx, hipGetErrorString(e));
            }
            attr_state[dev].store(stt, std::memory_order_release);
        }
    }
    if (stt < 0) return hipErrorInvalidValue;
    const dim3 grid(H, (N + 31) / 32);
#define FL_XS(NST) hipLaunchKernelGGL((attn_scores_exact_kernel<NST, true>), grid, dim3(512), lds, st, qkv, ldq, N, n_past, kc, ldk, scale, att, ld_att, head_stride, exp_tab, PLD, compact ? 1 : 0)
    if (D == 32) FL_XS(1);
    else if (D == 64) FL_XS(2);
    else if (D == 96) FL_XS(3);
    else FL_XS(4);
#undef FL_XS
    return hipGetLastError();
}
int g_pv_waves = 8;      // waves per workgroup of the V.P kernel behind a deep context (8: attn_pv_exact8_kernel; 4: the round's first form; fl_debug_set(8, .))
hipError_t attn_pv_exact(const float *att, int ld_att, int64_t head_stride, int D, int H, int N, int n_past, const float *vc, int n_ctx,
                         float *ao, int ldo, hipStream_t st, const fl_qact *out, bool with_h16, bool compact) {
    if (D % 32 != 0 || D > 128 || N < 1 || (n_ctx & 3) || (out && (ldo & 31))) return hipErrorInvalidValue;
    if (compact && (n_past + N <= XA_KT || (ld_att & 3))) return hipErrorInvalidValue;      // (one piece: the f32 form; softmax_rows / attn_scores_softmax_exact take the same flag)
    const size_t lds = (size_t)(2 * 32 * XA_LD > 4 * 32 * XD_LD ? 2 * 32 * XA_LD : 4 * 32 * XD_LD) * 4 + 4 * 64 * 16 * 4 + 32 * 33 * 4;
    static bool attr_set[64] = {false};      // (per device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_pv_exact_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_pv_exact_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_pv_exact_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_pv_exact8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_pv_exact8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    int8_t *oq = out ? out->q : nullptr;
    float *od = out ? out->d : nullptr, *os = out ? out->s : nullptr;
    uint16_t *oh = out && with_h16 ? out->h16 : nullptr;
    if (n_past + N <= XA_KT)                                  // (the eight-wave form on short contexts: 33.05 / 32.83 against 33.05 ms per eval -- the same)
        hipLaunchKernelGGL(attn_pv_exact_kernel<true>, dim3(H, (N + 31) / 32), dim3(256), lds, st, att, ld_att, head_stride, D, N, n_past, vc, n_ctx, ao, ldo,
                           oq, od, os, oh);
    else if (g_pv_waves >= 8 && compact)
        hipLaunchKernelGGL((attn_pv_exact8_kernel<true>), dim3(H, (N + 31) / 32), dim3(512), lds, st, att, ld_att, head_stride, D, N, n_past, vc, n_ctx, ao,
                           ldo, oq, od, os, oh);
    else if (g_pv_waves >= 8)
        hipLaunchKernelGGL((attn_pv_exact8_kernel<false>), dim3(H, (N + 31) / 32), dim3(512), lds, st, att, ld_att, head_stride, D, N, n_past, vc, n_ctx, ao,
                           ldo, oq, od, os, oh);
    else if (compact)
        hipLaunchKernelGGL((attn_pv_exact_kernel<false, true>), dim3(H, (N + 31) / 32), dim3(256), lds, st, att, ld_att, head_stride, D, N, n_past, vc, n_ctx, ao,
                           ldo, oq, od, os, oh);
    else
        hipLaunchKernelGGL(attn_pv_exact_kernel<false>, dim3(H, (N + 31) / 32), dim3(256), lds, st, att, ld_att, head_stride, D, N, n_past, vc, n_ctx, ao, ldo,
                           oq, od, os, oh);
    return hipGetLastError();
}

}  // namespace fl
