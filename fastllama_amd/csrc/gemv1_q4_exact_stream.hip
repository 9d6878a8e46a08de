// gemv1_q4_exact_stream.hip -- the reference-order ("exact") Q4 x Q8_0 matmul for N = 1 (decode), round 6 form: ONE WAVE PER ROW GROUP.
//
// What must be reproduced (ggml_vec_dot_q4_{0,1}_q8_0, AVX2 branch, /root/reference/lib/ggml.c:2445-2487, :2639-2689): per output row 8 f32
// accumulators, accumulator j taking  acc_j = fma(d_w * d_x, float(sum of the products of elements 4j..4j+3), acc_j)  block after block,
// then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ the scalar chain summs = fma(m_w, s_x, summs) for Q4_1].
//
// Round 4's kernel (gemv1_q4_exact_llc.hip) splits a 16-row group along K over the 4 (8) waves of a workgroup: every wave converts its slice to
// lane sums (96 registers) while all slices are in flight, then the chains run slice after slice through LDS -- a load phase, a sum phase and a
// chain phase per workgroup, one activation prologue per ROW GROUP (1376 of them for LLaMA-7B's w1|w3 on 768 residency slots: two rounds of
// workgroups, profiles/r05_decode_timeline.md).  That is the right shape for a matrix of one row group per CU (wo, w2).  For matrices of MANY row
// groups (wq|wk|wv, w1|w3, the lm-head; every matmul of LLaMA-65B) this file turns the split round:
//   * a wave owns a row group for the WHOLE of K.  lane = (row r, k-group g) keeps the chains 2g, 2g+1 of its row in two registers (the QWD
//     copy, q4_layout.h, exactly as the llc kernel reads it) and walks the block quads in order: wait for a quad, 3 unpack + 2 v_dot4 + 2 cvt
//     + scale product + 2 fma per block, request the quad U ahead into the registers just freed.  Nothing but the loads in flight is kept
//     (U x 5 registers): no lane-sum arrays, no chain hand-off, no barrier after the prologue; the arithmetic runs UNDER the stream;
//   * a workgroup = 4 such waves = 4 consecutive row groups sharing ONE activation prologue (rms_norm / Q8_0 in LDS, gemv_prologue.h): 344
//     prologues instead of 1376 for w1|w3, all workgroups co-resident (<= 2 per CU), any K (the loop's trip count) -- no multi-pass rows;
//   * woven w1|w3 (groups 4b .. 4b+3 = w1 | w3 rows of features 32b .. 32b+15, then of 32b+16 .. 32b+31): the workgroup owns a whole
//     32-feature block, so it forms silu(w1 x) * (w3 x) (ggml_silu + ggml_mul, lib/llama.cpp:428-431) AND the block's Q8_0 form -- the operand
//     of the w2 matmul (quantize_row_q8_0, lib/ggml.c:1299-1441) -- itself (EPI = 2): w2's prologue is a 13.8 KB copy instead of an f32
//     round trip + quantization.  EPI = 1 keeps the f32 features (tensor-parallel fold: the exchange moves f32 slices).
// Q4_0 bookkeeping as everywhere: unpacked values are 16 (nib - 8), the stored scale is d / 16.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <cstdlib>
#include <algorithm>
#include "q4_device.h"
#include "q4_kernels.h"
#include "gemv_prologue.h"
#include "tp_tail.h"

#pragma clang fp contract(off)

namespace fl {

typedef unsigned int sv4u __attribute__((ext_vector_type(4)));

template <int SRC>
__device__ __forceinline__ float sq_bcast(float v) {      // value of lane (quad base + SRC) of every quad
    return dpp_f32<SRC | (SRC << 2) | (SRC << 4) | (SRC << 6)>(v);
}

#ifdef LLC_TIMING   // development build only (scripts/dev/stream_timeline.py): per-workgroup wall-clock stamps of every launch since the last reset
constexpr unsigned ST_TL_CAP = 1u << 16;
constexpr int ST_TL_W = 12;       // int64 per record: entry, loads issued, prologue done, loop end of waves 0..3, chains reduced + stored, end, id, spare x2
__device__ long long st_tl[(size_t)ST_TL_CAP * ST_TL_W];
__device__ unsigned st_tl_cur;
#define ST_DECL long long tl_[4] = {0, 0, 0, 0}; __shared__ long long tlw_[4]
#define ST_STAMP(k) do { tl_[k] = wall_clock64(); } while (0)
#else
#define ST_DECL do {} while (0)
#define ST_STAMP(k) do {} while (0)
#endif

// K / V history prefetch (the wq|wk|wv launch of a decode layer): the attention that follows reads, per head, the first min(pos, 256) rows of the K cache and
// D rows of the V cache -- from HBM, 1.5-2 us behind its requests, while its 32 workgroups have nothing else to do.  Workgroup b of THIS launch runs
// on XCD b % 8 and head h's attention workgroup on XCD h % 8: b touches one dword of every 128-byte line of the heads of its XCD class, right behind its
// own weight requests; the lines wait in that XCD's L2 (which outlives the launch boundary for data nobody writes).  A hint: placement is what the
// dispatcher does, not a contract -- a miss costs what it cost before.
struct StreamPrefetch {
    const float *kc, *vc;          // this layer's K cache [n_ctx][E] and transposed V cache [E][n_ctx]
    const int *pos;                // device: the position (rows 0 .. pos - 1 are the history)
    int E, D, H, n_ctx;
};
thread_local StreamPrefetch stream_pending_prefetch = {nullptr, nullptr, nullptr, 0, 0, 0, 0};
int g_stream_helpers = 1;        // fl_debug_set(9, 0 / 1): prologue-only waves in front of the streaming ones (launches of at most two workgroups per CU)

// EPI 0: y[row] = dot (+ resid[row]); 1: woven w1|w3 -> f32 silu(w1 x) * (w3 x); 2: woven w1|w3 -> the Q8_0 blocks (QA1 planes) of those features
// nw <= MAXW: the waves of a workgroup that stream a row group each (workgroup b: row groups b nw .. b nw + nw - 1); the workgroup has 64 MAXW threads,
// the others only help with the prologue.  Four by default; launches that would not be resident at four take one workgroup per CU with equal shares
// (launch_stream below).
template <int TYPE, int PRO, int EPI, int U, int MAXW, int TAIL>
// (second bound: waves per SIMD -- two four-wave or two eight-wave workgroups per CU where the launch has more workgroups than CUs)
__global__ __launch_bounds__(64 * MAXW, MAXW == 4 ? 2 : (MAXW == 8 && U == 8) || MAXW == 16 ? 4 : 1) void gemv1_q4_exact_stream_kernel(
    int M, int groups, int nw, int KB, const uint32_t *__restrict__ qwd, const float *__restrict__ dW, const float *__restrict__ xf,
    const void *__restrict__ aux, const float *__restrict__ mW, const int8_t *__restrict__ xq, const float *__restrict__ xd,
    const float *__restrict__ xs, float *__restrict__ y, const float *__restrict__ resid, float *__restrict__ ynorm,
    const uint16_t *__restrict__ silu_tab, int8_t *__restrict__ oq, float *__restrict__ od, float *__restrict__ os,
    const TpTail *__restrict__ tt, const StreamPrefetch pf) {
    constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    constexpr int NT = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __shared__ double sh[4];
    __shared__ float ex[MAXW][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NQ = (KB + 3) >> 2, NR = (NQ + U - 1) / U, NQP = NR * U;          // quads of a row, rounds of U, quads incl. the padding of the last round
    // LDS: [LX: the Q8_0 activation as the lanes read it, [NQP][4 k-groups][4 blocks][8 B: e0..e3 | e4..e7]] [d [4 NQP]] [s [4 NQP]]
    unsigned char *lx = gsm;
    float *ld_ = reinterpret_cast<float *>(gsm + (size_t)NQP * 128);
    float *ls_ = ld_ + 4 * NQP;

    ST_DECL;
    ST_STAMP(0);
    GP_DECL(PRO);
    GemvPrologue<PRO, NT, true>::issue(pv, pw, psl, psb, xf, aux, KB, 0);

    // The LAST nw waves of the workgroup stream a row group each (streaming index sw); the waves in front of them only work on the prologue.  With four
    // such waves (MAXW = nw + 4: the rms_norm prologue is written for the first 256 threads) the prologue is computed by waves whose memory queue is
    // not full of weight requests: a streaming wave issues its 32 requests for ~3.5 us (the queue fills, the first bytes have to return) and only then
    // reached the norm -- prologue done 4.7 us into an 11 us launch of LLaMA-7B's wq|wk|wv (profiles/r06_decode_timeline.md).
    const int sw = wave - (MAXW - nw);
    const bool active = sw >= 0 && (int)blockIdx.x * nw + sw < groups;       // (wave-uniform) this wave streams a row group
    const int grp = min((int)blockIdx.x * nw + max(sw, 0), groups - 1);
    const bool live = active;
    const int r = lane >> 2, g = lane & 3;
    // buffer loads: the wave-uniform part (row group, quad) in the scalar offset, one lane offset register per plane; bytes past a plane read as zero
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(qwd), 0, (int)((uint32_t)groups * (uint32_t)NQ * 1024u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dW), 0, (int)((uint32_t)groups * (uint32_t)KB * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q41 ? mW : dW), 0, (int)((uint32_t)groups * (uint32_t)KB * 64u), 0x00020000);
    const int voff_w = lane * 16;
    const int voff_d = (g * 16 + r) * 4;                                     // lane (row, g) fetches the scale of block 4q + g of its row
    // (a partial last quad reads past the row's blocks: the next row group's first scales, or zeros behind the plane -- FINITE numbers that
    //  meet d_x = 0 and zero quants: the block adds +0 to a chain that can never hold -0; the same holds for the padding quads of the last round)
    sv4u w[U];
    float dw[U], mw[U];
    auto req_scale = [&](int i, int q) __attribute__((always_inline)) {
        const int sd = (grp * KB + 4 * min(q, NQ - 1)) * 64;
        dw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, voff_d, sd, 2));
        if (Q41) mw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff_d, sd, 2));
    };
    auto req_nib = [&](int i, int q) __attribute__((always_inline)) {
        w[i] = __builtin_bit_cast(sv4u, __builtin_amdgcn_raw_buffer_load_b128(rW, voff_w, (grp * NQ + min(q, NQ - 1)) * 1024, 2 /* nt: read once per token */));
    };
    // PRO = 0: the Q8_0 activation (QA1 in HBM) is requested before the weight stream (loads return in order)
    constexpr int XIT = 4;
    uint2 xa_[PRO == 0 ? XIT : 1];
    float xd_[PRO == 0 ? 2 : 1], xs_[PRO == 0 ? 2 : 1];
    if constexpr (PRO == 0) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = threadIdx.x + it * NT;
            xa_[it] = reinterpret_cast<const uint2 *>(xq)[i < KB * 4 ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x + it * NT;
            xd_[it] = xd[i < KB ? i : 0];
            xs_[it] = Q41 ? xs[i < KB ? i : 0] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float rsd = 0.f;
    if constexpr (EPI == 0) {
        if (resid && active) rsd = resid[min(grp * 16 + r, M - 1)];         // (wave-uniform condition, clamped address)
    }
    // the first U quads, each as (scale, nibbles) -- the order the loop re-requests them in, so that the compiler's count of the loads in flight
    // (s_waitcnt vmcnt) is the same at the loop's entry and at its back edge: with all scales ahead of all nibbles it waited, every round, as if
    // only half of them were outstanding (seen in the ISA).  The scheduling barriers keep the pairs in program order.
    // (ALL of them before the prologue.  Holding most of them back until the activation has arrived -- the last workgroups' prologues sit behind the
    //  27 MB of requests the launch starts with: done 9.4 us into a 14.8 us launch of LLaMA-7B's w1|w3 -- brings the prologues forward (5.5 us) and
    //  costs more than it gains, the stream starting late: 589 / 566 / 585 tok/s with 4 / 8 / 2 quads up front against 617.  Also measured and not kept,
    //  profiles/r06_decode_exact.md: a short s_sleep between the activation's requests and the weights' (0.2 us: no change, 0.5 us: -2 %); the next quad's
    //  activation read from LDS one quad ahead (no change at 16 quads in flight, -8 % at 8); the two chains as one v_pk_fma_f32 with the lane sums
    //  converted by a v_dot4 onto the bits of 1.5 x 2^23 and a packed subtract (-22 % VALU instructions in the loop, bit-identical, no change in tok/s).)
    if (active) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            req_scale(i, i);
            req_nib(i, i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the K / V history of the attention that follows: one dword per 128-byte line, for the heads of this workgroup's XCD class
    // (two loads per thread at most, requested and left alone: nothing waits for them before the kernel's end -- a loop would wait for each, and
    //  with it for every weight request issued before it)
    float pf0 = 0.f, pf1 = 0.f;
    if constexpr (EPI == 0 && PRO == 1) {
        if (pf.kc) {
            const int pos = min(*pf.pos, 256), cls = (int)blockIdx.x & 7, ncls_wg = ((int)gridDim.x + 7 - cls) >> 3;
            const int hpc = (pf.H + 7 - cls) >> 3;                            // heads h = cls, cls + 8, ...
            const int klines = pf.D >> 5, vlines = (pos + 31) >> 5;          // 128-byte lines per K row of a head / per V row's history
            // (K rows only: 618 / 394 tok/s at 7B / 13B against 619 / 397 with both and 617 / 385 without the hint, profiles/r06_decode_exact.md)
            const int per_head = pos * klines + pf.D * vlines, total = hpc * per_head;
            auto line = [&](int i) __attribute__((always_inline)) -> const float * {
                i = min(i, total - 1);
                const int hh = i / per_head, j = i - hh * per_head, h = cls + 8 * hh;
                if (j < pos * klines) return pf.kc + (size_t)(j / klines) * pf.E + h * pf.D + (j % klines) * 32;
                const int jj = j - pos * klines;
                return pf.vc + (size_t)(h * pf.D + jj / vlines) * pf.n_ctx + (jj % vlines) * 32;
            };
            if (total > 0) {
                const int i0 = ((int)blockIdx.x >> 3) * (64 * MAXW) + (int)threadIdx.x, stride = ncls_wg * 64 * MAXW;
                pf0 = *line(i0);
                pf1 = *line(i0 + stride);
            }
        }
    }
    ST_STAMP(1);
    // blocks past K (the partial last quad, the padding quads): d_x = s_x = 0, zero quants (written before the prologue's closing barrier)
    for (int i = KB * 4 + (int)threadIdx.x; i < NQP * 16; i += NT) {
        const int b = i >> 2, gg = i & 3;
        *reinterpret_cast<uint2 *>(lx + (((b >> 2) * 4 + gg) * 4 + (b & 3)) * 8) = make_uint2(0, 0);
        if (gg == 0) { ld_[b] = 0.f; ls_[b] = 0.f; }
    }
    if constexpr (PRO != 0) {
        GemvPrologue<PRO, NT, true>::finish(pv, pw, psl, psb, xf, aux, KB, 0, reinterpret_cast<int8_t *>(lx), ld_, ls_, sh, ynorm, blockIdx.x == 0);
    } else {                                          // QA1 in HBM (k-group bytes e0,e2,e4,e6 | e1,e3,e5,e7): re-laid on the way
        auto put = [&](int i, uint2 lh) __attribute__((always_inline)) {
            const int b = i >> 2, gg = i & 3;
            *reinterpret_cast<uint2 *>(lx + (((b >> 2) * 4 + gg) * 4 + (b & 3)) * 8) =
                make_uint2(__builtin_amdgcn_perm(lh.y, lh.x, 0x05010400u), __builtin_amdgcn_perm(lh.y, lh.x, 0x07030602u));
        };
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = threadIdx.x + it * NT;
            if (i < KB * 4) put(i, xa_[it]);
        }
        for (int i = threadIdx.x + XIT * NT; i < KB * 4; i += NT) put(i, reinterpret_cast<const uint2 *>(xq)[i]);      // (rows beyond K = 8192: the rest)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x + it * NT;
            if (i < KB) { ld_[i] = xd_[it]; ls_[i] = xs_[it]; }
        }
        for (int i = threadIdx.x + 2 * NT; i < KB; i += NT) { ld_[i] = xd[i]; ls_[i] = Q41 ? xs[i] : 0.f; }
        __syncthreads();
    }

    ST_STAMP(2);
    // ---- the stream: quad after quad, the two chains of this lane in block order
    const uint32_t m8 = 0xF0F0F0F0u;
    float a0 = 0.f, a1 = 0.f, summs = 0.f;
    auto consume = [&](int i, int q) __attribute__((always_inline)) {
        const uint4 x01 = *reinterpret_cast<const uint4 *>(lx + ((size_t)q * 4 + g) * 32);
        const uint4 x23 = *reinterpret_cast<const uint4 *>(lx + ((size_t)q * 4 + g) * 32 + 16);
        const float4 dx4 = *reinterpret_cast<const float4 *>(ld_ + 4 * q);
        const uint32_t wv[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
        const uint32_t xa[4] = {x01.x, x01.z, x23.x, x23.z}, xb[4] = {x01.y, x01.w, x23.y, x23.w};
        const float dxv[4] = {dx4.x, dx4.y, dx4.z, dx4.w};
        const float dwb[4] = {sq_bcast<0>(dw[i]), sq_bcast<1>(dw[i]), sq_bcast<2>(dw[i]), sq_bcast<3>(dw[i])};
        float sxv[4] = {0.f, 0.f, 0.f, 0.f}, msb[4] = {0.f, 0.f, 0.f, 0.f};
        if (Q41) {
            msb[0] = sq_bcast<0>(mw[i]); msb[1] = sq_bcast<1>(mw[i]); msb[2] = sq_bcast<2>(mw[i]); msb[3] = sq_bcast<3>(mw[i]);
            const float4 sx4 = *reinterpret_cast<const float4 *>(ls_ + 4 * q);
            sxv[0] = sx4.x; sxv[1] = sx4.y; sxv[2] = sx4.z; sxv[3] = sx4.w;
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            uint32_t wa, wb;
            if (TYPE == FL_TYPE_Q4_0) { wa = (wv[blk] << 4) & m8; wb = wv[blk] & m8; }     // 16 (nib - 8): elements 0..3 | 4..7
            else { wa = wv[blk] & 0x0F0F0F0Fu; wb = (wv[blk] >> 4) & 0x0F0F0F0Fu; }
            const float dd = __fmul_rn(dwb[blk], dxv[blk]);                 // rn(d_w d_x); a block past K: d_x = 0
            const float f0 = (float)__builtin_amdgcn_sdot4((int)wa, (int)xa[blk], 0, false);
            const float f1 = (float)__builtin_amdgcn_sdot4((int)wb, (int)xb[blk], 0, false);
            a0 = __fmaf_rn(dd, f0, a0);
            a1 = __fmaf_rn(dd, f1, a1);
            if (Q41) summs = __fmaf_rn(msb[blk], sxv[blk], summs);          // (a block past K: s_x = 0, m_w finite)
        }
    };
    if (active) {
#pragma unroll 1
        for (int rd = 0; rd < NR - 1; ++rd) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int q = rd * U + i;
                consume(i, q);
                req_scale(i, q + U);                                        // (the last round's padding quads: a cache-hot dummy, clamped; they meet d_x = 0)
                req_nib(i, q + U);
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) consume(i, (NR - 1) * U + i);
    }

#ifdef LLC_TIMING
    asm volatile("" :: "v"(a0), "v"(a1));
    if (lane == 0 && sw >= 0 && sw < 4) tlw_[sw] = wall_clock64();
#endif
    // ---- the row: ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) over the quad of lanes that holds it (lane g: accumulators 2g, 2g+1)
    float e = a0, o = a1;
    e = __fadd_rn(e, dpp_f32<DPP_XOR2>(e));      // a0+a4 | a2+a6
    o = __fadd_rn(o, dpp_f32<DPP_XOR2>(o));      // a1+a5 | a3+a7
    e = __fadd_rn(e, dpp_f32<DPP_XOR1>(e));      // (a0+a4)+(a2+a6)
    o = __fadd_rn(o, dpp_f32<DPP_XOR1>(o));      // (a1+a5)+(a3+a7)
    float v = __fadd_rn(e, o);
    if (Q41) v = __fadd_rn(v, summs);
    // (with a tail: written through to memory, and to every peer's region -- tp_put, tp_tail.h)
    auto put_y = [&](float *p, float val) __attribute__((always_inline)) {
        if constexpr (TAIL) tp_put(tt, p, val);
        else *p = val;
    };
    if constexpr (EPI == 0) {
        const int row = grp * 16 + r;
        if (g == 0 && row < M && live) {
            if (resid) v = __fadd_rn(v, rsd);
            put_y(y + row, v);
        }
    } else {
        if (g == 0 && sw >= 0) ex[sw][r] = v;
        __syncthreads();
        // the workgroup's nw / 4 blocks of 32 features: the first wave of a block's four finishes it
        const int blk = (int)blockIdx.x * (nw >> 2) + (max(sw, 0) >> 2);
        if (sw >= 0 && (sw & 3) == 0 && active) {                            // feature f of the block in lanes f and f + 32 (both halves compute, the lower stores)
            const int f = lane & 31, p2 = sw + (f >> 4) * 2;
            const float h1 = ex[p2][f & 15], h3 = ex[p2 + 1][f & 15];
            const uint16_t hx = __half_as_ushort(__float2half_rn(h1));                // GGML_FP32_TO_FP16
            const float sl = __half2float(__ushort_as_half(silu_tab[hx]));           // table_silu_f16
            const float h = __fmul_rn(sl, h3);                                        // ggml_mul(silu, tmp)
            if constexpr (EPI == 1) {
                if (lane < 32) put_y(y + blk * 32 + f, h);
            } else {
                // quantize_row_q8_0 of the block (lib/ggml.c:1299-1441, AVX2: amax, d = amax / 127, id = 127 / amax, round to nearest even)
                const float amax = wave_max_f32(fabsf(h));
                const float dq = __fdiv_rn(amax, 127.0f);
                const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
                const int qi = (int)rintf(__fmul_rn(h, id));
                int isum = quad_sum_i32(qi);
                isum += dpp_i32<DPP_HALF_MIRROR>(isum);
                isum += dpp_i32<DPP_MIRROR>(isum);
                isum += __shfl_xor(isum, 16);                                         // (32 features; the upper half of the wave holds the same 32)
                if (lane < 32) {
                    const int i8 = f & 7;                                             // QA1: k-group bytes e0,e2,e4,e6 | e1,e3,e5,e7 (q4_layout.h)
                    oq[(size_t)blk * 32 + (f & 24) + ((i8 & 1) * 4 + (i8 >> 1))] = (int8_t)qi;
                    if (lane == 0) {
                        od[blk] = dq;
                        os[blk] = __fmul_rn(dq, (float)isum);
                    }
                }
            }
        }
    }
#ifdef LLC_TIMING
    ST_STAMP(3);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s_ = atomicAdd(&st_tl_cur, 1u);
        if (s_ < ST_TL_CAP) {
            long long *rec = st_tl + (size_t)s_ * ST_TL_W;
            rec[0] = tl_[0]; rec[1] = tl_[1]; rec[2] = tl_[2];
            for (int k_ = 0; k_ < 4; ++k_) rec[3 + k_] = tlw_[k_];
            rec[7] = tl_[3]; rec[8] = wall_clock64();
            rec[9] = ((long long)(PRO * 100 + EPI * 10 + (U == 16)) << 32) | blockIdx.x;      // (stamps of the first four waves only)
        }
    }
#endif
    asm volatile("" :: "v"(pf0), "v"(pf1));                   // (the prefetch loads are waited for here at the latest; nothing uses them)
    if constexpr (TAIL) tp_tail<false, false, true>(tt);
}
#ifdef LLC_TIMING
// reset != 0: empty the ring; else copy up to max_rec records of ST_TL_W x int64 and return how many there are
extern "C" __attribute__((visibility("default"))) int fl_debug_stream_timeline(long long *out, int max_rec, int reset) {
    unsigned n = 0;
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(st_tl_cur), &n, sizeof n);
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(st_tl_cur), sizeof n) != hipSuccess) return -1;
    if (n > ST_TL_CAP) n = ST_TL_CAP;
    if ((int)n > max_rec) n = (unsigned)max_rec;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(st_tl), sizeof(long long) * ST_TL_W * (size_t)n) != hipSuccess) return -1;
    return (int)n;
}
#endif

// fl_debug_set(7, n) (tests): streaming waves per workgroup of the rms_norm forms, 0 = automatic
int g_stream_force_nw = 0;

// false: no QWD copy, or a shape outside this form's reach -> the caller takes the llc kernel
template <int TYPE, int PRO, int EPI>
static bool launch_stream(const fl_qtensor &W, const fl_qact *xq, float *y, hipStream_t st, const float *resid, const float *xf, const void *aux,
                          float *ynorm, const uint16_t *silu_tab, const fl_qact *out, bool take_tail) {
    if (!W.qwd) return false;
    const int KB = W.KB, NQ = (KB + 3) / 4, groups = W.M16 / 16;
    if (groups < 1 || KB < 1) return false;
    if (EPI != 0 && (groups % 4 != 0 || W.M != W.M16)) return false;          // a block = the w1 | w3 rows of 32 whole features = four row groups
    if (PRO == 1 && W.K > 8192) return false;                                 // (the rms_norm prologue keeps the row in registers: 256 threads x 32)
    if (qwd_bytes(W) >= (1ull << 31) || (size_t)W.M16 * (size_t)KB * 4 >= (1ull << 31)) return false;      // 32-bit buffer offsets
    static const int n_cus = [] { int d = 0, c = 0; return (hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && c > 0) ? c : 256; }();
    // Row groups per workgroup: four, as long as that is at most two workgroups per CU (all resident at 16 quads in flight).  Measured against ONE
    // workgroup per CU with the same number of row groups in each (profiles/r06_decode_exact.md): LLaMA-7B's wq|wk|wv as 256 x 3 instead of 192 x 4
    // 10.1 against 9.2 us, w1|w3 as 172 x 8 instead of 344 x 4 15.4 against 15.5, 13B 374 against 390 tok/s -- four it is.  Beyond two workgroups per
    // CU (LLaMA-65B's w1|w3: 688 of four) the launch IS one workgroup per CU with equal shares (230 x 12: 98 -> 103 tok/s).
    // (The Q8_0 / plain-operand prologues are written for 256 threads: always four -- no LLaMA matrix takes them.)
    int nw = 4;
    if (PRO == 1) {
        const int units = EPI == 0 ? groups : groups / 4, per = EPI == 0 ? 1 : 4;
        if ((groups + 3) / 4 > 2 * n_cus) nw = std::max(4, per * ((units + n_cus - 1) / n_cus));
        if (g_stream_force_nw > 0) nw = EPI == 0 ? g_stream_force_nw : (g_stream_force_nw + 3) / 4 * 4;
        nw = std::min(nw, 12);
    }
    int maxw = nw <= 4 ? 4 : nw <= 8 ? 8 : 12;
    // four prologue-only waves in front of four streaming ones, where every workgroup has a CU of its own anyway (wq|wk|wv: 192 workgroups)
    // (more workgroups than CUs: two eight-wave workgroups per CU at 8 quads in flight, 122-128 registers)
    if (g_stream_helpers && PRO == 1 && nw == 4 && (groups + 3) / 4 <= 2 * n_cus) maxw = 8;
    else if (g_stream_helpers && PRO == 1 && nw > 8) maxw = 16;            // (one workgroup per CU with up to twelve streaming waves: LLaMA-65B's w1|w3)
    // quads in flight per wave: 16 (20 KB per wave, ~170 registers) when that divides the row (K = 4096, 8192) and the workgroup has at most 8 waves
    // (two per SIMD); else 8
    // ... and with the prologue-only waves: 8 -- the streaming waves reach the barrier behind the prologue sooner (wq|wk|wv: 1.529 -> 1.507 ms per token;
    // 4 quads: 1.60)
    const bool helpers = maxw == 8 && nw == 4;
    const bool u16 = (NQ % 16 == 0 || NQ % 16 >= 13) && maxw <= 8 && !helpers;
    const int U = u16 ? 16 : 8, NQP = (NQ + U - 1) / U * U;
    const size_t lds = (size_t)NQP * 160;
    if (lds > 60 * 1024) return false;
    const TpTail *tt = take_tail ? tp_take_tail() : nullptr;
    StreamPrefetch pf = {nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (PRO == 1 && EPI == 0) { pf = stream_pending_prefetch; stream_pending_prefetch.kc = nullptr; }      // (offered by the model for exactly this launch)
    const dim3 grid((groups + nw - 1) / nw), block(64 * maxw);
#define FL_ST_GO(UU, MW, TL)                                                                                                              \
    hipLaunchKernelGGL((gemv1_q4_exact_stream_kernel<TYPE, PRO, EPI, UU, MW, TL>), grid, block, lds, st, W.M, groups, nw, KB, W.qwd, W.d, xf, \
                       aux, W.m, xq ? xq->q : nullptr, xq ? xq->d : nullptr, xq ? xq->s : nullptr, y, resid, ynorm, silu_tab,              \
                       out ? out->q : nullptr, out ? out->d : nullptr, out ? out->s : nullptr, tt, pf)
#define FL_ST_TL(UU, MW)                                                                                                                  \
    do {                                                                                                                                  \
        if constexpr (EPI == 2) FL_ST_GO(UU, MW, 0);          /* (no exchange of Q8_0 blocks: the fold path keeps the f32 features) */      \
        else if (tt) FL_ST_GO(UU, MW, 1);                                                                                                 \
        else FL_ST_GO(UU, MW, 0);                                                                                                         \
    } while (0)
    if constexpr (PRO == 1) {
        if (maxw == 4) { if (u16) FL_ST_TL(16, 4); else FL_ST_TL(8, 4); }
        else if (maxw == 8) { if (u16) FL_ST_TL(16, 8); else FL_ST_TL(8, 8); }
        else if (maxw == 12) FL_ST_TL(8, 12);
        else FL_ST_TL(8, 16);
    } else {
        if (u16) FL_ST_TL(16, 4); else FL_ST_TL(8, 4);
    }
#undef FL_ST_TL
#undef FL_ST_GO
    return true;
}

// Which matrices take this form: those with three row groups per CU or more (wq|wk|wv, w1|w3, the lm-head of every LLaMA size); a matrix of one or
// two row groups per CU (wo, w2) keeps the K-sliced workgroups of the llc kernel -- a wave per SIMD or fewer would run a whole row's arithmetic
// alone and keep too few bytes in flight (LLaMA-65B's w2: 512 waves x 10 KB).  g_stream_min_groups (fl_debug_set(6, n), tests) overrides the bound.
int g_stream_min_groups = -1;
static int stream_min_groups() {
    if (g_stream_min_groups >= 0) return g_stream_min_groups;
    static const int v = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        return 3 * cus;
    }();
    return v;
}
static bool stream_wanted(const fl_qtensor &W) { return W.M16 / 16 >= stream_min_groups(); }

void gemv1_stream_offer_kv_prefetch(const float *kc, const float *vc, const int *pos_dev, int E, int D, int H, int n_ctx) {
    stream_pending_prefetch = StreamPrefetch{kc, vc, pos_dev, E, D, H, n_ctx};
}
#define FL_TYPED(CALL0, CALL1) (W.type == FL_TYPE_Q4_0 ? (CALL0) : (CALL1))
bool gemv1_stream(const fl_qtensor &W, const fl_qact &xq, float *y, hipStream_t st, const float *resid) {
    if (!stream_wanted(W)) return false;
    return FL_TYPED((launch_stream<FL_TYPE_Q4_0, 0, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr, nullptr, nullptr, true)),
                    (launch_stream<FL_TYPE_Q4_1, 0, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr, nullptr, nullptr, true)));
}
bool gemv1_stream_norm(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st) {
    if (!stream_wanted(W)) return false;
    return FL_TYPED((launch_stream<FL_TYPE_Q4_0, 1, 0>(W, nullptr, y, st, nullptr, x, norm_w, ynorm, nullptr, nullptr, true)),
                    (launch_stream<FL_TYPE_Q4_1, 1, 0>(W, nullptr, y, st, nullptr, x, norm_w, ynorm, nullptr, nullptr, true)));
}
bool gemv1_stream_quant(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st) {
    if (!stream_wanted(W)) return false;
    return FL_TYPED((launch_stream<FL_TYPE_Q4_0, 3, 0>(W, nullptr, y, st, resid, x, nullptr, nullptr, nullptr, nullptr, true)),
                    (launch_stream<FL_TYPE_Q4_1, 3, 0>(W, nullptr, y, st, resid, x, nullptr, nullptr, nullptr, nullptr, true)));
}
bool gemv1_stream_norm_silu(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act, hipStream_t st) {
    if (!stream_wanted(W)) return false;
    return FL_TYPED((launch_stream<FL_TYPE_Q4_0, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, silu_tab, nullptr, true)),
                    (launch_stream<FL_TYPE_Q4_1, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, silu_tab, nullptr, true)));
}
bool gemv1_stream_norm_silu_q8(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, const fl_qact &out, hipStream_t st) {
    if (!stream_wanted(W)) return false;
    return FL_TYPED((launch_stream<FL_TYPE_Q4_0, 1, 2>(W, nullptr, nullptr, st, nullptr, x, norm_w, nullptr, silu_tab, &out, false)),
                    (launch_stream<FL_TYPE_Q4_1, 1, 2>(W, nullptr, nullptr, st, nullptr, x, norm_w, nullptr, silu_tab, &out, false)));
}
#undef FL_TYPED

}  // namespace fl
