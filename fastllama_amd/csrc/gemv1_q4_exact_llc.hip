// gemv1_q4_exact_llc.hip -- the reference-order ("exact") Q4 x Q8_0 matmul for N = 1 (decode), round 4 form: LANE-LOCAL CHAINS.
//
// What must be reproduced (ggml_vec_dot_q4_{0,1}_q8_0, AVX2 branch, /root/reference/lib/ggml.c:2445-2487, :2639-2689): per output
// row 8 f32 accumulators, accumulator j taking  acc_j = fma(d_w * d_x, float(sum of the products of elements 4j..4j+3), acc_j)
// block after block, then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ the scalar chain summs = fma(m_w, s_x, summs) for Q4_1].
//
// Round 3's kernel (gemv1_q4_exact_kernel, exact_kernels.hip) split the work between producer waves (unpack, v_dot4 -> 8 integer lane
// sums per row and block, 32 bytes, through LDS) and chain waves (the fma chains), a barrier per 64-block chunk: 1.5-1.8x the fast
// GEMV's time, all of it the producers' order-free work and its LDS hand-off (profiles/r03_decode_experiments.txt).  A bare stream of
// the bytes takes what the FAST GEMV takes (profiles/r04_gemv_stream_table.md), so the target is a kernel whose arithmetic hides
// under its own stream.  Here a lane OWNS two of the eight chains of a row for the whole of K:
//   * lane = 4 * row + g (16 rows x 4 k-groups = one wave = one 16-row group): k-group g of a block is elements 8g .. 8g+7 =
//     the AVX2 lanes j = 2g (elements 0..3) and 2g + 1 (elements 4..7); the lane keeps acc_2g and acc_2g+1 in two registers and runs
//     their fma chains over the blocks in order -- no hand-off, no LDS traffic for partial results, no barrier after the prologue;
//   * the weights come from a second nibble copy made for this access (QWD, q4_layout.h): the lane's dword of four consecutive
//     blocks in one 16-byte load, nibbles ordered so that  (v << 4) & 0xF0F0F0F0  IS the int8x4 of elements 0..3 (x 16) and
//     v & 0xF0F0F0F0  that of elements 4..7: 3 unpack operations per block instead of 20 (no v_perm);
//   * per lane and block: 3 unpack + 2 v_dot4 + 1 scale product (d_w of the block sits in ONE lane of the row's quad and reaches the
//     other three through the DPP operand of the multiply) + 2 cvt + 2 fma = 10 VALU operations (round 3: ~45 + 68 bytes of LDS);
//   * the stream: U block-quads (1 KiB of nibbles + 256 B of scales per wave and quad) in flight per wave, requested before the
//     prologue and re-requested as they are consumed; nontemporal loads (every byte is read once per token).
// The activation's Q8_0 form is built in LDS once per workgroup by the prologues of round 3 (gemv_prologue.h: rms_norm / silu * mul /
// plain Q8_0, bit-identical to the separate kernels), then re-laid so that a lane reads its 8 bytes of four blocks in two 16-byte
// LDS reads.  Woven w1|w3 (ggml_silu + ggml_mul of lib/llama.cpp:428-431 as the epilogue): PAIR = 1: a workgroup takes the w1 group and
// then the w3 group of the same 16 features and stores silu(w1 x) * (w3 x).  PAIR = 2: the two groups are two ordinary workgroups that
// meet, feature by feature, in a 64-bit slot of a workspace (an atomic exchange: whoever finds the other's value there finishes the
// feature) -- the launch keeps the plain matrix's shape (every byte requested at once, no second chain phase behind the first) and
// the slots are back at zero when the launch ends.
// Q4_0 bookkeeping as everywhere: unpacked values are 16 (nib - 8), the stored scale is d / 16: fma(rn((d/16) d_x), 16 q, a) rounds
// the same real number as the reference's fma(rn(d d_x), q, a).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <cstdlib>
#include <algorithm>
#include "q4_device.h"
#include "q4_kernels.h"
#include <type_traits>
#include "gemv_prologue.h"
#include "tp_tail.h"

#pragma clang fp contract(off)

namespace fl {

typedef unsigned int ntv4u __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- the QWD copy (q4_layout.h) ------------------------------------
// QW16 qs [group][KB][16 rows][4 dwords] (dword position p holds k-group p ^ 2 (row >> 3 & 1); byte t of a k-group dword =
// elements 2t | 2t+1 << 4) -> QWD [group][NQ = ceil(KB / 4)][16 rows][4 k-groups][4 blocks] dwords, byte t = element t | element t+4 << 4
template <int TYPE>
__global__ __launch_bounds__(256) void qw16_to_qwd_kernel(const uint32_t *__restrict__ qs, uint32_t *__restrict__ qwd, int64_t n /* groups * NQ * 256 */,
                                                          int KB, int NQ) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one output dword: (group, quad, row, g, blk)
    if (u >= n) return;
    const int blk = (int)(u & 3), g = (int)((u >> 2) & 3), row = (int)((u >> 4) & 15);
    const int64_t gq = u >> 8;
    const int grp = (int)(gq / NQ), q = (int)(gq % NQ), b = 4 * q + blk;
    uint32_t o = TYPE == FL_TYPE_Q4_0 ? 0u : 0u;                    // a block past K: nibble value 0 after the transform (never read as live)
    if (b < KB) {
        const uint32_t v = qs[(((int64_t)grp * KB + b) * 16 + row) * 4 + (g ^ (((row >> 3) & 1) << 1))];
        // v: nibble 2t = element 2t, nibble 2t+1 = element 2t+1 (t = byte).  out byte t = element t (low) | element t + 4 (high)
        uint32_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (v >> (4 * k)) & 0xF;
        o = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) o |= (e[t] | (e[t + 4] << 4)) << (8 * t);
    }
    qwd[u] = o;
}

// QWD -> QW16 qs: the inverse (a permutation of nibbles: lossless) -- a tensor whose QW16 nibble plane was dropped gets it back from its QWD copy
template <int TYPE>
__global__ __launch_bounds__(256) void qwd_to_qw16_kernel(const uint32_t *__restrict__ qwd, uint32_t *__restrict__ qs, int64_t n /* groups * KB * 64 */, int KB, int NQ) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one output dword: (group, block, row, position p)
    if (u >= n) return;
    const int p = (int)(u & 3), row = (int)((u >> 2) & 15);
    const int64_t gb = u >> 6;
    const int grp = (int)(gb / KB), b = (int)(gb % KB);
    const int g = p ^ (((row >> 3) & 1) << 1);                      // the k-group stored at position p of this row
    const uint32_t v = qwd[((((int64_t)grp * NQ + (b >> 2)) * 16 + row) * 4 + g) * 4 + (b & 3)];
    uint32_t o = 0;                                                  // QWD byte t = element t | element t + 4 << 4; QW16 nibble k = element k
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t by = (v >> (8 * t)) & 0xFF;
        o |= (by & 0xF) << (4 * t);
        o |= (by >> 4) << (4 * (t + 4));
    }
    qs[u] = o;
}
hipError_t qwd_to_qw16(const fl_qtensor &W, uint32_t *qs, hipStream_t st) {
    if (!W.qwd) return hipErrorInvalidValue;
    const int NQ = (W.KB + 3) / 4;
    const int64_t n = (int64_t)(W.M16 / 16) * W.KB * 64;
    if (n == 0) return hipSuccess;
    const int64_t nb = (n + 255) / 256;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    if (W.type == FL_TYPE_Q4_0) hipLaunchKernelGGL(qwd_to_qw16_kernel<FL_TYPE_Q4_0>, dim3((unsigned)nb), dim3(256), 0, st, W.qwd, qs, n, W.KB, NQ);
    else hipLaunchKernelGGL(qwd_to_qw16_kernel<FL_TYPE_Q4_1>, dim3((unsigned)nb), dim3(256), 0, st, W.qwd, qs, n, W.KB, NQ);
    return hipGetLastError();
}

size_t qwd_bytes(const fl_qtensor &W) { return (size_t)(W.M16 / 16) * (size_t)((W.KB + 3) / 4) * 1024; }

hipError_t qw16_to_qwd(const fl_qtensor &W, uint32_t *qwd, hipStream_t st) {
    const int NQ = (W.KB + 3) / 4;
    const int64_t n = (int64_t)(W.M16 / 16) * NQ * 256;
    if (n == 0) return hipSuccess;
    const int64_t nb = (n + 255) / 256;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    if (W.type == FL_TYPE_Q4_0) hipLaunchKernelGGL(qw16_to_qwd_kernel<FL_TYPE_Q4_0>, dim3((unsigned)nb), dim3(256), 0, st, W.qs, qwd, n, W.KB, NQ);
    else hipLaunchKernelGGL(qw16_to_qwd_kernel<FL_TYPE_Q4_1>, dim3((unsigned)nb), dim3(256), 0, st, W.qs, qwd, n, W.KB, NQ);
    return hipGetLastError();
}

// ---------------------------------------------------------------- the kernel -----------------------------------------------------
// value of lane (quad base + SRC) of every quad: the DPP quad_perm broadcast
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
    return dpp_f32<SRC | (SRC << 2) | (SRC << 4) | (SRC << 6)>(v);
}

#ifdef LLC_TIMING   // development build only: per-workgroup clocks of EVERY launch since the last reset, in one ring (scripts/dev/decode_timeline.py):
                    // the 100 MHz wall clock is common to all launches, so the records of a token's kernels line up on one time axis
constexpr unsigned LLC_TL_CAP = 1u << 17;
constexpr int LLC_TL_W = 12;      // int64 per record: 7 stamps, id, 3 prologue stamps, spare
__device__ long long llc_tl[(size_t)LLC_TL_CAP * LLC_TL_W];
__device__ unsigned llc_tl_cur;
#define LLC_T_DECL long long tl_[7] = {0, 0, 0, 0, 0, 0, 0}, tlp_[4] = {0, 0, 0, 0}
#define LLC_STAMP(k) do { tl_[k] = wall_clock64(); } while (0)
#define LLC_COMMIT(kid)                                                                        \
    do {                                                                                       \
        if (threadIdx.x == 0) {                                                                \
            const unsigned s_ = atomicAdd(&llc_tl_cur, 1u);                                    \
            if (s_ < LLC_TL_CAP) {                                                             \
                for (int k_ = 0; k_ < 7; ++k_) llc_tl[(size_t)s_ * LLC_TL_W + k_] = tl_[k_];   \
                llc_tl[(size_t)s_ * LLC_TL_W + 7] = ((long long)(kid) << 32) | blockIdx.x;     \
                for (int k_ = 0; k_ < 4; ++k_) llc_tl[(size_t)s_ * LLC_TL_W + 8 + k_] = tlp_[k_]; \
            }                                                                                  \
        }                                                                                      \
    } while (0)
#else
#define LLC_T_DECL do {} while (0)
#define LLC_STAMP(k) do {} while (0)
#define LLC_COMMIT(kid) do {} while (0)
#endif
// NK waves share a 16-row group along K (wave k: quads [k NQ / NK, (k+1) NQ / NK)); PAIR = 1: the workgroup takes the w1 group and then
// the w3 group of the same 16 features (one after the other: three workgroups per CU cover each other's round trips).  QPW: most quads a wave can hold (its lane sums stay in registers until its turn in the chain).
// PERSIST = 1: the row-group loop (a launch with fewer workgroups than row groups).  Its own instantiation: the loop costs ~45 registers
// (200: two workgroups per CU instead of three), which a launch whose row groups all fit on the chip at once need not pay.
// (Round 5 also built workgroups of several row groups sharing one prologue -- "teams": 551 against 594 tok/s, twelve waves behind one barrier stretch
//  the chain phase -- and persistent workgroups for K <= 4096 -- 581 against 593: both removed in round 6, profiles/r05_decode_exact.md has the numbers.
//  Matrices of many row groups now take gemv1_q4_exact_stream.hip, whose four row groups per workgroup share a prologue without sharing a barrier.)
template <int TYPE, int NK, int PRO, int PAIR, int QPW, int PERSIST, int TAIL = 0>
__global__ __launch_bounds__(64 * NK, (!PERSIST && QPW <= 8 && NK == 4) ? 3 : 1) void gemv1_q4_exact_llc_kernel(
    int M, int units, int KB, int woven,
    const uint32_t *__restrict__ qwd, const float *__restrict__ dW, const float *__restrict__ xf, const void *__restrict__ aux,
    const float *__restrict__ mW, const int8_t *__restrict__ xq, const float *__restrict__ xd, const float *__restrict__ xs,
    float *__restrict__ y, const float *__restrict__ resid, float *__restrict__ ynorm, const uint16_t *__restrict__ aux2,
    float *pair_ws /* PAIR = 2: [units / 2][16] 64-bit slots (zero between launches) */,
    const TpTail *__restrict__ tt /* tensor parallel: the exchange of this launch's rows as its tail (tp_tail.h); NULL: none */,
    int npass /* PERSIST: K passes of NK x QPW quads per row group (rows longer than one pass holds in registers); else 1 */) {
    constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    constexpr int G2 = PAIR == 1 ? 2 : 1, NT = 64 * NK;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __shared__ double sh[4];
    __shared__ float accs[2][64][3];                                            // the chains' state between the K slices: a_2g, a_2g+1, summs
                                                                                // (two copies: consecutive row groups of a persistent workgroup alternate)
    __shared__ int chain_seq[2];                                                // chain turns taken, per copy (only grows): the hand-off from slice to slice
    const int lane = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = wave_;                                                        // (the slice index in an SGPR: the `i < nq` tests below are scalar branches)
    const int NQ = (KB + 3) >> 2;
    // LDS: [LX: the Q8_0 activation as the lanes read it, [NQ][4 k-groups][4 blocks][8 B: e0..e3 | e4..e7]] [d [4 NQ]] [s [4 NQ]]
    // (round 5: the prologues write LX directly -- round 4 built the QA1 layout first and re-laid it in a second pass behind a barrier)
    unsigned char *lx = gsm;
    float *ld_ = reinterpret_cast<float *>(gsm + (size_t)NQ * 128);             // (d and s padded to whole quads: 16-byte reads; the padding is
    float *ls_ = ld_ + 4 * NQ;                                                  //  zero, so a block past K has dd = 0 and adds nothing)

    if (threadIdx.x < 2) chain_seq[threadIdx.x] = 0;                            // (ahead of the prologue's closing barrier)
    LLC_T_DECL;
    LLC_STAMP(0);
    GP_DECL(PRO);
    GemvPrologue<PRO, NT, true>::issue(pv, pw, psl, psb, xf, aux, KB, woven);

    // ---- this wave's slice of the weight stream: every load goes out before anything waits (nontemporal: read once per token)
    // PERSIST = 1 (round 5, opt-in): a launch has at most one workgroup per residency slot; a workgroup takes the row groups blockIdx.x,
    // blockIdx.x + gridDim.x, ... -- the prologue (the activation's Q8_0 form in LDS) is paid once, and the next row group's weights are requested
    // behind this one's chains (do_group, below).  (LLaMA-7B's woven w1|w3 is 1376 row groups on 768 slots: as one workgroup per row group the second
    // round's workgroups are dispatched 7-10 us into the launch, each redoing the prologue: profiles/r05_decode_timeline.md -- and still the faster form.)
    int unit = (int)blockIdx.x;
    constexpr bool live = true;
    // MULTI-PASS rows (PERSIST instantiation, round 5): a row longer than NK x QPW quads is taken in npass passes of that many quads, pass after pass
    // through the same registers -- the chain state goes from the last wave of a pass to the first wave of the next through LDS, so the summation
    // order is the row's block order as ever, and a K = 8192 .. 22016 row group runs in the 4 x 8 form (three workgroups per CU) instead of
    // the 8 x 11 form / round 3's kernel (one per CU).  slice_of: wave k's quads of pass p (wave-uniform; nq <= QPW by the launcher's choice).
    constexpr int PQ = NK * QPW;
    auto slice_of = [&](int p_, int &qlo_, int &nq_) __attribute__((always_inline)) {
        if (PERSIST) {
            const int lo = (p_ * NQ) / npass, nqp = ((p_ + 1) * NQ) / npass - lo;          // (npass equal parts of the row: <= PQ quads each)
            qlo_ = lo + (k * nqp) / NK;
            nq_ = lo + ((k + 1) * nqp) / NK - qlo_;
        } else {
            qlo_ = (k * NQ) / NK;
            nq_ = ((k + 1) * NQ) / NK - qlo_;
        }
    };
    int qlo, nq;
    slice_of(0, qlo, nq);
    const int r = lane >> 2, g = lane & 3;
    ntv4u w[QPW];
    float dw[QPW], mw[QPW];
    // Addresses: buffer loads with the wave-uniform part (row group, quad) in the SCALAR offset and one lane offset register per plane
    // (round 5).  The flat 64-bit addresses of round 4 cost two VGPRs per load -- kept alive across the row-group loop they were 45
    // registers, the difference between three workgroups per CU and two.  Bytes past a plane read as zero (descriptor bounds).
    const int groups = units * G2;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(qwd), 0, (int)((uint32_t)groups * (uint32_t)NQ * 1024u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dW), 0, (int)((uint32_t)groups * (uint32_t)KB * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q41 ? mW : dW), 0, (int)((uint32_t)groups * (uint32_t)KB * 64u), 0x00020000);
    const int voff_w = lane * 16;
    const int voff_d = (g * 16 + r) * 4;                                         // lane (row, g) fetches the scale of block 4q + g of its row
    // (a partial last quad reads past the row's blocks: the next row group's first scales, or zeros behind the plane -- FINITE numbers
    //  (fl_qtensor_upload rejects anything else) that meet d_x = 0 and zero nibbles: the block adds +0 to a chain that can never hold -0)
    // The scales go out FIRST, then the nibbles, quad by quad: loads return in order, so a quad can be summed as soon as ITS sixteen bytes per
    // lane are there.  (Issued quad by quad -- nibbles, scale, nibbles, scale -- the compiler's scheduler moved all the scale loads behind
    // all the nibble loads, and the first quad's sums waited for the wave's whole slice: round 5, seen in the ISA.)
    auto load_group = [&](int grp, int qlo, int nq) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int q = qlo + (i < nq ? i : 0);                               // (past the slice: a cache-hot dummy, never used -- unconditional
            const int sd = (grp * KB + 4 * q) * 64;                             //  loads let the compiler count the ones in flight)
            dw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, voff_d, sd, 2));
            if (Q41) mw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff_d, sd, 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int q = qlo + (i < nq ? i : 0);
            const int gq = grp * NQ + q;
            w[i] = __builtin_bit_cast(ntv4u, __builtin_amdgcn_raw_buffer_load_b128(rW, voff_w, gq * 1024, 2 /* nt */));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // PRO = 0: the Q8_0 activation (QA1 in HBM) is requested BEFORE the weight stream as well (round 5): behind it -- loads return in order --
    // its copy to LDS waited for the wave's whole slice, and every lane sum of the launch was computed after the last byte had arrived
    constexpr int XIT = 4;                                                      // 8-byte k-group entries per thread held in registers
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
    load_group(unit * G2, qlo, nq);
    LLC_STAMP(1);

    // blocks past K in a partial last quad: zero quants, d_x = s_x = 0 (written before the prologue's closing barrier)
    if ((int)threadIdx.x < (4 * NQ - KB) * 4) {
        const int b = KB + ((int)threadIdx.x >> 2), gg = threadIdx.x & 3;
        *reinterpret_cast<uint2 *>(lx + (((b >> 2) * 4 + gg) * 4 + (b & 3)) * 8) = make_uint2(0, 0);
        if (gg == 0) { ld_[b] = 0.f; ls_[b] = 0.f; }
    }
    if constexpr (PRO != 0) {
#ifdef LLC_TIMING
        GemvPrologue<PRO, NT, true>::finish(pv, pw, psl, psb, xf, aux, KB, woven, reinterpret_cast<int8_t *>(lx), ld_, ls_, sh, ynorm, blockIdx.x == 0, tlp_);
#else
        GemvPrologue<PRO, NT, true>::finish(pv, pw, psl, psb, xf, aux, KB, woven, reinterpret_cast<int8_t *>(lx), ld_, ls_, sh, ynorm, blockIdx.x == 0);
#endif
    } else {                                          // the activation is already Q8_0 (QA1 in HBM: k-group bytes e0,e2,e4,e6 | e1,e3,e5,e7): re-lay it on the way
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
        for (int i = threadIdx.x + XIT * NT; i < KB * 4; i += NT) put(i, reinterpret_cast<const uint2 *>(xq)[i]);      // (very long rows: the rest)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x + it * NT;
            if (i < KB) { ld_[i] = xd_[it]; ls_[i] = xs_[it]; }
        }
        for (int i = threadIdx.x + 2 * NT; i < KB; i += NT) { ld_[i] = xd[i]; ls_[i] = Q41 ? xs[i] : 0.f; }
        __syncthreads();
    }
    LLC_STAMP(2);

    const uint32_t m8 = 0xF0F0F0F0u;
    // (with a tail: written through to memory, and to every peer's region -- tp_put, tp_tail.h)
    auto put_y = [&](float *p, float v) __attribute__((always_inline)) {
        if constexpr (TAIL) tp_put(tt, p, v);       // (its own instantiation: the launches of an unsharded model are the code they were before the tail)
        else *p = v;
    };
    float y1 = 0.f;
    int par = 0;                                                                // which copy of the chain state this row group uses
    int pass = 0;                                                               // PERSIST: which K pass of the row group
    int turn0[2] = {0, 0};                                                      // chain_seq[copy] when the current row (group) of that copy started
    auto do_group = [&](auto GI) __attribute__((always_inline)) {
        constexpr int gi = decltype(GI)::value;
        // the residual of this lane's row is requested NOW (PAIR = 0): asked for in the epilogue it was a dependent round trip of ~0.6 us at
        // the very end of every wo / w2 launch, behind the last chain (round 5)
        float rsd = 0.f;
        if constexpr (PAIR == 0) {
            const int row_ = unit * 16 + r;
            if (resid && k == NK - 1 && (!PERSIST || pass == npass - 1)) rsd = resid[min(row_, M - 1)];      // (wave-uniform condition, clamped address: a load behind a lane-dependent branch makes the compiler drain every load in flight)
        }
        // ---- order-free part, all waves at once: per block the two lane sums of this lane's k-group as floats, rn(d_w d_x), m_w
        float f0[QPW][4], f1[QPW][4], dd[QPW][4];
        // Q4_1, eight-wave forms with 8+ quads per wave (one workgroup per CU whatever the registers, up to 256): m_w broadcast and s_x read HERE.
        // Left in the chain -- 11 LDS reads behind the turn's acquire, 44 DPP moves -- they made every slice's turn of LLaMA-7B's w2 longer: the
        // launch took 15.0 us where the bytes (1.2 x Q4_0's) ask for 11 (round 6).  The 4-quad forms with a prologue keep them in the chain: 104-106
        // registers as they are, and 128 is what gives them two workgroups per CU.
        constexpr bool PRE_M = Q41 && (QPW >= 8 || PRO == 0);                     // (PRO = 0, 4 quads: 68 -> ~100 registers, still two workgroups per CU)
        float msb_[PRE_M ? QPW : 1][4], sxv_[PRE_M ? QPW : 1][4];
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            if (i < nq) {                                                       // (wave-uniform)
                const int q = qlo + i;
                if constexpr (PRE_M) {
                    msb_[i][0] = quad_bcast<0>(mw[i]); msb_[i][1] = quad_bcast<1>(mw[i]); msb_[i][2] = quad_bcast<2>(mw[i]); msb_[i][3] = quad_bcast<3>(mw[i]);
                    const float4 sx4 = *reinterpret_cast<const float4 *>(ls_ + 4 * q);
                    sxv_[i][0] = sx4.x; sxv_[i][1] = sx4.y; sxv_[i][2] = sx4.z; sxv_[i][3] = sx4.w;
                }
                const uint4 x01 = *reinterpret_cast<const uint4 *>(lx + ((size_t)q * 4 + g) * 32);
                const uint4 x23 = *reinterpret_cast<const uint4 *>(lx + ((size_t)q * 4 + g) * 32 + 16);
                const float4 dx4 = *reinterpret_cast<const float4 *>(ld_ + 4 * q);
                const uint32_t wv[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
                const uint32_t xa[4] = {x01.x, x01.z, x23.x, x23.z}, xb[4] = {x01.y, x01.w, x23.y, x23.w};
                const float dxv[4] = {dx4.x, dx4.y, dx4.z, dx4.w};
                const float dwb[4] = {quad_bcast<0>(dw[i]), quad_bcast<1>(dw[i]), quad_bcast<2>(dw[i]), quad_bcast<3>(dw[i])};
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    uint32_t wa, wb;
                    if (TYPE == FL_TYPE_Q4_0) { wa = (wv[blk] << 4) & m8; wb = wv[blk] & m8; }     // 16 (nib - 8): elements 0..3 | 4..7
                    else { wa = wv[blk] & 0x0F0F0F0Fu; wb = (wv[blk] >> 4) & 0x0F0F0F0Fu; }
                    f0[i][blk] = (float)__builtin_amdgcn_sdot4((int)wa, (int)xa[blk], 0, false);
                    f1[i][blk] = (float)__builtin_amdgcn_sdot4((int)wb, (int)xb[blk], 0, false);
                    dd[i][blk] = __fmul_rn(dwb[blk], dxv[blk]);                 // rn(d_w d_x); a block past K: d_x = 0
                }
#ifdef LLC_TIMING
                if (gi == 0 && i == 0) { asm volatile("" :: "v"(f0[0][0]), "v"(f1[0][3])); LLC_STAMP(6); }      // the first quad has arrived and is summed
#endif
            }
        }
        if (gi == 0) LLC_STAMP(3);
        // ---- the chains, slice after slice: wave k continues from the state wave k - 1 left in LDS.  The hand-off is wave to wave (round 6): a
        // wave waits for ITS predecessor's turn number in chain_seq, not at a workgroup barrier -- behind barriers the second slice's chain could
        // not start before the LAST slice's bytes had arrived and been summed, and the NK chain phases of a row stood one after the other behind the
        // whole stream (2.7 us of LLaMA-7B's 10 us w2 launch, profiles/r05_decode_timeline.md); now a slice's chain runs as soon as its own lane
        // sums and its predecessor's state are there.  LDS serves a wave's accesses in order: state first, turn number behind it.
        float a0 = 0.f, a1 = 0.f, summs = 0.f;
        {
            const int turn = turn0[par] + pass * NK + k;                        // this slice's turn in the row's chain (multi-pass rows: pass after pass)
            {
                if (pass * NK + k > 0) {                                        // (wave-uniform; a row's first slice starts from zero)
                    while (__hip_atomic_load(&chain_seq[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < turn) __builtin_amdgcn_s_sleep(1);
                    a0 = accs[par][lane][0]; a1 = accs[par][lane][1]; if (Q41) summs = accs[par][lane][2];
                }
#pragma unroll
                for (int i = 0; i < QPW; ++i) {
                    if (i < nq) {
                        float sxv[4] = {0.f, 0.f, 0.f, 0.f}, msb[4] = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (PRE_M) {
#pragma unroll
                            for (int blk = 0; blk < 4; ++blk) { msb[blk] = msb_[i][blk]; sxv[blk] = sxv_[i][blk]; }
                        } else if (Q41) {
                            // (m_w of a block sits in one lane of the row's quad, as d_w does; broadcast here, in the chain -- kept as four registers
                            //  per quad since the order-free part it cost Q4_1 its third workgroup per CU: round 5)
                            msb[0] = quad_bcast<0>(mw[i]); msb[1] = quad_bcast<1>(mw[i]); msb[2] = quad_bcast<2>(mw[i]); msb[3] = quad_bcast<3>(mw[i]);
                            const float4 sx4 = *reinterpret_cast<const float4 *>(ls_ + 4 * (qlo + i));
                            sxv[0] = sx4.x; sxv[1] = sx4.y; sxv[2] = sx4.z; sxv[3] = sx4.w;
                        }
#pragma unroll
                        for (int blk = 0; blk < 4; ++blk) {
                            a0 = __fmaf_rn(dd[i][blk], f0[i][blk], a0);
                            a1 = __fmaf_rn(dd[i][blk], f1[i][blk], a1);
                            if (Q41) summs = __fmaf_rn(msb[blk], sxv[blk], summs);      // (a block past K: s_x = 0, m_w finite)
                        }
                    }
                }
                if (k < NK - 1 || (PERSIST && pass < npass - 1)) { accs[par][lane][0] = a0; accs[par][lane][1] = a1; if (Q41) accs[par][lane][2] = summs; }
                __hip_atomic_store(&chain_seq[par], turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (!PERSIST || pass == npass - 1) turn0[par] += npass * NK;        // (the copy's next row starts behind this one's last turn)
        }
        if (gi == 0) LLC_STAMP(4);
        // ---- the row group is complete in its last wave: ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) over the quad of lanes that holds the
        // row (lane g holds accumulators 2g, 2g+1; every lane of the quad ends with the same bits), then the store / the PAIR epilogue
        if (k == NK - 1 && (!PERSIST || pass == npass - 1)) {
            float e = a0, o = a1;
            e = __fadd_rn(e, dpp_f32<DPP_XOR2>(e));      // a0+a4 | a2+a6
            o = __fadd_rn(o, dpp_f32<DPP_XOR2>(o));      // a1+a5 | a3+a7
            e = __fadd_rn(e, dpp_f32<DPP_XOR1>(e));      // (a0+a4)+(a2+a6)
            o = __fadd_rn(o, dpp_f32<DPP_XOR1>(o));      // (a1+a5)+(a3+a7)
            float v = __fadd_rn(e, o);
            if (Q41) v = __fadd_rn(v, summs);
            const int row = (unit * G2 + gi) * 16 + r;
            if constexpr (PAIR == 2) {
                // groups 2u (w1) and 2u + 1 (w3) of the woven matrix are this pair.  One 64-bit slot per feature: each side EXCHANGES
                // (1 << 32 | its dot product) into it; the side that gets the other's word back arrived second and finishes the
                // feature, then clears the slot -- one agent-scope atomic round trip per workgroup, no fence (a release / acquire
                // fence writes back / invalidates the whole L2 of the XCD: 58 us for this launch), no order between the sides.
                if (g == 0 && row < M && live) {
                    unsigned long long *slot = reinterpret_cast<unsigned long long *>(pair_ws) + (size_t)(unit >> 1) * 16 + r;
                    const unsigned long long mine = (1ull << 32) | (unsigned long long)__float_as_uint(v);
                    const unsigned long long old = __hip_atomic_exchange(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old >> 32) {
                        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float other = __uint_as_float((unsigned)old);
                        const float h1 = (unit & 1) ? other : v, h3 = (unit & 1) ? v : other;
                        const uint16_t hx = __half_as_ushort(__float2half_rn(h1));                    // GGML_FP32_TO_FP16
                        const float sl = __half2float(__ushort_as_half(aux2[hx]));                   // table_silu_f16
                        put_y(y + (unit >> 1) * 16 + r, __fmul_rn(sl, h3));                           // ggml_mul(silu, tmp)
                    }
                }
            } else if constexpr (PAIR == 1) {
                if (gi == 0) {
                    y1 = v;                                                    // w1 . x of this feature; its w3 row comes next
                } else if (g == 0 && row < M) {
                    const uint16_t hx = __half_as_ushort(__float2half_rn(y1));                // GGML_FP32_TO_FP16
                    const float sl = __half2float(__ushort_as_half(aux2[hx]));               // table_silu_f16
                    put_y(y + unit * 16 + r, __fmul_rn(sl, v));                               // ggml_mul(silu, tmp)
                }
            } else if (g == 0 && row < M && live) {
                if (resid) v = __fadd_rn(v, rsd);
                put_y(y + row, v);
            }
        }
        if constexpr (PAIR != 1 && PERSIST) {
            // the next row group's bytes start now -- behind this wave's chain and, for the last wave, behind the store (a load issued before the
            // residual's round trip would make the store wait for the whole prefetch: the compiler counts loads, it does not tell them apart).
            // Requested right after the lane sums (when w / dw / mw die) they would keep 40 registers busy next to the 96 of the lane sums.
            // (multi-pass rows: the row group's next pass; the last wave's state reaches the first wave of that pass through chain_seq)
            const bool more = pass + 1 < npass;
            const int next_unit = more ? unit : unit + (int)gridDim.x;
            int nqlo, nnq;
            slice_of(more ? pass + 1 : 0, nqlo, nnq);
            if (next_unit < units) load_group(next_unit, nqlo, nnq);
        }
        if (PAIR == 1 && gi == 0) {
            // the w3 group's slice is requested only now.  Requested before the w1 chains it keeps both groups' registers alive: 214
            // VGPRs = two workgroups per CU = 1.3 rounds of the 688 workgroups, 24 us; squeezed under the 170-register cap of three
            // workgroups per CU it spills (22 us, also with the lane sums packed as int16 pairs); this order: 19.5 us
            // (profiles/r04_decode_exact.md).  What it costs: HBM idles while the whole launch sits in its w1 chain phase.
            load_group(unit * G2 + 1, qlo, nq);
            __syncthreads();                            // (the chain state in LDS is free again for the second group)
        }
    };
    if constexpr (PAIR == 1) {
        do_group(std::integral_constant<int, 0>{});
        do_group(std::integral_constant<int, 1>{});
    } else if constexpr (!PERSIST) {
        do_group(std::integral_constant<int, 0>{});
    } else {
#pragma unroll 1
        for (;;) {
            // (compiler barrier: the activation's LDS reads do not depend on the row group, and hoisted out of this loop they cost 96 registers)
            asm volatile("" ::: "memory");
            do_group(std::integral_constant<int, 0>{});
            if (++pass < npass) { slice_of(pass, qlo, nq); continue; }
            pass = 0;
            slice_of(0, qlo, nq);
            unit += (int)gridDim.x;
            if (unit >= units) break;
            par ^= 1;
        }
    }
    LLC_STAMP(5);
    LLC_COMMIT(PRO * 100 + PAIR * 10 + NK);
    if constexpr (TAIL) tp_tail<false, false, true>(tt);
}
#ifdef LLC_TIMING
// reset != 0: empty the ring; else copy up to max_rec records of 8 x int64 {t0 entry, t1 loads issued, t2 prologue done, t3 lane sums of the
// wave's slice done (= its last byte has arrived), t4 chains handed through, t5 end, t6 first quad summed, (kernel id << 32) | workgroup}
// and return how many there are
extern "C" __attribute__((visibility("default"))) int fl_debug_llc_timeline(long long *out, int max_rec, int reset) {
    unsigned n = 0;
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(llc_tl_cur), &n, sizeof n);
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(llc_tl_cur), sizeof n) != hipSuccess) return -1;
    if (n > LLC_TL_CAP) n = LLC_TL_CAP;
    if ((int)n > max_rec) n = (unsigned)max_rec;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(llc_tl), sizeof(long long) * LLC_TL_W * (size_t)n) != hipSuccess) return -1;
    return (int)n;
}
#endif

// residency slots of a kernel instantiation on this device: workgroups per CU (occupancy query with the launch's dynamic LDS) x CUs, rounded
// down to an even number
static int llc_slots(const void *fn, int threads, size_t lds) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
    return (per_cu * cus) & ~1;
}

// false: no QWD copy, or a shape outside the kernel's reach (activation beyond LDS) -> the caller takes round 3's kernel
template <int TYPE, int PRO, int PAIR>
static bool launch_llc(const fl_qtensor &W, const fl_qact *xq, float *y, hipStream_t st, const float *resid, const float *xf, const void *aux,
                       float *ynorm, int woven, const uint16_t *aux2, float *pair_ws = nullptr) {
    if (!W.qwd) return false;
    constexpr int G2 = PAIR == 1 ? 2 : 1;
    const int KB = W.KB, NQ = (KB + 3) / 4, units = W.M16 / 16 / G2;
    const size_t lds = (size_t)NQ * 128 + (size_t)NQ * 32;                   // LX + d_x + s_x
    // Which form (profiles/r05_decode_exact.md): a wave keeps its whole K slice in registers -- 4 waves x 8 quads covers K <= 4096 with the fewest
    // registers (three workgroups per CU), 4 x 11 K <= 5632, 8 x 11 K <= 11264 -- but an 8-wave workgroup at 186 registers is ONE per CU: good for a
    // matrix of <= 256 row groups (LLaMA-7B's w2), bad for many groups of long rows.  Those, and rows beyond 11264 (13B / 65B w2), are MULTI-PASS rows:
    // the row in npass equal parts of at most 4 x 6 quads (Q4_1: 8 x 4), pass after pass through the same registers (142-146 VGPRs, three workgroups
    // per CU), one workgroup per residency slot walking the row groups.
    const bool mp = PAIR != 1 && NQ > 32 && (NQ > 88 || (NQ > 44 && units > 256));
    if (lds > 60 * 1024 || units < 1 || (NQ > 88 && !mp)) return false;
    if (qwd_bytes(W) >= (1ull << 31) || (size_t)W.M16 * (size_t)KB * 4 >= (1ull << 31)) return false;      // 32-bit buffer offsets (no LLaMA tensor comes close)
    if (NQ > 44 && units > 256 && !mp) return false;
    const TpTail *tt = tp_take_tail();
#define FL_LLC_GO(NK, QPW, PS, GRID, NPASS)                                                                                               \
    do {                                                                                                                                  \
        if (tt)                                                                                                                           \
            hipLaunchKernelGGL((gemv1_q4_exact_llc_kernel<TYPE, NK, PRO, PAIR, QPW, PS, 1>), dim3(GRID), dim3(64 * NK), lds, st, W.M, units, KB, \
                               woven, W.qwd, W.d, xf, aux, W.m, xq ? xq->q : nullptr, xq ? xq->d : nullptr, xq ? xq->s : nullptr, y, resid, \
                               ynorm, aux2, pair_ws, tt, NPASS);                                                                          \
        else                                                                                                                              \
            hipLaunchKernelGGL((gemv1_q4_exact_llc_kernel<TYPE, NK, PRO, PAIR, QPW, PS, 0>), dim3(GRID), dim3(64 * NK), lds, st, W.M, units, KB, \
                               woven, W.qwd, W.d, xf, aux, W.m, xq ? xq->q : nullptr, xq ? xq->d : nullptr, xq ? xq->s : nullptr, y, resid, \
                               ynorm, aux2, pair_ws, tt, NPASS);                                                                          \
    } while (0)
    if constexpr (PAIR != 1) {
        if (mp) {
#define FL_LLC_MP_GO(MNK, MQPW)                                                                                                            \
            do {                                                                                                                           \
                static int slots = 0;                                                                                                      \
                static size_t slots_lds = 0;                                                                                               \
                if (!slots || slots_lds != lds)                                                                                            \
                    slots_lds = lds, slots = llc_slots(reinterpret_cast<const void *>(&gemv1_q4_exact_llc_kernel<TYPE, MNK, PRO, PAIR, MQPW, 1>), 64 * MNK, lds); \
                FL_LLC_GO(MNK, MQPW, 1, (units < slots ? units : slots), (NQ + MNK * MQPW - 1) / (MNK * MQPW));                             \
            } while (0)
            if constexpr (TYPE == FL_TYPE_Q4_1) FL_LLC_MP_GO(8, 4);
            else FL_LLC_MP_GO(4, 6);
#undef FL_LLC_MP_GO
            return true;
        }
    }
    // one workgroup per row group
    if constexpr (TYPE == FL_TYPE_Q4_1) {             // Q4_1 carries m_w as well: 4 x 8 needs 174 registers = two waves per SIMD; 8 x 4 needs <= 126
        if (NQ <= 32) { FL_LLC_GO(8, 4, 0, units, 1); return true; }
    }
    if (NQ <= 32) FL_LLC_GO(4, 8, 0, units, 1);
    else if (NQ <= 44) FL_LLC_GO(4, 11, 0, units, 1);
    else FL_LLC_GO(8, 11, 0, units, 1);
#undef FL_LLC_GO
    return true;
}

#define FL_TYPED(CALL0, CALL1) (W.type == FL_TYPE_Q4_0 ? (CALL0) : (CALL1))
bool gemv1_llc(const fl_qtensor &W, const fl_qact &xq, float *y, hipStream_t st, const float *resid) {
    return FL_TYPED((launch_llc<FL_TYPE_Q4_0, 0, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr, 0, nullptr)),
                    (launch_llc<FL_TYPE_Q4_1, 0, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr, 0, nullptr)));
}
bool gemv1_llc_norm(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st) {
    return FL_TYPED((launch_llc<FL_TYPE_Q4_0, 1, 0>(W, nullptr, y, st, nullptr, x, norm_w, ynorm, 0, nullptr)),
                    (launch_llc<FL_TYPE_Q4_1, 1, 0>(W, nullptr, y, st, nullptr, x, norm_w, ynorm, 0, nullptr)));
}
bool gemv1_llc_silu(const fl_qtensor &W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid, hipStream_t st, bool woven) {
    return FL_TYPED((launch_llc<FL_TYPE_Q4_0, 2, 0>(W, nullptr, y, st, resid, h13, silu_tab, nullptr, woven ? 1 : 0, nullptr)),
                    (launch_llc<FL_TYPE_Q4_1, 2, 0>(W, nullptr, y, st, resid, h13, silu_tab, nullptr, woven ? 1 : 0, nullptr)));
}
size_t gemv1_llc_pair_ws_bytes(int M) { return (size_t)(M + 31) / 32 * 16 * 8; }
bool gemv1_llc_norm_silu(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act, hipStream_t st,
                         float *pair_ws, int form /* 0: automatic; 1 / 2 pins a form (tests, A/B: profiles/r04_decode_exact.md) */) {
    if (pair_ws && (W.M16 / 16) % 2 == 0 && form != 1)   // (pair_ws: gemv1_llc_pair_ws_bytes(W.M) bytes, zero before its first use)
        return FL_TYPED((launch_llc<FL_TYPE_Q4_0, 1, 2>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab, pair_ws)),
                        (launch_llc<FL_TYPE_Q4_1, 1, 2>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab, pair_ws)));
    return FL_TYPED((launch_llc<FL_TYPE_Q4_0, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab)),
                    (launch_llc<FL_TYPE_Q4_1, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab)));
}
bool gemv1_llc_quant(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st) {
    return FL_TYPED((launch_llc<FL_TYPE_Q4_0, 3, 0>(W, nullptr, y, st, resid, x, nullptr, nullptr, 0, nullptr)),
                    (launch_llc<FL_TYPE_Q4_1, 3, 0>(W, nullptr, y, st, resid, x, nullptr, nullptr, 0, nullptr)));
}
#undef FL_TYPED

}  // namespace fl
