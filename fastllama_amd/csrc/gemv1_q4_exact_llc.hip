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
#include "q4_device.h"
#include "q4_kernels.h"
#include <type_traits>
#include "gemv_prologue.h"

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

#ifdef LLC_TIMING   // development build only: per-workgroup clocks of a launch (scripts/dev/llc_timeline.py)
__device__ long long llc_dbg[2048 * 8];
#define LLC_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) llc_dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define LLC_STAMP(k) do {} while (0)
#endif
// NK waves share a 16-row group along K (wave k: quads [k NQ / NK, (k+1) NQ / NK)); PAIR = 1: the workgroup takes the w1 group and then
// the w3 group of the same 16 features (one after the other: three workgroups per CU cover each other's round trips).  QPW: most quads a wave can hold (its lane sums stay in registers until its turn in the chain).
template <int TYPE, int NK, int PRO, int PAIR, int QPW>
__global__ __launch_bounds__(64 * NK, (PAIR == 1 && QPW <= 8 && NK == 4) ? 3 : 1) void gemv1_q4_exact_llc_kernel(
    int M, int units, int KB, int woven,
    const uint32_t *__restrict__ qwd, const float *__restrict__ dW, const float *__restrict__ xf, const void *__restrict__ aux,
    const float *__restrict__ mW, const int8_t *__restrict__ xq, const float *__restrict__ xd, const float *__restrict__ xs,
    float *__restrict__ y, const float *__restrict__ resid, float *__restrict__ ynorm, const uint16_t *__restrict__ aux2,
    float *pair_ws /* PAIR = 2: [units / 2][16] 64-bit slots (zero between launches) */) {
    constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    constexpr int G2 = PAIR == 1 ? 2 : 1, NT = 64 * NK;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __shared__ double sh[4];
    __shared__ float accs[64][3];                                               // the chains' state between the K slices: a_2g, a_2g+1, summs
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int NQ = (KB + 3) >> 2;
    // LDS: [Q8_0 activation, QA1 layout: q [KB][32], d [4 NQ], s [4 NQ]] [LX: the lanes' view, [NQ][4 k-groups][4 blocks][8 B]]
    int8_t *lq = reinterpret_cast<int8_t *>(gsm);
    float *ld_ = reinterpret_cast<float *>(gsm + (size_t)KB * 32);              // (d and s padded to whole quads: 16-byte reads; the padding is
    float *ls_ = ld_ + 4 * NQ;                                                  //  zero, so a block past K has dd = 0 and adds nothing)
    unsigned char *lx = reinterpret_cast<unsigned char *>(ls_ + 4 * NQ);

    LLC_STAMP(0);
    GP_DECL(PRO);
    GemvPrologue<PRO, NT>::issue(pv, pw, psl, psb, xf, aux, KB, woven);

    // ---- this wave's slice of the weight stream: every load goes out before anything waits (nontemporal: read once per token)
    const int unit = blockIdx.x;
    const int qlo = (k * NQ) / NK, nq = ((k + 1) * NQ) / NK - qlo;              // (wave-uniform; nq <= QPW by the launcher's choice of NK)
    const int r = lane >> 2, g = lane & 3;
    ntv4u w[QPW];
    float dw[QPW], mw[QPW];
    auto load_group = [&](int grp) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int q = qlo + (i < nq ? i : 0);                               // (past the slice: a cache-hot dummy, never used -- unconditional
            const int64_t gq = (int64_t)grp * NQ + q;                           //  loads let the compiler count the ones in flight)
            w[i] = __builtin_nontemporal_load(reinterpret_cast<const ntv4u *>(qwd) + gq * 64 + lane);
            const int b = min(4 * q + g, KB - 1);                               // lane (row, g) fetches the scale of block 4q + g of its row
            dw[i] = __builtin_nontemporal_load(dW + ((int64_t)grp * KB + b) * 16 + r);
            if (Q41) mw[i] = __builtin_nontemporal_load(mW + ((int64_t)grp * KB + b) * 16 + r);
        }
    };
    load_group(unit * G2);
    LLC_STAMP(1);

    if constexpr (PRO != 0) {
        GemvPrologue<PRO, NT>::finish(pv, pw, psl, psb, xf, aux, KB, woven, lq, ld_, ls_, sh, ynorm, blockIdx.x == 0);
    } else {                                          // the activation is already Q8_0 (QA1 in HBM): copy it
        for (int i = threadIdx.x; i < KB * 2; i += NT) reinterpret_cast<uint4 *>(lq)[i] = reinterpret_cast<const uint4 *>(xq)[i];
        for (int i = threadIdx.x; i < KB; i += NT) {
            ld_[i] = xd[i];
            ls_[i] = Q41 ? xs[i] : 0.f;
        }
        __syncthreads();
    }
    // the lanes' view of the activation, once per workgroup: k-group g of block b, bytes (e0,e2,e4,e6 | e1,e3,e5,e7) -> (e0..e3 | e4..e7),
    // at LX[b >> 2][g][b & 3]: a lane reads the 32 bytes of its four blocks as two ds_read_b128 (four addresses per wave: broadcasts)
    for (int i = threadIdx.x; i < NQ * 16; i += NT) {
        const int q = i >> 4, gg = (i >> 2) & 3, blk = i & 3, b = 4 * q + blk;
        uint2 o = make_uint2(0, 0);
        if (b < KB) {
            const uint2 lh = *reinterpret_cast<const uint2 *>(lq + b * 32 + gg * 8);
            o = make_uint2(__builtin_amdgcn_perm(lh.y, lh.x, 0x05010400u), __builtin_amdgcn_perm(lh.y, lh.x, 0x07030602u));
        }
        *reinterpret_cast<uint2 *>(lx + (size_t)i * 8) = o;
    }
    if ((int)threadIdx.x < 4 * NQ - KB) {             // d_x / s_x of the blocks past K in a partial last quad: zero
        ld_[KB + threadIdx.x] = 0.f;
        ls_[KB + threadIdx.x] = 0.f;
    }
    __syncthreads();
    LLC_STAMP(2);

    const uint32_t m8 = 0xF0F0F0F0u;
    float y1 = 0.f;
    auto do_group = [&](auto GI) __attribute__((always_inline)) {
        constexpr int gi = decltype(GI)::value;
        // ---- order-free part, all waves at once: per block the two lane sums of this lane's k-group as floats, rn(d_w d_x), m_w
        float f0[QPW][4], f1[QPW][4], dd[QPW][4], ms[Q41 ? QPW : 1][4];
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            if (i < nq) {                                                       // (wave-uniform)
                const int q = qlo + i;
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
                if (Q41) { ms[i][0] = quad_bcast<0>(mw[i]); ms[i][1] = quad_bcast<1>(mw[i]); ms[i][2] = quad_bcast<2>(mw[i]); ms[i][3] = quad_bcast<3>(mw[i]); }
            }
        }
        if (gi == 0) LLC_STAMP(3);
        // ---- the chains, slice after slice: wave k continues from the state wave k - 1 left in LDS
        float a0 = 0.f, a1 = 0.f, summs = 0.f;
#pragma unroll 1
        for (int ph = 0; ph < NK; ++ph) {
            if (k == ph) {                                                      // (wave-uniform)
                if (ph > 0) { a0 = accs[lane][0]; a1 = accs[lane][1]; if (Q41) summs = accs[lane][2]; }
#pragma unroll
                for (int i = 0; i < QPW; ++i) {
                    if (i < nq) {
                        float sxv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (Q41) {
                            const float4 sx4 = *reinterpret_cast<const float4 *>(ls_ + 4 * (qlo + i));
                            sxv[0] = sx4.x; sxv[1] = sx4.y; sxv[2] = sx4.z; sxv[3] = sx4.w;
                        }
#pragma unroll
                        for (int blk = 0; blk < 4; ++blk) {
                            a0 = __fmaf_rn(dd[i][blk], f0[i][blk], a0);
                            a1 = __fmaf_rn(dd[i][blk], f1[i][blk], a1);
                            if (Q41) summs = __fmaf_rn(ms[i][blk], sxv[blk], summs);      // (a block past K: s_x = 0, m_w finite)
                        }
                    }
                }
                if (ph < NK - 1) { accs[lane][0] = a0; accs[lane][1] = a1; if (Q41) accs[lane][2] = summs; }
            }
            if (ph < NK - 1) __syncthreads();
        }
        if (gi == 0) LLC_STAMP(4);
        // ---- the row group is complete in its last wave: ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) over the quad of lanes that holds the
        // row (lane g holds accumulators 2g, 2g+1; every lane of the quad ends with the same bits), then the store / the PAIR epilogue
        if (k == NK - 1) {
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
                if (g == 0 && row < M) {
                    unsigned long long *slot = reinterpret_cast<unsigned long long *>(pair_ws) + (size_t)(unit >> 1) * 16 + r;
                    const unsigned long long mine = (1ull << 32) | (unsigned long long)__float_as_uint(v);
                    const unsigned long long old = __hip_atomic_exchange(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old >> 32) {
                        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float other = __uint_as_float((unsigned)old);
                        const float h1 = (unit & 1) ? other : v, h3 = (unit & 1) ? v : other;
                        const uint16_t hx = __half_as_ushort(__float2half_rn(h1));                    // GGML_FP32_TO_FP16
                        const float sl = __half2float(__ushort_as_half(aux2[hx]));                   // table_silu_f16
                        y[(unit >> 1) * 16 + r] = __fmul_rn(sl, h3);                                  // ggml_mul(silu, tmp)
                    }
                }
            } else if constexpr (PAIR == 1) {
                if (gi == 0) {
                    y1 = v;                                                    // w1 . x of this feature; its w3 row comes next
                } else if (g == 0 && row < M) {
                    const uint16_t hx = __half_as_ushort(__float2half_rn(y1));                // GGML_FP32_TO_FP16
                    const float sl = __half2float(__ushort_as_half(aux2[hx]));               // table_silu_f16
                    y[unit * 16 + r] = __fmul_rn(sl, v);                                      // ggml_mul(silu, tmp)
                }
            } else if (g == 0 && row < M) {
                if (resid) v = __fadd_rn(v, resid[row]);
                y[row] = v;
            }
        }
        if (PAIR == 1 && gi == 0) {
            // the w3 group's slice is requested only now.  Requested before the w1 chains it keeps both groups' registers alive: 214
            // VGPRs = two workgroups per CU = 1.3 rounds of the 688 workgroups, 24 us; squeezed under the 170-register cap of three
            // workgroups per CU it spills (22 us, also with the lane sums packed as int16 pairs); this order: 19.5 us
            // (profiles/r04_decode_exact.md).  What it costs: HBM idles while the whole launch sits in its w1 chain phase.
            load_group(unit * G2 + 1);
            __syncthreads();                            // (the chain state in LDS is free again for the second group)
        }
    };
    do_group(std::integral_constant<int, 0>{});
    if constexpr (PAIR == 1) do_group(std::integral_constant<int, 1>{});
    LLC_STAMP(5);
}
#ifdef LLC_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_llc_timing(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(llc_dbg), sizeof(long long) * 2048 * 8); }
#endif

// false: no QWD copy, or a shape outside the kernel's reach (rows too long for the slices' registers, activation beyond LDS)
// -> the caller takes round 3's kernel
template <int TYPE, int PRO, int PAIR>
static bool launch_llc(const fl_qtensor &W, const fl_qact *xq, float *y, hipStream_t st, const float *resid, const float *xf, const void *aux,
                       float *ynorm, int woven, const uint16_t *aux2, float *pair_ws = nullptr) {
    if (!W.qwd) return false;
    constexpr int G2 = PAIR == 1 ? 2 : 1;
    const int KB = W.KB, NQ = (KB + 3) / 4, units = W.M16 / 16 / G2;
    const size_t lds = (size_t)KB * 32 + (size_t)NQ * 32 + (size_t)NQ * 16 * 8;
    if (lds > 60 * 1024 || units < 1 || NQ > 88) return false;
    // (waves along K, quads a wave holds): 4 x 8 covers K <= 4096 with the fewest registers (three waves per SIMD), 4 x 11 K <= 5632,
    // 8 x 11 K <= 11264 -- but an 8-wave workgroup at 186 registers is ONE per CU: good for a matrix of <= 256 row groups (LLaMA-7B's w2:
    // 12.5 us against round 3's 15.1), bad for many groups of long rows (65B width, K = 8192: 57 / 93 us against 35 / 63 for
    // wq|wk|wv / w1|w3, scripts/dev/dec_ab.sh) -- those, and rows beyond 11264 (13B / 65B w2), stay on round 3's kernel
    if (NQ > 44 && units > 256) return false;
#define FL_LLC(NK, QPW)                                                                                                                   \
    hipLaunchKernelGGL((gemv1_q4_exact_llc_kernel<TYPE, NK, PRO, PAIR, QPW>), dim3(units), dim3(64 * NK), lds, st, W.M, units, KB, woven, \
                       W.qwd, W.d, xf, aux, W.m, xq ? xq->q : nullptr, xq ? xq->d : nullptr, xq ? xq->s : nullptr, y, resid, ynorm, aux2, pair_ws)
    if constexpr (TYPE == FL_TYPE_Q4_1) {             // Q4_1 carries m_w as well: 4 x 8 needs 174 registers = two waves per SIMD; 8 x 4 needs <= 126
        if (NQ <= 32) { FL_LLC(8, 4); return true; }   // (four): LLaMA-7B Q4_1 decode 412 -> 424 tok/s.  (Q4_0, 139 registers at 4 x 8: no gain)
    }
    if (NQ <= 32) FL_LLC(4, 8);
    else if (NQ <= 44) FL_LLC(4, 11);
    else FL_LLC(8, 11);
#undef FL_LLC
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
