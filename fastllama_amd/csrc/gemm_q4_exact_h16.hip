// gemm_q4_exact_h16.hip -- the reference-order ("exact") Q4 x Q8_0 matmul for prefill (N >= 9), round 4 form.
//
// What must be reproduced (ggml_vec_dot_q4_{0,1}_q8_0, AVX2 branch, /root/reference/lib/ggml.c:2445-2487, :2639-2689): per output
// 8 f32 accumulators, accumulator j taking  acc_j = fma(d_w * d_x, float(sum of the products of elements 4j..4j+3), acc_j)
// block after block, then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ the scalar chain summs = fma(m_w, s_x, summs) for Q4_1].
// The 8 fma per (output, block) are VALU work nothing can remove (128 v_fma_f32 per 32x32 tile and block = 256+ cycles; the kernel is compiled
// with packed f32 OFF: beside MFMAs v_pk_fma_f32 costs twice what two v_fma_f32 do, profiles/r04_ubench_coexec5.txt) and the
// 8 lane sums cost four v_mfma_f32_32x32x4_2b_f16 (two lane sums of a 32x32 tile each, 64 cycles: the matrix pipe delivers 32
// results per cycle whatever the shape) = 256 cycles; on gfx950 the two pipes of a SIMD do not overlap, so 512 + 32 (the d_w x d_x
// outer product, one MFMA per block PAIR) is the floor of the reference's order.  Round 3's kernel (gemm_q4_exact_mfma.hip)
// spent 1020: it unpacked nibbles -> f16 for every 32-column tile again (40 VALU ops per tile and block), converted the
// activations while staging them, and kept a wave's weights private.  Here NOTHING but the fma chain is left on the VALU:
//   * both operands are read as ready-made f16 MFMA fragments (q4_layout.h "H16 copies": WH16 built once per tensor, XH16 by
//     whoever produces the Q8_0 activations), 2 KiB per (32-row tile, block), moved HBM/L2 -> LDS by buffer_load ... lds;
//   * a workgroup = 4 waves = 64 x 64 outputs (2 x 2 wave tiles): every fragment staged in LDS is read by two waves;
//     K-steps of 4 blocks, TWO LDS stages of 34-36 KiB, one barrier per K-step; a K-step's nine DMA pieces per wave are issued one by one in
//     the shadow of the previous step's first MFMAs and have landed (s_waitcnt vmcnt(0) at the top of the step: nothing else is in flight) by then;
//   * the scales reach LDS by the same DMA (gathered per lane from the QW16 / QA16 planes); d_w x d_x of a block pair is ONE
//     v_mfma_f32_32x32x1_2b_f32 (exact products, rounded once: rn(d_w d_x));
//   * the order-free tails of the reference graph run as epilogues on the accumulators, bit-identical to the separate
//     kernels by construction: residual add; rope + K/V-cache stores (wq|wk|wv); silu * mul -> Q8_0 (woven w1|w3), which also
//     writes the XH16 operand of the w2 matmul.
// Q4_0: the WH16 values are 16 (nib - 8) and the stored scale is d / 16: fma(rn((d/16) d_x), 16 q, a) rounds the same real number
// as the reference's fma(rn(d d_x), q, a).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <cstdlib>
#include "q4_device.h"
#include "q4_kernels.h"
#include <type_traits>
#include "gemm_epi.h"

#pragma clang fp contract(off)

namespace fl {

typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));

template <int TYPE>
struct XH {
    static constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    static constexpr int KS = 4;                               // blocks per K-step (one barrier per step)
    static constexpr int OFF_B = 16384;                        // A: [2 row tiles][4 blocks][2 parts][1 KiB], then B the same
    static constexpr int OFF_SC = 32768;                       // d_w [4 groups][4 blocks][16 rows], d_x the same (, m_w, s_x): 1 KiB each
    static constexpr int STAGE = OFF_SC + (Q41 ? 4096 : 2048);
    static constexpr int LPW = 9;                              // DMA instructions per wave and stage: 8 fragment pieces + 1 scale piece
    static constexpr int ACT_BYTES = 32 * 64 * 4;              // f32 tile of the silu epilogue (reuses the ring)
    static constexpr int LDS_BYTES = 2 * STAGE;                // two stages: 68 / 72 KiB, two workgroups per CU
};

enum { EPI_PLAIN = 0, EPI_ROPE = 1, EPI_SILU = 2 };
#ifdef XH_TIMING   // development build only: per-workgroup clocks of the launch phases (scripts/dev/xh_timeline.py)
__device__ long long xh_dbg[8192 * 8];
#define XH_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) xh_dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define XH_STAMP(k) do {} while (0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// (no packed f32: beside MFMAs v_pk_fma_f32 costs twice what two v_fma_f32 do, profiles/r04_ubench_coexec5.txt)
#define XH_ATTR __attribute__((target("no-packed-fp32-ops")))
#define XIC(x) std::integral_constant<int, (x)>{}
template <int TYPE, int EPI>
__global__ __launch_bounds__(256, 2) XH_ATTR void gemm_q4_exact_h16_kernel(
    const uint16_t *wh, const float *dW, const float *mW, const uint16_t *xh, const float *xd, const float *xs, int N, int M,
    int MT32 /* 32-row tiles of wh */, int MGT /* 16-row groups of dW */, int NT32, int NGT, int KB, float *__restrict__ y, int ldy,
    const float *__restrict__ resid, int ldr, GemmSiluEpi epi) {
    using C = XH<TYPE>;
    constexpr bool Q41 = C::Q41;
    constexpr int KS = C::KS;
    XH_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;

    // ---- XCD-aware bijective remap of the tile id (as gemm_q4_mfma32.hip): workgroup b runs on XCD b % 8; the column tiles that
    //      share a weight row panel get consecutive ids on ONE XCD, so the panel comes from HBM once and is re-read from that L2.
    //      A workgroup is PERSISTENT when the launch has fewer workgroups than tiles (launch_xh: one per residency slot): it takes the
    //      tiles vb = blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8: vb & 7 stays the XCD it runs on), and the DMA of
    //      a tile's LAST K-step, which used to fetch a K-step past the row, fetches the first K-step of the NEXT tile instead -- that
    //      tile starts with its operands in LDS, and the ramp of every tile but the first is gone.
    const int tiles_m = (MT32 + 1) >> 1, tiles_n = (NT32 + 1) >> 1, nwg = tiles_m * tiles_n;
    auto remap = [&](int vb) XH_ATTR __attribute__((always_inline)) {
        const int q = nwg >> 3, rem = nwg & 7, xcd = vb & 7, k = vb >> 3;
        return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    };
    int vbid = blockIdx.x;
    int bid = remap(vbid);
    int tn = bid % tiles_n, tm = bid / tiles_n;

    // ---- DMA plan.  Per K-step (4 blocks) a wave moves 8 fragment pieces of 1 KiB (piece p = wave + 4 j of 32: side p >> 4, tile
    //      (p >> 3) & 1, block (p >> 1) & 3, part p & 1) and one scale piece of 1 KiB (Q4_0: waves 0 / 1 d_w / d_x, waves 2 / 3 the same
    //      again -- every wave issues the same number of loads, so one counted wait is valid for all; Q4_1: d_w, d_x, m_w, s_x).
    //      Everything wave-uniform travels in SGPRs (descriptor, scalar offset); the lane part is ONE VGPR per kind: lane * 16 for
    //      the fragments, the (group, block, 4 rows) gather offset for the scales.  Tiles / groups past the tensor re-read the last
    //      one (their outputs are never stored); blocks past K re-read block K - 1 for the fragments and read ZERO scales on the
    //      activation side (bounds check of the descriptor): d_x = s_x = 0, the block contributes nothing.
    uint32_t lane16 = (uint32_t)lane * 16u;
    const uint64_t wbp = (uint64_t)(uintptr_t)wh, xbp = (uint64_t)(uintptr_t)xh;
    v4i rA = v4i{(int)(uint32_t)wbp, (int)((wbp >> 32) & 0xFFFF), (int)((uint32_t)MT32 * (uint32_t)KB * 2048u), 0x00020000};
    v4i rB = v4i{(int)(uint32_t)xbp, (int)((xbp >> 32) & 0xFFFF), (int)((uint32_t)NT32 * (uint32_t)KB * 2048u), 0x00020000};
    const int sc_kind = Q41 ? wave : (wave & 1);                               // 0 d_w, 1 d_x, 2 m_w, 3 s_x
    bool sc_act = (sc_kind & 1) != 0;
    const float *scp = sc_kind == 0 ? dW : sc_kind == 1 ? xd : sc_kind == 2 ? mW : xs;
    const uint64_t sbp = (uint64_t)(uintptr_t)scp;
    const int sc_ng = sc_act ? NGT : MGT;
    v4i rS = v4i{__builtin_amdgcn_readfirstlane((int)(uint32_t)sbp), __builtin_amdgcn_readfirstlane((int)((sbp >> 32) & 0xFFFF)),
                       __builtin_amdgcn_readfirstlane((int)((uint32_t)sc_ng * (uint32_t)KB * 64u)), 0x00020000};
    // lane -> (group lane >> 4, 16-byte chunk lane & 15 of the group's 4 blocks x 16 rows): byte offset relative to block kb0 of group 0
    const int sc_blk = (lane & 15) >> 2;
    auto sc_voff_of = [&](int tn_, int tm_) XH_ATTR __attribute__((always_inline)) {
        const int sc_g = min((sc_act ? tn_ : tm_) * 4 + (lane >> 4), sc_ng - 1);
        return ((uint32_t)sc_g * (uint32_t)KB + (uint32_t)sc_blk) * 64u + (uint32_t)(lane & 3) * 16u;
    };
    int sc_loff = C::OFF_SC + (Q41 ? ((sc_kind & 1) * 1024 + (sc_kind >> 1) * 2048) : sc_kind * 1024);
    uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // fragment piece j (0..7) of this wave for the K-step starting at block kb0, into stage st.  With p = wave + 4 j: side = j >> 2,
    // tile = (j >> 1) & 1 are compile-time, block = (wave >> 1) + 2 (j & 1), part = wave & 1 -- scalars.
    const int wb = __builtin_amdgcn_readfirstlane(wave >> 1), wp = __builtin_amdgcn_readfirstlane(wave & 1);
    // byte offset of block 0, part wp of tiles A0, A1, B0, B1 of a workgroup tile
    auto tbase_of = [&](int tn_, int tm_, uint32_t (&tb)[4]) XH_ATTR __attribute__((always_inline)) {
        tb[0] = (uint32_t)__builtin_amdgcn_readfirstlane(min(tm_ * 2, MT32 - 1) * KB * 2048 + wp * 1024);
        tb[1] = (uint32_t)__builtin_amdgcn_readfirstlane(min(tm_ * 2 + 1, MT32 - 1) * KB * 2048 + wp * 1024);
        tb[2] = (uint32_t)__builtin_amdgcn_readfirstlane(min(tn_ * 2, NT32 - 1) * KB * 2048 + wp * 1024);
        tb[3] = (uint32_t)__builtin_amdgcn_readfirstlane(min(tn_ * 2 + 1, NT32 - 1) * KB * 2048 + wp * 1024);
    };
    uint32_t tbase[4];
    tbase_of(tn, tm, tbase);
    uint32_t sc_voff = sc_voff_of(tn, tm);
    uint32_t dstw = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds0 + (wb * 2 + wp) * 1024);
    auto fill_frag = [&](auto JJ, int st, int kb0, const uint32_t (&ftb)[4]) XH_ATTR __attribute__((always_inline)) {
        constexpr int j = decltype(JJ)::value, side = j >> 2, t = (j >> 1) & 1;
        const int kb = min(kb0 + wb + 2 * (j & 1), KB - 1);
        const uint32_t soff = ftb[side * 2 + t] + (uint32_t)kb * 2048u;
        const uint32_t dst = dstw + (uint32_t)(st * C::STAGE + side * C::OFF_B + (t * 4 + 2 * (j & 1)) * 2048);
        const uint32_t vo = lane16;                                            // (named copies: a generic lambda does not capture a
        const v4i rs = side ? rB : rA;                                         //  variable that only an asm operand mentions)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "m0");
    };
    auto fill_scale = [&](int st, int kb0, uint32_t fsc_voff) XH_ATTR __attribute__((always_inline)) {
        uint32_t vo = fsc_voff;
        if (sc_act && kb0 + sc_blk >= KB) vo = 0x80000000u;                     // activation scales of a block past K: zero
        else if (kb0 + sc_blk >= KB) vo = fsc_voff - (uint32_t)(kb0 + sc_blk - (KB - 1)) * 64u;  // weight scales: block K - 1 again (finite)
        const uint32_t dst = lds0 + (uint32_t)(st * C::STAGE + sc_loff);
        const v4i rs = rS;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(dst), "v"(vo), "s"(rs), "s"(kb0 * 64) : "memory", "m0");
    };

    v16f acc[8], summs;
    const v32f zero32 = {};

    // per-lane LDS offsets inside a stage: fragments of block u at + u * 2048 (+ 1024 for the second part); scales of block u
    const int a_off = wr * 8192 + lane * 16, b_off = C::OFF_B + wc * 8192 + lane * 16;
    const int sw_off = C::OFF_SC + ((2 * wr + (i >> 4)) * 4 + h) * 64 + (i & 15) * 4;          // + 128 for the second pair, + 2048: m_w
    const int sx_off = C::OFF_SC + 1024 + ((2 * wc + (i >> 4)) * 4 + h) * 64 + (i & 15) * 4;   // + 2048: s_x

    const int nsteps = (KB + KS - 1) / KS;
    fill_frag(XIC(0), 0, 0, tbase); fill_frag(XIC(1), 0, 0, tbase); fill_frag(XIC(2), 0, 0, tbase); fill_frag(XIC(3), 0, 0, tbase);
    fill_frag(XIC(4), 0, 0, tbase); fill_frag(XIC(5), 0, 0, tbase); fill_frag(XIC(6), 0, 0, tbase); fill_frag(XIC(7), 0, 0, tbase);
    fill_scale(0, 0, sc_voff);
    int cur = 0;
#pragma unroll 1
  for (;;) {                                                                   // ---- tiles of this workgroup
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) summs[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[j]));    // (opaque zeros: no peeled first trip)
    // the tile after this one: its bases replace this tile's for the DMA issued during the LAST K-step (no next tile: this tile's again --
    // a K-step nobody reads, as before)
    const int nvb = vbid + (int)gridDim.x;
    const bool has_next = nvb < nwg;
    const int nbid = remap(has_next ? nvb : vbid), ntn = nbid % tiles_n, ntm = nbid / tiles_n;
    uint32_t ntbase[4];
    tbase_of(ntn, ntm, ntbase);
    const uint32_t nsc_voff = sc_voff_of(ntn, ntm);
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's pieces of K-step t have landed
        __builtin_amdgcn_s_barrier();                                          // ... everyone's have, and everyone is done with step t - 1
        if (t == 0) XH_STAMP(1);
        if (t == 1) XH_STAMP(2);
        if (t == 2) XH_STAMP(3);
        const unsigned char *base = smem + cur * C::STAGE;
        const bool last = t + 1 == nsteps;
        const int nst = cur ^ 1, nkb = last ? 0 : (t + 1) * KS;                // K-step t + 1 (the next tile's step 0) goes into the stage step t - 1 occupied
        uint32_t ftb[4];                                                       // (wave-uniform selects: SGPRs)
#pragma unroll
        for (int q = 0; q < 4; ++q) ftb[q] = last ? ntbase[q] : tbase[q];
        const uint32_t fsc_voff = last ? nsc_voff : sc_voff;
        // Software pipeline inside the wave: the lane-sum MFMA of group g + 1 is issued BEFORE the 32 fma of group g (two D tiles),
        // so a wave alone on its SIMD (its neighbour waiting at a barrier) still keeps both pipes fed: coexec5, 277 vs 529 ns.
        // fragments: 8 bytes per lane and MFMA (steps 2p, 2p + 1 of a block sit side by side in a 16-byte slot), read one group ahead
        auto frag_a = [&](auto G) XH_ATTR __attribute__((always_inline)) -> v2u32 {
            constexpr int g = decltype(G)::value, u = g >> 2, s = g & 3;
            return *reinterpret_cast<const v2u32 *>(base + a_off + u * 2048 + (s >> 1) * 1024 + (s & 1) * 8);
        };
        auto frag_b = [&](auto G) XH_ATTR __attribute__((always_inline)) -> v2u32 {
            constexpr int g = decltype(G)::value, u = g >> 2, s = g & 3;
            return *reinterpret_cast<const v2u32 *>(base + b_off + u * 2048 + (s >> 1) * 1024 + (s & 1) * 8);
        };
        // B128 (Q4_0): a lane's two MFMA steps of a 16-byte slot in ONE ds_read_b128 (two ds_read_b64 put lanes l and l + 8 on the same
        // banks: half of the kernel's LDS cycles were conflicts), one slot ahead; Q4_1 has no registers left for it
        constexpr bool B128 = !Q41;
        typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
        auto slot_a = [&](auto P) XH_ATTR __attribute__((always_inline)) -> v4u32 {
            constexpr int p = decltype(P)::value;
            return *reinterpret_cast<const v4u32 *>(base + a_off + (p >> 1) * 2048 + (p & 1) * 1024);
        };
        auto slot_b = [&](auto P) XH_ATTR __attribute__((always_inline)) -> v4u32 {
            constexpr int p = decltype(P)::value;
            return *reinterpret_cast<const v4u32 *>(base + b_off + (p >> 1) * 2048 + (p & 1) * 1024);
        };
        v4u32 sa = {}, sb = {}, san = {}, sbn = {};
        v2u32 fa = {}, fb = {}, fan = {}, fbn = {};
        if (B128) { sa = slot_a(XIC(0)); sb = slot_b(XIC(0)); san = slot_a(XIC(1)); sbn = slot_b(XIC(1)); }
        else { fa = frag_a(XIC(0)); fb = frag_b(XIC(0)); fan = frag_a(XIC(1)); fbn = frag_b(XIC(1)); }
        float dw = *reinterpret_cast<const float *>(base + sw_off), dx = *reinterpret_cast<const float *>(base + sx_off);
        v32f P = __builtin_amdgcn_mfma_f32_32x32x1f32(dw, dx, zero32, 0, 0, 0);             // rn(d_w d_x) of blocks 0 (regs 0..15), 1
        v32f D0;       // (ONE lane-sum tile alive: issuing the next MFMA ahead of the fma of the current one -- two tiles -- buys nothing at two waves per SIMD
                       //  and does not fit the register file; the setprio / no-LDS / no-DMA / no-barrier ablations: profiles/r05_gemm_exact.md)
        auto group = [&](auto G) XH_ATTR __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value, u = g >> 2, s = g & 3;
            v32f &Dg = D0;
            if (B128) {
                const v2u32 ga = (g & 1) ? v2u32{sa.z, sa.w} : v2u32{sa.x, sa.y}, gb = (g & 1) ? v2u32{sb.z, sb.w} : v2u32{sb.x, sb.y};
                D0 = __builtin_amdgcn_mfma_f32_32x32x4f16(__builtin_bit_cast(v4h, ga), __builtin_bit_cast(v4h, gb), zero32, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (g & 1) {                         // the slot is used up: the next one moves in, the one after it is requested
                    sa = san; sb = sbn;
                    if ((g >> 1) + 2 < 2 * KS) {
                        san = slot_a(XIC((g >> 1) + 2 < 2 * KS ? (g >> 1) + 2 : 0));
                        sbn = slot_b(XIC((g >> 1) + 2 < 2 * KS ? (g >> 1) + 2 : 0));
                    }
                }
            } else {
                D0 = __builtin_amdgcn_mfma_f32_32x32x4f16(__builtin_bit_cast(v4h, fa), __builtin_bit_cast(v4h, fb), zero32, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                fa = fan; fb = fbn;
                if (g + 2 < 4 * KS) {
                    fan = frag_a(XIC(g + 2 < 4 * KS ? g + 2 : 0));
                    fbn = frag_b(XIC(g + 2 < 4 * KS ? g + 2 : 0));
                }
            }
            if (g == 4) {                        // scales of the second block pair
                dw = *reinterpret_cast<const float *>(base + sw_off + 128);
                dx = *reinterpret_cast<const float *>(base + sx_off + 128);
            }
            if (g < 8) fill_frag(XIC(g < 8 ? g : 0), nst, nkb, ftb);     // ... and the DMA of K-step t + 1
            if (g == 8) fill_scale(nst, nkb, fsc_voff);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[2 * s][e] = __builtin_fmaf(P[16 * (u & 1) + e], Dg[e], acc[2 * s][e]);
                acc[2 * s + 1][e] = __builtin_fmaf(P[16 * (u & 1) + e], Dg[16 + e], acc[2 * s + 1][e]);
            }
            // Pin the order: these 32 fma complete before anything of the next group (left alone, the compiler sinks them).
            asm volatile("" : "+v"(acc[2 * s]), "+v"(acc[2 * s + 1]));
            if (g == 7 && KS > 2) {              // blocks 2, 3: their d_w x d_x (after the last use of the first pair's)
                asm volatile("" : "+v"(dw), "+v"(dx), "+v"(acc[2 * s]));
                P = __builtin_amdgcn_mfma_f32_32x32x1f32(dw, dx, zero32, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (Q41) {                                                             // summs: one chain in block order (k = 0, 1 per MFMA)
            const float mw0 = *reinterpret_cast<const float *>(base + sw_off + 2048), sx0 = *reinterpret_cast<const float *>(base + sx_off + 2048);
            const float mw1 = *reinterpret_cast<const float *>(base + sw_off + 2048 + 128), sx1 = *reinterpret_cast<const float *>(base + sx_off + 2048 + 128);
            summs = __builtin_amdgcn_mfma_f32_32x32x2f32(mw0, sx0, summs, 0, 0, 0);
            summs = __builtin_amdgcn_mfma_f32_32x32x2f32(mw1, sx1, summs, 0, 0, 0);
        }
        group(XIC(0)); group(XIC(1)); group(XIC(2)); group(XIC(3)); group(XIC(4)); group(XIC(5)); group(XIC(6)); group(XIC(7));
        group(XIC(8)); group(XIC(9)); group(XIC(10)); group(XIC(11)); group(XIC(12)); group(XIC(13)); group(XIC(14)); group(XIC(15));
        cur ^= 1;
    }
    XH_STAMP(4);
    auto epilogue = [&]() XH_ATTR __attribute__((always_inline)) {

    // ---- ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ summs]: C layout col = i, row = (e & 3) + 8 (e >> 2) + 4 h ----
    v16f out;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float v = __fadd_rn(__fadd_rn(__fadd_rn(acc[0][e], acc[4][e]), __fadd_rn(acc[2][e], acc[6][e])),
                            __fadd_rn(__fadd_rn(acc[1][e], acc[5][e]), __fadd_rn(acc[3][e], acc[7][e])));
        if (Q41) v = __fadd_rn(v, summs[e]);
        out[e] = v;
    }
    const int n = (tn * 2 + wc) * 32 + i;                                      // this lane's column (token)
    const int rowbase = (tm * 2 + wr) * 32;

    if (EPI == EPI_SILU) {
        // ---- silu(w1 x) * (w3 x) -> Q8_0 (ggml_silu + ggml_mul, lib/llama.cpp:428-431, then quantize_row_q8_0 of the w2 matmul's
        //      INIT phase).  W is woven by 16-row groups: rows 0..15 of a wave's 32 are w1 of 16 features, rows 16..31 w3 of the same;
        //      the workgroup's 64 rows are ONE 32-feature block of every column.
        // [32 features][64 columns] f32 in the stage the last K-step was read from (the other one is receiving the next tile's first
        // K-step).  Barriers without a vmcnt wait: that DMA stays in flight under the epilogue.
        float *act = reinterpret_cast<float *>(smem + (cur ^ 1) * C::STAGE);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                          // every wave is done with that stage
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a1 = out[4 * g + e], a3 = out[4 * (g + 2) + e];
                const uint16_t hx = __half_as_ushort(__float2half_rn(a1));                      // GGML_FP32_TO_FP16
                const float sl = __half2float(__ushort_as_half(epi.silu_tab[hx]));              // table_silu_f16
                act[(wr * 16 + 8 * g + 4 * h + e) * 64 + wc * 32 + i] = __fmul_rn(sl, a3);     // ggml_mul(silu, tmp)
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // four threads per column, 8 features each; amax and the integer sum meet through DPP quad reductions
        const int tid = threadIdx.x, nl = tid >> 2, part = tid & 3;
        const int nn = tn * 64 + nl, gfb = tm;                                 // column, 32-feature block
        float v[8];
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = act[(part * 8 + e) * 64 + nl];
            amax = fmaxf(amax, fabsf(v[e]));
        }
        amax = quad_max_f32(amax);
        const float dd = __fdiv_rn(amax, 127.0f);
        const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
        int qi[8], sum = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qi[e] = (int)rintf(__fmul_rn(v[e], id));
            sum += qi[e];
        }
        sum = quad_sum_i32(sum);
        if (nn < NGT * 16 && gfb < epi.KBo) {
            const int c = nn & 15;
            const int64_t cb = ((int64_t)(nn >> 4) * epi.KBo + gfb) * 16 + c;
            auto pk = [](int a, int b, int cc, int d) -> uint32_t {
                return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(cc & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
            };
            if (epi.oq)        // QA16: k-group `part` of the block, bytes e0,e2,e4,e6,e1,e3,e5,e7
                *reinterpret_cast<uint2 *>(epi.oq + cb * 32 + qw16_pos(c, part) * 8) =
                    make_uint2(pk(qi[0], qi[2], qi[4], qi[6]), pk(qi[1], qi[3], qi[5], qi[7]));
            if (epi.oh) {      // XH16: MFMA step s = part, halves h = 0 / 1 = elements 0..3 / 4..7 of the k-group
                uint16_t *ob = epi.oh + (((int64_t)(nn >> 5) * epi.KBo + gfb) * 2 + (part >> 1)) * 512 + (part & 1) * 4;
                auto hb = [](int q) -> uint32_t { return (uint32_t)__half_as_ushort(__int2half_rn(q)); };
                *reinterpret_cast<uint2 *>(ob + ((nn & 31)) * 8) = make_uint2(hb(qi[0]) | (hb(qi[1]) << 16), hb(qi[2]) | (hb(qi[3]) << 16));
                *reinterpret_cast<uint2 *>(ob + ((nn & 31) + 32) * 8) = make_uint2(hb(qi[4]) | (hb(qi[5]) << 16), hb(qi[6]) | (hb(qi[7]) << 16));
            }
            if (part == 0) {
                epi.od[cb] = dd;
                epi.os[cb] = __fmul_rn(dd, (float)sum);
            }
        }
        return;
    }
    if (n >= N) return;
    if (EPI == EPI_ROPE) {
        // ---- rope on Q (-> y) and K (-> the K-cache row of the token's position), V transposed into the V cache: the arithmetic of
        //      rope_kv_kernel (ggml_rope + the two ggml_cpy, lib/llama.cpp:328-347).  A lane holds features row0..row0+3 = two pairs.
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row0 = rowbase + 8 * g + 4 * h;
            if (row0 >= M) continue;
            const float o0 = out[4 * g], o1 = out[4 * g + 1], o2 = out[4 * g + 2], o3 = out[4 * g + 3];
            const int part = row0 / epi.El, f = row0 - part * epi.El, pos = epi.n_past + n;
            if (part < 2) {
                const float2 *cs = epi.rope_tab + (int64_t)pos * (epi.D >> 1) + ((f % epi.D) >> 1);
                const float2 c0 = cs[0], c1 = cs[1];
                float4 q;
                q.x = __fmaf_rn(o0, c0.x, -__fmul_rn(o1, c0.y));
                q.y = __fmaf_rn(o0, c0.y, __fmul_rn(o1, c0.x));
                q.z = __fmaf_rn(o2, c1.x, -__fmul_rn(o3, c1.y));
                q.w = __fmaf_rn(o2, c1.y, __fmul_rn(o3, c1.x));
                float *dst = part == 0 ? y + (int64_t)n * ldy + row0 : epi.kc + (int64_t)pos * epi.El + f;
                *reinterpret_cast<float4 *>(dst) = q;
            } else {
                epi.vc[(int64_t)(f + 0) * epi.n_ctx + pos] = o0;
                epi.vc[(int64_t)(f + 1) * epi.n_ctx + pos] = o1;
                epi.vc[(int64_t)(f + 2) * epi.n_ctx + pos] = o2;
                epi.vc[(int64_t)(f + 3) * epi.n_ctx + pos] = o3;
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = rowbase + 8 * g + 4 * h;                               // rows row .. row + 3
        if (row >= M) continue;
        float4 o = make_float4(out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3]);
        float *yp = y + (int64_t)n * ldy + row;
        const float *rp = resid ? resid + (int64_t)n * ldr + row : nullptr;
        if (row + 3 < M && (ldy & 3) == 0 && (!resid || (ldr & 3) == 0)) {
            if (rp) {
                const float4 rr = *reinterpret_cast<const float4 *>(rp);
                o.x = __fadd_rn(o.x, rr.x); o.y = __fadd_rn(o.y, rr.y); o.z = __fadd_rn(o.z, rr.z); o.w = __fadd_rn(o.w, rr.w);
            }
            *reinterpret_cast<float4 *>(yp) = o;
        } else {
            const float ov[4] = {o.x, o.y, o.z, o.w};
            for (int k = 0; k < 4 && row + k < M; ++k) yp[k] = rp ? __fadd_rn(ov[k], rp[k]) : ov[k];
        }
    }
    };      // epilogue
    epilogue();
    XH_STAMP(5);
    if (!has_next) break;
    vbid = nvb; tn = ntn; tm = ntm;
#pragma unroll
    for (int q = 0; q < 4; ++q) tbase[q] = ntbase[q];
    sc_voff = nsc_voff;
    // (this tile's first K-step is in flight or landed; the loop's first wait covers it -- and the epilogue's stores)
  }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the K-step fetched past the last tile (unused) must have landed before the LDS is released
}
#else
template <int TYPE, int EPI>
__global__ void gemm_q4_exact_h16_kernel(const uint16_t *, const float *, const float *, const uint16_t *, const float *, const float *, int,
                                         int, int, int, int, int, int, float *, int, const float *, int, GemmSiluEpi) {}
#endif

// ---------------------------------------------------------------- the H16 copies (q4_layout.h) ----------------------------------
// byte offset of (tile, block, part, lane) in a WH16 / XH16 buffer is ((tile * KB + block) * 2 + part) * 1024 + lane * 16
__device__ __forceinline__ uint32_t h16_of(int v) { return (uint32_t)__half_as_ushort(__int2half_rn(v)); }

template <int TYPE>
__global__ __launch_bounds__(256) void qw16_to_h16_kernel(const uint4 *__restrict__ qs, uint16_t *__restrict__ wh, int64_t n_rows /* M16 * KB */,
                                                          int KB) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (16-row group, block, row)
    if (u >= n_rows) return;
    const int r16 = (int)(u & 15);
    const int64_t gb = u >> 4;
    const int grp = (int)(gb / KB), b = (int)(gb % KB);
    const uint4 raw = qs[u];
    const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
    const int tile = grp >> 1, i = (grp & 1) * 16 + r16;
    unsigned char *blk = reinterpret_cast<unsigned char *>(wh) + ((int64_t)tile * KB + b) * 2048;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int g = p ^ (((r16 >> 3) & 1) << 1);                    // logical k-group stored at dword position p = MFMA step s
        int el[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t bb = (dw[p] >> (8 * j)) & 0xFF;
            int lo = (int)(bb & 15), hi = (int)(bb >> 4);
            if (TYPE == FL_TYPE_Q4_0) {                               // stored nibble = nib ^ 8; value 16 (nib - 8)
                lo = (((lo ^ 8) - 8)) * 16;
                hi = (((hi ^ 8) - 8)) * 16;
            }
            el[2 * j] = lo;
            el[2 * j + 1] = hi;
        }
        unsigned char *dst = blk + (g >> 1) * 1024 + (g & 1) * 8;
        *reinterpret_cast<uint2 *>(dst + i * 16) = make_uint2(h16_of(el[0]) | (h16_of(el[1]) << 16), h16_of(el[2]) | (h16_of(el[3]) << 16));
        *reinterpret_cast<uint2 *>(dst + (i + 32) * 16) = make_uint2(h16_of(el[4]) | (h16_of(el[5]) << 16), h16_of(el[6]) | (h16_of(el[7]) << 16));
    }
}

size_t wh16_bytes(const fl_qtensor &W) { return (size_t)((W.M16 + 31) / 32) * (size_t)W.KB * 2048; }
size_t xh16_bytes(int N, int K) { return (size_t)((N + 31) / 32) * (size_t)(K / FL_QK) * 2048; }

hipError_t qw16_to_h16(const fl_qtensor &W, uint16_t *wh, hipStream_t st) {
    const int64_t n = (int64_t)W.M16 * W.KB;
    if (n == 0) return hipSuccess;
    const int64_t nb = (n + 255) / 256;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    if (W.M16 % 32) {                                                 // the last tile's rows 16..31 do not exist: zeros
        const size_t tail = (size_t)W.KB * 2048;
        hipError_t e = hipMemsetAsync(reinterpret_cast<unsigned char *>(wh) + wh16_bytes(W) - tail, 0, tail, st);
        if (e != hipSuccess) return e;
    }
    if (W.type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(qw16_to_h16_kernel<FL_TYPE_Q4_0>, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const uint4 *>(W.qs), wh, n, W.KB);
    else
        hipLaunchKernelGGL(qw16_to_h16_kernel<FL_TYPE_Q4_1>, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const uint4 *>(W.qs), wh, n, W.KB);
    return hipGetLastError();
}

// QA16 -> XH16: one thread per (column group, block, column); the 8-byte position p of QA16 holds k-group p ^ (((col >> 3) & 1) << 1),
// bytes e0,e2,e4,e6,e1,e3,e5,e7.  Column groups past N16 (the second half of the last 32-column tile) are written as zeros.
__global__ __launch_bounds__(256) void qa16_to_h16_kernel(const uint4 *__restrict__ q, uint16_t *__restrict__ xh, int64_t n_cols /* NG32 * 2 * KB * 16 */,
                                                          int KB, int NGT) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_cols) return;
    const int c16 = (int)(u & 15);
    const int64_t gb = u >> 4;
    const int grp = (int)(gb / KB), b = (int)(gb % KB);
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (grp < NGT) { r0 = q[2 * u]; r1 = q[2 * u + 1]; }
    const uint32_t dw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const int tile = grp >> 1, i = (grp & 1) * 16 + c16;
    unsigned char *blk = reinterpret_cast<unsigned char *>(xh) + ((int64_t)tile * KB + b) * 2048;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int g = p ^ (((c16 >> 3) & 1) << 1);
        int el[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            el[2 * t] = (int)(int8_t)((dw[2 * p] >> (8 * t)) & 0xFF);        // even elements
            el[2 * t + 1] = (int)(int8_t)((dw[2 * p + 1] >> (8 * t)) & 0xFF);  // odd elements
        }
        unsigned char *dst = blk + (g >> 1) * 1024 + (g & 1) * 8;
        *reinterpret_cast<uint2 *>(dst + i * 16) = make_uint2(h16_of(el[0]) | (h16_of(el[1]) << 16), h16_of(el[2]) | (h16_of(el[3]) << 16));
        *reinterpret_cast<uint2 *>(dst + (i + 32) * 16) = make_uint2(h16_of(el[4]) | (h16_of(el[5]) << 16), h16_of(el[6]) | (h16_of(el[7]) << 16));
    }
}

hipError_t qa16_to_h16(const fl_qact &xq, int N, hipStream_t st) {
    if (!xq.h16) return hipErrorInvalidValue;
    const int NGT = fl_roundup(N, 16) / 16, NG2 = (N + 31) / 32 * 2;
    const int64_t n = (int64_t)NG2 * xq.KB * 16;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qa16_to_h16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(xq.q), xq.h16, n,
                       xq.KB, NGT);
    return hipGetLastError();
}

// ---------------------------------------------------------------- launch ---------------------------------------------------------
bool gemm_q4_exact_h16_supports(const fl_qtensor &W, const fl_qact &xq, int N) {
    if (!W.h16 || !xq.h16 || N < 1 || W.KB < 1) return false;
    return wh16_bytes(W) < (1ull << 31) && xh16_bytes(N, W.K) < (1ull << 31);    // 32-bit buffer offsets
}

template <int TYPE, int EPI>
static hipError_t launch_xh(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid, int ldr,
                            const GemmSiluEpi &epi) {
    using C = XH<TYPE>;
    static_assert(C::ACT_BYTES <= C::LDS_BYTES && 2 * C::LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    static bool attr_set = false;                                              // (more than 64 KiB of dynamic LDS must be asked for)
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_q4_exact_h16_kernel<TYPE, EPI>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int MT32 = (W.M16 + 31) / 32, NT32 = (N + 31) / 32;
    const int tiles = ((MT32 + 1) / 2) * ((NT32 + 1) / 2);
    // one workgroup per residency slot (two per CU), each walking its tiles with the next tile's first K-step requested ahead
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        slots = 2 * cus / 8 * 8;                                               // (a multiple of the 8 XCDs: a workgroup's tiles stay on its XCD)
    }
    const int grid = tiles < slots ? tiles : slots;
    hipLaunchKernelGGL((gemm_q4_exact_h16_kernel<TYPE, EPI>), dim3(grid), dim3(256), C::LDS_BYTES, st, W.h16, W.d, W.m, xq.h16, xq.d, xq.s, N,
                       W.M, MT32, W.M16 / 16, NT32, fl_roundup(N, 16) / 16, W.KB, y, ldy, resid, ldr, epi);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_xh_t(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid, int ldr,
                              const GemmSiluEpi &epi) {
    if (!gemm_q4_exact_h16_supports(W, xq, N)) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_xh<FL_TYPE_Q4_0, EPI>(W, xq, N, y, ldy, st, resid, ldr, epi)
                                  : launch_xh<FL_TYPE_Q4_1, EPI>(W, xq, N, y, ldy, st, resid, ldr, epi);
}

hipError_t gemm_q4_exact_h16(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st, const float *resid, int ldr) {
    return launch_xh_t<EPI_PLAIN>(W, xq, N, y, ldy, st, resid, ldr, GemmSiluEpi{});
}

// W = wq|wk|wv stacked ([3 El][K]); y <- rope(Q) rows ([N][ldy], the first El columns), K-cache rows n_past.. <- rope(K),
// transposed V-cache columns n_past.. <- V
hipError_t gemm_q4_exact_h16_qkv(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, const float *rope_tab, float *kc, float *vc,
                                 int El, int D, int n_past, int n_ctx, hipStream_t st) {
    if (!rope_tab || W.M != 3 * El || El % 4 != 0 || D % 4 != 0 || (ldy & 3) != 0) return hipErrorInvalidValue;
    GemmSiluEpi epi{};
    epi.rope_tab = reinterpret_cast<const float2 *>(rope_tab);
    epi.kc = kc; epi.vc = vc; epi.El = El; epi.D = D; epi.n_past = n_past; epi.n_ctx = n_ctx;
    return launch_xh_t<EPI_ROPE>(W, xq, N, y, ldy, st, nullptr, 0, epi);
}

// W = w1|w3 woven by 16-row groups; out <- Q8_0(silu(w1 x) * (w3 x)): QA16 planes (q optional) + the XH16 copy
hipError_t gemm_q4_exact_h16_silu(const fl_qtensor &W, const fl_qact &xq, int N, const uint16_t *silu_tab, const fl_qact &out, hipStream_t st) {
    if (!silu_tab || W.M % 64 != 0 || !out.d || !out.s || (!out.q && !out.h16)) return hipErrorInvalidValue;
    GemmSiluEpi epi{};
    epi.silu_tab = silu_tab; epi.oq = out.q; epi.od = out.d; epi.os = out.s; epi.oh = out.h16; epi.KBo = W.M / 64;
    return launch_xh_t<EPI_SILU>(W, xq, N, nullptr, 4, st, nullptr, 0, epi);
}

}  // namespace fl
#ifdef XH_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_xh_timing(long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fl::xh_dbg), sizeof(long long) * (size_t)n * 8); }
#endif
