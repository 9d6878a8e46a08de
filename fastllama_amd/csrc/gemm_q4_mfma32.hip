// gemm_q4_mfma32.hip -- prefill path (N >= 9) of ggml_compute_forward_mul_mat_q_f32 on the gfx950 double-rate
// integer MFMA  v_mfma_i32_32x32x32_i8  (/root/reference/lib/ggml.c:7928-8176, COMPUTE phase :8127-8163).
//
//   y[n][m] = sum_b  (d_w[m,b] * d_x[n,b]) * isum[m,n,b]   (+ m_w[m,b] * s_x[n,b] for Q4_1)
//   isum[m,n,b] = sum_{i<32} w_i * q_i          -- the int dot of ggml_vec_dot_q4_{0,1}_q8_0 (:2368, :2561)
//
// K = 32 of this instruction is exactly ONE quant block, so a 32x32 tile of exact block dots costs 8 passes of the
// matrix pipe -- half of what four v_mfma_i32_16x16x32_i8 cost (scripts/ubench/coexec3.hip, profiles/r02_ubench.txt:
// 15.7 ns against 29 ns per SIMD).  The arithmetic per output is the one of gemm_q4_mfma.hip, in the same order
// (acc = fma(float(isum_b), d_w*d_x, acc) for b = 0, 1, ...): the two kernels return the same bits.
//
// What the ubench says about gfx950 and what follows from it:
//   * matrix pipe and VALU of one SIMD do not overlap (2 x (mfma32 + 32 v_add) = the sum of both, at any occupancy):
//     time ~ MFMA passes + 2 cycles per VALU op.  Per 32x32 tile and block: 32 cycles i8 MFMA, 28 cycles of the
//     d_w x d_x outer-product MFMA (v_mfma_f32_32x32x1_2b: two tiles in 16 passes), 32 VALU ops (16 magic subtracts =
//     int -> float, 16 FMAs).  An f16 exact-integer variant (no subtract, twice the MFMA passes, 14 unpack ops per
//     fragment) measures the same sum -- so the operand formats stay QW16 / QA16 as they are.
//   * what the 16x16 kernel loses on top of its instruction mix (25-60 %) is synchronisation and LDS operand
//     traffic.  Here the WEIGHTS do not go through LDS at all: a wave owns 32 rows and streams its A fragments
//     (8 bytes per lane and block) and scales straight from L2/HBM into a register ring two K-steps deep with
//     bounds-checked buffer loads (scalar offsets: no address VALU).  Only the activations, which every wave of
//     the workgroup needs, are staged in LDS by global_load_lds; small 4-wave workgroups (two or three per CU)
//     keep the SIMDs busy while one of them sits at its K-step barrier.
//
// Fragment layouts (lane = 32 h + i):  A: row i, bytes 16h..16h+15 of the block;  B: column i, same bytes;
// D[8 (v/4) + 4 h + v%4][i] in VGPR v.  QW16 keeps the 8-byte half h of row r at slot h ^ (r>>3), QA16 the 16-byte
// half of column c at slot h ^ (c>>3): one global dwordx2 / one conflict-free ds_read_b128 per fragment.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>
#include "q4_device.h"
#include "q4_kernels.h"
#include "gemm_epi.h"

namespace fl {

typedef __attribute__((address_space(3))) void lds_void32;
typedef const __attribute__((address_space(1))) void glb_void32;
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));


#if defined(__HIP_DEVICE_COMPILE__)
#define FL_NOPK32 __attribute__((target("no-packed-fp32-ops")))
#else
#define FL_NOPK32
#endif

// WM waves stacked along M, each owning 32 rows x (32 RN) columns; the workgroup tile is (32 WM) x (32 RN).
template <int TYPE, int WM, int RN>
struct G32 {
    static constexpr int KS = 4, NSTAGE = 3, NW = WM, RING = 8;
    static constexpr int NG = 2 * RN;                                // 16-column groups per tile
    static constexpr int BLKB = 512;                                 // activation bytes of one (column group, block)
    static constexpr int BLKA = 256;                                 // weight bytes of one (row group, block)
    static constexpr int B_BYTES = NG * KS * BLKB;                   // activations of one K-step
    static constexpr int B_PIECES = B_BYTES / 1024;                  // 1-KiB global_load_lds pieces
    static constexpr int N_PLANES = TYPE == FL_TYPE_Q4_1 ? 2 : 1;    // d_x (, s_x), one padded piece each
    static constexpr int PIECES = B_PIECES + N_PLANES;
    static constexpr int LPW = (PIECES + NW - 1) / NW;               // pieces issued by EVERY wave per stage
    static constexpr int OFF_PL = B_BYTES;
    static constexpr int STAGE = OFF_PL + N_PLANES * 1024;
    static constexpr int OFF_SINK = NSTAGE * STAGE;
    static constexpr int ACT_BYTES = 16 * WM * 32 * RN * 4;          // f32 tile of the silu epilogue (reuses the ring)
    static constexpr int LDS_BYTES = OFF_SINK + 1024 > ACT_BYTES ? OFF_SINK + 1024 : ACT_BYTES;
    static constexpr int ND = TYPE == FL_TYPE_Q4_1 ? 2 : 1;          // scale planes on the weight side (d_w (, m_w))
    // buffer loads a wave issues per block: RN = 2: A + scales; RN = 1: A, and the scales of a block PAIR with the even block
    static constexpr int L_EVEN = 1 + ND, L_ODD = RN == 2 ? 1 + ND : 1;
    // VMEM operations issued after the loads of the first block of K-step t+1 when the boundary into t+1 is reached:
    // blocks +1, +2 | fill | +3, +4, +5, +6   (see the loop)
    static constexpr int VM_AT_BOUNDARY = LPW + 3 * L_EVEN + 3 * L_ODD;
};

#ifdef G32_TIMING   // development build only: per-workgroup clocks of the launch phases (scripts/dev/g32_timeline.py)
__device__ long long g32_dbg[4096 * 8];
#define G32_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g32_dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define G32_STAMP(k) do {} while (0)
#endif
// The kernel body, for the tiles of row groups [mg_base, mg_base + mg_count) of the matrix; `bid` numbers this region's tiles.
// (A launch is one region -- gemm_q4_mfma32_kernel -- or two regions with different tile shapes -- gemm_q4_mfma32_mixed_kernel.)
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the kernels' launch stubs (and loses them if it has to instantiate
                                      // the generic lambdas below, whose bodies use gfx950 builtins)
template <int TYPE, int WM, int RN, int MINW, bool PDB>
__device__ __forceinline__ FL_NOPK32 void gemm32_body(
    const uint32_t *qs, const float *dW, const float *mW,   // (no __restrict__: the ring loads must stay where they are issued)
    const int8_t *__restrict__ xq, const float *__restrict__ xd, const float *__restrict__ xs, int N, int M,
    int MGT /* row groups total */, int NGT /* col groups total */, int KB, float *__restrict__ y, int ldy,
    const float *__restrict__ resid, int ldr, const GemmSiluEpi &epi, int bid, int mg_base, int mg_count) {
    G32_STAMP(0);
#ifdef G32_TIMING
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 128 && blockIdx.x < 4096) g32_dbg[blockIdx.x * 8 + 5 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((15 << 11) | 4);
#endif
    using C = G32<TYPE, WM, RN>;
    constexpr int KS = C::KS;
    constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid_k = threadIdx.x, lane_k = tid_k & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_k >> 6);
    const int i31_k = lane_k & 31, h_k = lane_k >> 5, c15_k = lane_k & 15, g1 = (lane_k >> 4) & 1;

    // ---- XCD-aware bijective remap of the tile id: block b runs on XCD b % 8; the N-tiles that share a W row panel
    //      get consecutive ids on one XCD, so the panel is fetched from HBM once and re-read from that XCD's L2.
    const int tiles_m = (mg_count + 2 * WM - 1) / (2 * WM), tiles_n = (NGT + C::NG - 1) / C::NG;
    {
        const int nwg = tiles_m * tiles_n;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int mg0 = mg_base + tm * 2 * WM, ng0 = tn * C::NG;

    // ---- LDS fill plan of the activation side: piece ids [0, B_PIECES) int8, then d_x (, s_x).  Wave w owns pieces w, w+NW, ...;
    //      every wave issues exactly LPW LDS-DMA loads per stage (missing ones read out of range into a sink) so that one
    //      counted s_waitcnt vmcnt is valid for all waves.  MUBUF loads with a descriptor cut at NGT column groups: groups
    //      past the batch and blocks past K (offset forced out of range) arrive as zeros.
    //      The DMA loads are issued from inline asm.  hipcc cannot tell the stages of the ring apart: with an LDS-DMA it
    //      knows of in flight it makes the next ds_read (builtin MUBUF form) or the next use of ANY loaded register
    //      (global_load_lds, a FLAT operation for it) wait for that DMA -- vmcnt(0) once per K-step, which drains both the
    //      stage ring and the weight ring.  An asm DMA has no register destination (nothing the compiler could copy too
    //      early); hipcc's own counted waits for the weight ring ignore it and become a few operations conservative.
    v4i frsrc[C::LPW];
    int funit[C::LPW], lds_off[C::LPW];
    uint32_t fvoff[C::LPW];      // byte offset (a multiple of 16) | block of the K-step the lane's 16 bytes belong to (bits 0..1)
#pragma unroll
    for (int s = 0; s < C::LPW; ++s) {
        const int p = wave + s * C::NW;                    // (wave-uniform)
        const void *fbase = xq;
        int fbytes = 0;
        funit[s] = 0; lds_off[s] = C::OFF_SINK; fvoff[s] = 0x80000000u;
        if (p < C::B_PIECES) {
            constexpr int U = C::BLKB / 16;                  // 16-byte units of one (group, block)
            const int c = p * 64 + lane_k, gi = c / (KS * U), e = c % (KS * U);
            fbytes = NGT * KB * C::BLKB; funit[s] = C::BLKB;
            lds_off[s] = p * 1024;
            fvoff[s] = ((uint32_t)(ng0 + gi) * (uint32_t)KB * (uint32_t)C::BLKB + (uint32_t)e * 16u) | (uint32_t)(e / U);      // group >= NGT: out of range
        } else if (p < C::PIECES) {
            const int pl = p - C::B_PIECES;
            const int gi = lane_k / (KS * 4), e = lane_k % (KS * 4);
            fbase = pl == 0 ? xd : xs; fbytes = NGT * KB * 64; funit[s] = 64;
            lds_off[s] = C::OFF_PL + pl * 1024;
            if (gi < C::NG) fvoff[s] = ((uint32_t)(ng0 + gi) * (uint32_t)KB * 64u + (uint32_t)e * 16u) | (uint32_t)(e / 4);
        }
        // buffer descriptor (raw, stride 0) in SGPRs: everything in it is wave-uniform
        const uint64_t bp = (uint64_t)(uintptr_t)fbase;
        frsrc[s] = v4i{__builtin_amdgcn_readfirstlane((int)(uint32_t)bp), __builtin_amdgcn_readfirstlane((int)((bp >> 32) & 0xFFFF)),
                       __builtin_amdgcn_readfirstlane(fbytes), 0x00020000};
        funit[s] = __builtin_amdgcn_readfirstlane(funit[s]);
        lds_off[s] = __builtin_amdgcn_readfirstlane(lds_off[s]);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;    // LDS byte address of the ring
    auto fill = [&](int st, int kb0) FL_NOPK32 __attribute__((always_inline)) {
        const bool tail = kb0 + KS > KB;   // blocks >= KB must read zeros (K tail / past the end)
#pragma unroll
        for (int s = 0; s < C::LPW; ++s) {
            uint32_t vo = fvoff[s] & ~3u;
            if (tail && kb0 + (int)(fvoff[s] & 3u) >= KB) vo = 0x80000000u;
            const uint32_t dst = lds0 + (lds_off[s] == C::OFF_SINK ? C::OFF_SINK : st * C::STAGE + lds_off[s]);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(vo), "s"(frsrc[s]), "s"(kb0 * funit[s]) : "memory", "m0");
        }
    };

    // ---- weight side: bounds-checked buffer loads (rows past M16 and bytes past the tensor read as zero) ----
    const uint32_t wbytes = (uint32_t)MGT * (uint32_t)KB * 256u;
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(qs), 0, (int)wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dW), 0, (int)(wbytes >> 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q41 ? mW : dW), 0, (int)(wbytes >> 2), 0x00020000);
    const uint32_t rgrp = (uint32_t)(mg0 + 2 * wave + g1);
    // the 8-byte half h of the row's nibbles
    const uint32_t voffA = (rgrp * (uint32_t)KB * 16u + (uint32_t)c15_k) * 16u + (uint32_t)((h_k ^ (c15_k >> 3)) << 3);
    const uint32_t voffD = (rgrp * (uint32_t)KB * 16u + (uint32_t)c15_k) * 4u;

    v2u araw[C::RING];
    float saw[C::RING], maw[C::RING];
    // block kb of this wave's rows into ring slot `slot` (compile-time).  Blocks past the end re-read the last one: its
    // scales are finite, and the activation side is zero there.
    auto load_w = [&](auto SLOT, int kb) FL_NOPK32 __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
        const int kba = kb < KB ? kb : KB - 1;
        araw[slot] = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rA, voffA, kba * 256, 0));
        if (RN == 2) {
            saw[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, voffD, kba * 64, 0));
            if (Q41) maw[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voffD, kba * 64, 0));
        } else if ((slot & 1) == 0) {
            // scales of the block PAIR (kb, kb+1): lanes h_k = 1 read block kb + 1 (KB may be odd under tensor parallelism)
            const int kb1 = kb + 1 < KB ? kb + 1 : KB - 1;
            const uint32_t vo = voffD + (uint32_t)(h_k ? kb1 : kba) * 64u;
            saw[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, vo, 0, 0));
            if (Q41) maw[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, vo, 0, 0));
        }
    };

    // ---- per-lane_k LDS offsets of the B fragments and the activation scales ----
    const int b_off = (g1 * KS) * C::BLKB + c15_k * 32 + ((h_k ^ (c15_k >> 3)) << 4);           // + jt * 2 KS BLKB + u * BLKB
    const int sb_off = C::OFF_PL + (RN == 2 ? ((2 * h_k + g1) * KS) * 64 : (g1 * KS + h_k) * 64) + c15_k * 4;   // + u * 64

    v16f acc[RN];
    v32f ms2;        // Q4_1, RN = 2: m_w x s_x of both column tiles
    v16f ms1;        // Q4_1, RN = 1
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[j][e] = 0.f;
        ms1[e] = 0.f;
        ms2[e] = 0.f;
        ms2[16 + e] = 0.f;
    }
    v16i magic;
#pragma unroll
    for (int e = 0; e < 16; ++e) magic[e] = 0x4B400000;   // 12582912.0f = 1.5 * 2^23: D reinterpreted as f32 is magic + isum
    float negmagic = -12582912.0f;
    asm volatile("" : "+v"(negmagic));
    v32f zero32;
#pragma unroll
    for (int e = 0; e < 32; ++e) zero32[e] = 0.f;

    v4i afr[2], bfr[2][RN];
    float sbw[2] = {0.f, 0.f}, mbw[2] = {0.f, 0.f};
    v32f P[2];
    v16i D[2];
    // one 32x32 tile of exact block dots into D[dst] (as magic + isum)
    auto tile_dot = [&](auto DST, auto BUF, auto JJ) FL_NOPK32 __attribute__((always_inline)) {
        constexpr int dst = decltype(DST)::value, buf = decltype(BUF)::value, j = decltype(JJ)::value;
        D[dst] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[buf], bfr[buf][j], magic, 0, 0, 0);
    };

    auto read_b = [&](auto BUF, const unsigned char *base, auto UU) FL_NOPK32 __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, u = decltype(UU)::value;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const unsigned char *bp = base + b_off + j * (2 * KS * C::BLKB) + u * C::BLKB;
            bfr[buf][j] = *reinterpret_cast<const v4i *>(bp);
        }
        if (RN == 2) {
            sbw[buf] = *reinterpret_cast<const float *>(base + sb_off + u * 64);
            if (Q41) mbw[buf] = *reinterpret_cast<const float *>(base + sb_off + 1024 + u * 64);
        } else if ((u & 1) == 0) {                         // scales of the block pair (u, u + 1)
            sbw[(u >> 1) & 1] = *reinterpret_cast<const float *>(base + sb_off + u * 64);
            if (Q41) mbw[(u >> 1) & 1] = *reinterpret_cast<const float *>(base + sb_off + 1024 + u * 64);
        }
    };
    auto unpack = [&](auto BUF, auto SLOT) FL_NOPK32 __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value, slot = decltype(SLOT)::value;
        uint32_t l0, h0, l1, h1;
        unpack_nibbles<TYPE>(araw[slot].x, l0, h0);
        unpack_nibbles<TYPE>(araw[slot].y, l1, h1);
        afr[buf] = v4i{(int)l0, (int)h0, (int)l1, (int)h1};
    };
    // acc += float(isum) * (d_w * d_x): 16 magic subtracts (exact int -> float) + 16 FMAs (ggml.c:2452, :2478)
    auto scale_acc = [&](v16f &a, const v16i &d, const v32f &p, auto HALF) FL_NOPK32 __attribute__((always_inline)) {
        constexpr int half = decltype(HALF)::value;
        const v16f df = __builtin_bit_cast(v16f, d);     // (bit_cast of a single vector ELEMENT lvalue reads element 0: clang bug)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float f = df[e] + negmagic;
            a[e] = __builtin_fmaf(f, p[half * 16 + e], a[e]);
        }
    };
#define IC(x) std::integral_constant<int, (x)>{}

    const int nsteps = ((KB + 2 * KS - 1) / (2 * KS)) * 2;   // whole trips of the 8-block loop body (blocks past KB are zero)
    // ---- prologue: stages 0..2 and the weight ring (blocks 0..7) in flight, block 0 staged in registers ----
    fill(0, 0);
    load_w(IC(0), 0); load_w(IC(1), 1); load_w(IC(2), 2); load_w(IC(3), 3);
    fill(1, KS);
    load_w(IC(4), 4); load_w(IC(5), 5); load_w(IC(6), 6); load_w(IC(7), 7);
    G32_STAMP(1);
#ifdef FL_G32_SAFE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::LPW + 2 * C::L_EVEN + 2 * C::L_ODD) : "memory");   // stage 0 and blocks 0..3 landed
#endif
    __builtin_amdgcn_s_barrier();
    G32_STAMP(2);
    fill(2, 2 * KS);
    int cur = 0;
    const unsigned char *base = smem;
    read_b(IC(0), base, IC(0));
    unpack(IC(0), IC(0));
    if (RN == 2) {
        P[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(saw[0], sbw[0], zero32, 0, 0, 0);
        if (Q41) ms2 = __builtin_amdgcn_mfma_f32_32x32x1f32(maw[0], mbw[0], ms2, 0, 0, 0);
    } else {
        P[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(saw[0], sbw[0], zero32, 0, 0, 0);
        if (Q41) ms1 = __builtin_amdgcn_mfma_f32_32x32x2f32(maw[0], mbw[0], ms1, 0, 0, 0);
    }
    tile_dot(IC(0), IC(0), IC(0));
    __builtin_amdgcn_sched_barrier(0);

    // ---- main loop: 8 blocks (two K-steps) per trip; S = ring slot of the block, u = S % 4 its place in the K-step ----
    // RN = 2, block b:   D1(b) | operands of b+1 (LDS reads, unpack, ring reload b+8) | E(D0(b)) | P(b+1) D0(b+1) | E(D1(b))
    // RN = 1, block b:   operands of b+1 | [P(pair b+1)] D(b+1) | E(D(b))
    // At u = 3 the operands of b+1 sit in the next stage: own pieces landed (counted vmcnt), everyone's did and everyone is
    // done with this stage (barrier), the stage is refilled with K-step t+3, and the wave moves on with MFMAs in flight.
    auto block = [&](auto SS, int kb, int t) FL_NOPK32 __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value, u = S & 3, nb = (S + 1) & 1, cb = S & 1, ns = (S + 1) & 7, nu = (u + 1) & 3;
        if (RN == 2) tile_dot(IC(1), IC(cb), IC(RN - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (u == 3) {
#ifdef FL_G32_SAFE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::VM_AT_BOUNDARY) : "memory");
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fill(cur, (t + 3) * KS);                       // K-step t+3 into the stage K-step t occupied
            cur = cur + 1 == C::NSTAGE ? 0 : cur + 1;
            base = smem + cur * C::STAGE;
        }
        read_b(IC(nb), base, IC(nu));
        unpack(IC(nb), IC(ns));
        load_w(SS, kb + C::RING);
        __builtin_amdgcn_sched_barrier(0);
        if (RN == 2 && PDB) {
            scale_acc(acc[0], D[0], P[cb], IC(0));
            __builtin_amdgcn_sched_barrier(0);
            P[nb] = __builtin_amdgcn_mfma_f32_32x32x1f32(saw[ns], sbw[nb], zero32, 0, 0, 0);
            if (Q41) ms2 = __builtin_amdgcn_mfma_f32_32x32x1f32(maw[ns], mbw[nb], ms2, 0, 0, 0);
            tile_dot(IC(0), IC(nb), IC(0));
            __builtin_amdgcn_sched_barrier(0);
            scale_acc(acc[RN - 1], D[1], P[cb], IC(1));
        } else if (RN == 2) {
            // one P buffer (32 VGPRs less: a third wave per SIMD): the scale product of block b+1 is issued after the last
            // use of block b's, and has the operand reads of the next block to complete under
            scale_acc(acc[0], D[0], P[0], IC(0));
            __builtin_amdgcn_sched_barrier(0);
            tile_dot(IC(0), IC(nb), IC(0));
            __builtin_amdgcn_sched_barrier(0);
            scale_acc(acc[RN - 1], D[1], P[0], IC(1));
            __builtin_amdgcn_sched_barrier(0);
            P[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(saw[ns], sbw[nb], zero32, 0, 0, 0);
            if (Q41) ms2 = __builtin_amdgcn_mfma_f32_32x32x1f32(maw[ns], mbw[nb], ms2, 0, 0, 0);
        } else {
            if ((ns & 1) == 0) {                           // b+1 opens a block pair
                constexpr int pb = (ns >> 1) & 1;
                P[pb] = __builtin_amdgcn_mfma_f32_32x32x1f32(saw[ns], sbw[(nu >> 1) & 1], zero32, 0, 0, 0);
                if (Q41) ms1 = __builtin_amdgcn_mfma_f32_32x32x2f32(maw[ns], mbw[(nu >> 1) & 1], ms1, 0, 0, 0);
            }
            tile_dot(IC(nb), IC(nb), IC(0));
            __builtin_amdgcn_sched_barrier(0);
            if (cb == 0) scale_acc(acc[0], D[cb], P[(S >> 1) & 1], IC(0));
            else scale_acc(acc[0], D[cb], P[(S >> 1) & 1], IC(1));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nsteps; t += 2) {
        const int kb = t * KS;
        block(IC(0), kb + 0, t); block(IC(1), kb + 1, t); block(IC(2), kb + 2, t); block(IC(3), kb + 3, t);
        block(IC(4), kb + 4, t + 1); block(IC(5), kb + 5, t + 1); block(IC(6), kb + 6, t + 1); block(IC(7), kb + 7, t + 1);
    }
    G32_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the (zero / repeated) tail fills and ring loads
#undef IC

    // ---- results: lane (i31, h) holds rows 8 g + 4 h + {0..3} (g = 0..3) of column i31 of each of its RN tiles ----
    // (the lane coordinates are taken afresh: as values that live across the loop they are what the allocator spills when two
    //  bodies share a kernel -- and a kernel that touches scratch at all starts its waves slower)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int i31 = lane_e & 31, h = lane_e >> 5, c15 = lane_e & 15, lane = lane_e, tid = lane_e + 64 * wave;
    (void)c15; (void)lane; (void)tid;
    const int rowbase = (mg0 + 2 * wave) * 16;
    auto out4 = [&](int j, int g) FL_NOPK32 __attribute__((always_inline)) -> v4f {
        v4f o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = acc[j][4 * g + e];
            if (Q41) o[e] += RN == 2 ? ms2[16 * j + 4 * g + e] : ms1[4 * g + e];   // + sum_b m_w*s_x (ggml.c:2651)
        }
        return o;
    };
    if (epi.silu_tab) {
        // ---- silu(w1 x) * (w3 x) -> Q8_0 (QA16).  Rows 0..15 of a wave's 32 are w1 of 16 features, rows 16..31 w3 of the same.
        constexpr int ACT_LD = 32 * RN, NFEAT = 16 * WM;
        static_assert(WM % 2 == 0 && NFEAT * ACT_LD * 4 <= C::LDS_BYTES, "activation tile must fit the operand ring");
        float *act = reinterpret_cast<float *>(smem);           // [NFEAT][ACT_LD] f32
        __syncthreads();                                         // every wave is done with the operand ring
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const v4f a1 = out4(j, g), a3 = out4(j, g + 2);
                const int fl0 = wave * 16 + 8 * g + 4 * h, nl = j * 32 + i31;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t hx = __half_as_ushort(__float2half_rn(a1[e]));             // GGML_FP32_TO_FP16
                    const float sl = __half2float(__ushort_as_half(epi.silu_tab[hx]));        // table_silu_f16
                    act[(fl0 + e) * ACT_LD + nl] = __fmul_rn(sl, a3[e]);                      // ggml_mul(silu, tmp)
                }
            }
        __syncthreads();
        // one thread = one (token, 32-feature block): quantize_row_q8_0 arithmetic (lib/ggml.c:1341-1403, 1433-1440)
        for (int u = tid; u < (NFEAT / 32) * ACT_LD; u += 64 * WM) {
            const int fb = u / ACT_LD, nl = u % ACT_LD;
            const int n = ng0 * 16 + nl, gfb = (mg0 >> 2) + fb;
            if (n >= NGT * 16 || gfb >= epi.KBo) continue;
            float v[32];
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                v[e] = act[(fb * 32 + e) * ACT_LD + nl];
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
            const int64_t cb = ((int64_t)(n >> 4) * epi.KBo + gfb) * 16 + c;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                auto pk = [](int a, int b, int cc, int d) -> uint32_t {
                    return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(cc & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
                };
                const uint2 w2 = make_uint2(pk(qi[8 * g], qi[8 * g + 2], qi[8 * g + 4], qi[8 * g + 6]),
                                            pk(qi[8 * g + 1], qi[8 * g + 3], qi[8 * g + 5], qi[8 * g + 7]));
                *reinterpret_cast<uint2 *>(epi.oq + cb * 32 + qw16_pos(c, g) * 8) = w2;
            }
            epi.od[cb] = dd;
            epi.os[cb] = __fmul_rn(dd, (float)sum);
        }
        return;
    }
    if (epi.rope_tab) {
        // lane holds features row0..row0+3 (two rope pairs) of token n
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int n = ng0 * 16 + j * 32 + i31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row0 = rowbase + 8 * g + 4 * h;
                const v4f o = out4(j, g);
                if (n >= N || row0 >= M) continue;
                const int part = row0 / epi.El, f = row0 - part * epi.El, pos = epi.n_past + n;
                if (part < 2) {
                    const float2 *cs = epi.rope_tab + (int64_t)pos * (epi.D >> 1) + ((f % epi.D) >> 1);
                    const float2 c0 = cs[0], c1 = cs[1];
                    v4f q;
                    q[0] = __builtin_fmaf(o[0], c0.x, -(o[1] * c0.y));
                    q[1] = __builtin_fmaf(o[0], c0.y, o[1] * c0.x);
                    q[2] = __builtin_fmaf(o[2], c1.x, -(o[3] * c1.y));
                    q[3] = __builtin_fmaf(o[2], c1.y, o[3] * c1.x);
                    float *dst = part == 0 ? y + (int64_t)n * ldy + row0 : epi.kc + (int64_t)pos * epi.El + f;
                    *reinterpret_cast<v4f *>(dst) = q;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) epi.vc[(int64_t)(f + e) * epi.n_ctx + pos] = o[e];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = ng0 * 16 + j * 32 + i31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row0 = rowbase + 8 * g + 4 * h;
            v4f o = out4(j, g);
            if (n < N && row0 < M) {
                float *p = y + (int64_t)n * ldy + row0;
                const float *pr = resid ? resid + (int64_t)n * ldr + row0 : nullptr;
                if (row0 + 3 < M) {
                    if (pr) o += *reinterpret_cast<const v4f *>(pr);   // ggml_add(cur, inp) fused into the store
                    *reinterpret_cast<v4f *>(p) = o;     // (nontemporal stores here: 5-10 % slower, profiles/r02_gemm32_notes.md)
                } else {
                    for (int r = 0; r < 4 && row0 + r < M; ++r) p[r] = o[r] + (pr ? pr[r] : 0.f);
                }
            }
        }
    }
    G32_STAMP(4);
}
#endif   // __HIP_DEVICE_COMPILE__

template <int TYPE, int WM, int RN, int MINW, bool PDB>
__global__ __launch_bounds__(64 * WM, TYPE == FL_TYPE_Q4_1 ? 2 : MINW) FL_NOPK32 void gemm_q4_mfma32_kernel(
    const uint32_t *qs, const float *dW, const float *mW, const int8_t *__restrict__ xq, const float *__restrict__ xd,
    const float *__restrict__ xs, int N, int M, int MGT, int NGT, int KB, float *__restrict__ y, int ldy,
    const float *__restrict__ resid, int ldr, GemmSiluEpi epi) {
#if defined(__HIP_DEVICE_COMPILE__)
    gemm32_body<TYPE, WM, RN, MINW, PDB>(qs, dW, mW, xq, xd, xs, N, M, MGT, NGT, KB, y, ldy, resid, ldr, epi, (int)blockIdx.x, 0, MGT);
#endif
}

// Two tile shapes in one launch.  A launch's duration is set by the SIMD that draws the most wave tiles: with 128 x 64 tiles,
// LLaMA-7B's w1|w3 matmul at n_batch 512 is 1376 workgroups = 5.375 per CU -- six rounds' worth of time for 5.4 rounds of work
// (M scan, profiles/r02_gemm32_mscan.txt).  The first n_a workgroups take 128 x 64 tiles of row groups [0, mg_split) -- whole
// multiples of 256 workgroups -- and the rest covers the remaining row groups with 128 x 32 tiles: twice as many workgroups of
// half the work each, which fill the slots the last full round frees.  Same arithmetic per output in both regions (every tile
// configuration returns identical bits), so the result does not depend on where the split falls.
template <int TYPE>
__global__ __launch_bounds__(256, TYPE == FL_TYPE_Q4_1 ? 2 : 3) FL_NOPK32 void gemm_q4_mfma32_mixed_kernel(
    const uint32_t *qs, const float *dW, const float *mW, const int8_t *__restrict__ xq, const float *__restrict__ xd,
    const float *__restrict__ xs, int N, int M, int MGT, int NGT, int KB, float *__restrict__ y, int ldy,
    const float *__restrict__ resid, int ldr, GemmSiluEpi epi, int n_a, int mg_split) {
#if defined(__HIP_DEVICE_COMPILE__)
    if ((int)blockIdx.x < n_a)
        gemm32_body<TYPE, 4, 2, 3, false>(qs, dW, mW, xq, xd, xs, N, M, MGT, NGT, KB, y, ldy, resid, ldr, epi, (int)blockIdx.x, 0, mg_split);
    else
        gemm32_body<TYPE, 4, 1, 3, true>(qs, dW, mW, xq, xd, xs, N, M, MGT, NGT, KB, y, ldy, resid, ldr, epi, (int)blockIdx.x - n_a, mg_split,
                                         MGT - mg_split);
#endif
}

// ------------------------------------------------------------------------------------------------
// configurations (ids 100...) and launch
// ------------------------------------------------------------------------------------------------
//                        id  WM RN minwaves/SIMD  P double-buffered     tile      waves
#define FL_GEMM32_CONFIGS(X)                                                           \
    X(100, 4, 2, 2, true)   /* 128 x 64   4 waves of 32x64                          */ \
    X(101, 4, 1, 3, true)   /* 128 x 32   4 waves of 32x32                          */ \
    X(102, 2, 2, 2, true)   /*  64 x 64   2 waves of 32x64                          */ \
    X(103, 8, 2, 2, true)   /* 256 x 64   8 waves of 32x64                          */ \
    X(104, 8, 1, 3, true)   /* 256 x 32   8 waves of 32x32                          */ \
    X(105, 2, 1, 3, true)   /*  64 x 32   2 waves of 32x32                          */ \
    X(106, 4, 2, 3, false)  /* 128 x 64   4 waves of 32x64, three waves per SIMD    */ \
    X(108, 8, 2, 3, false)  /* 256 x 64   8 waves of 32x64, three waves per SIMD    */

template <int TYPE, int WM, int RN, int MINW, bool PDB>
static hipError_t launch_gemm32(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                                const float *resid, int ldr, const GemmSiluEpi &epi) {
    using C = G32<TYPE, WM, RN>;
    const int MGT = W.M16 / 16, NGT = fl_roundup(N, 16) / 16;
    const int tiles = ((MGT + 2 * WM - 1) / (2 * WM)) * ((NGT + C::NG - 1) / C::NG);
    static_assert(C::LDS_BYTES <= 65536, "no dynamic-LDS attribute needed");
    hipLaunchKernelGGL((gemm_q4_mfma32_kernel<TYPE, WM, RN, MINW, PDB>), dim3(tiles), dim3(64 * WM), C::LDS_BYTES, st,
                       W.qs, W.d, W.m, xq.q,
                       xq.d, xq.s, N, W.M, MGT, NGT, W.KB, y, ldy, resid, ldr, epi);
    return hipGetLastError();
}

// cfg 116: 128 x 64 tiles for the row groups that fill whole rounds of 256 workgroups, 128 x 32 tiles for the rest.
// Row groups [0, mg_split) -> n_a workgroups of 128 x 64; [mg_split, MGT) -> n_b workgroups of 128 x 32.
void gemm32_mixed_split(int MGT, int NGT, int *n_a, int *mg_split, int *n_b) {
    const int tn_a = (NGT + 3) / 4, tn_b = (NGT + 1) / 2;              // column tiles of 64 / 32
    const int tm_all = (MGT + 7) / 8;                                  // 128-row tiles
    int tm_a = (int)((int64_t)tm_all * tn_a / 256 * 256 / tn_a);       // row tiles of whole 256-workgroup rounds ...
    while (tm_a > 0 && (tm_a * tn_a) % 8 != 0) --tm_a;                 // ... and a multiple of the 8 XCDs (the remap of region B)
    *mg_split = tm_a * 8 < MGT ? tm_a * 8 : MGT;
    *n_a = tm_a * tn_a;
    *n_b = ((MGT - *mg_split + 7) / 8) * tn_b;
}

template <int TYPE>
static hipError_t launch_gemm32_mixed(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                                      const float *resid, int ldr, const GemmSiluEpi &epi) {
    using CA = G32<TYPE, 4, 2>;
    using CB = G32<TYPE, 4, 1>;
    const int MGT = W.M16 / 16, NGT = fl_roundup(N, 16) / 16;
    int n_a, mg_split, n_b;
    gemm32_mixed_split(MGT, NGT, &n_a, &mg_split, &n_b);
    constexpr int lds = CA::LDS_BYTES > CB::LDS_BYTES ? CA::LDS_BYTES : CB::LDS_BYTES;
    static_assert(lds <= 65536, "no dynamic-LDS attribute needed");
    hipLaunchKernelGGL((gemm_q4_mfma32_mixed_kernel<TYPE>), dim3(n_a + n_b), dim3(256), lds, st,
                       W.qs, W.d, W.m, xq.q,
                       xq.d, xq.s, N, W.M, MGT, NGT, W.KB, y, ldy, resid, ldr, epi, n_a, mg_split);
    return hipGetLastError();
}

bool gemm32_supports(const fl_qtensor &W, int cfg, bool silu) {
    if ((uint64_t)(W.M16 / 16 + 16) * (uint64_t)W.KB * 256u >= (1ull << 31)) return false;   // 32-bit buffer offsets
    (void)cfg; (void)silu;
    return W.KB >= 1;
}

hipError_t gemm32_launch(int cfg, const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                         const float *resid, int ldr, const GemmSiluEpi &epi) {
#define X(ID, WM, RN, MINW, PDB)                                                                                          \
    if (cfg == ID)                                                                                                        \
        return W.type == FL_TYPE_Q4_0 ? launch_gemm32<FL_TYPE_Q4_0, WM, RN, MINW, PDB>(W, xq, N, y, ldy, st, resid, ldr, epi) \
                                      : launch_gemm32<FL_TYPE_Q4_1, WM, RN, MINW, PDB>(W, xq, N, y, ldy, st, resid, ldr, epi);
    FL_GEMM32_CONFIGS(X)
#undef X
    if (cfg == 116)
        return W.type == FL_TYPE_Q4_0 ? launch_gemm32_mixed<FL_TYPE_Q4_0>(W, xq, N, y, ldy, st, resid, ldr, epi)
                                      : launch_gemm32_mixed<FL_TYPE_Q4_1>(W, xq, N, y, ldy, st, resid, ldr, epi);
    return hipErrorInvalidValue;
}

}  // namespace fl
#ifdef G32_TIMING
extern "C" __attribute__((visibility("default"))) int fl_debug_g32_timing(long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fl::g32_dbg), sizeof(long long) * (size_t)n * 8); }
#endif
