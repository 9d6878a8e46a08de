// gemm_q4_mfma.hip -- prefill path (N >= 9) of ggml_compute_forward_mul_mat_q_f32
// (/root/reference/lib/ggml.c:7928-8176, COMPUTE phase :8127-8163): exact-integer MFMA GEMM for gfx950.
//
//   y[n][m] = sum_b  (d_w[m,b] * d_x[n,b]) * isum[m,n,b]   (+ m_w[m,b] * s_x[n,b] for Q4_1)
//   isum[m,n,b] = sum_{i<32} w_i * q_i  -- the int dot of ggml_vec_dot_q4_{0,1}_q8_0 (:2368, :2561)
//
// One MFMA K-step is exactly one 32-element quant block: v_mfma_i32_16x16x32_i8 (A = 16 W rows,
// B = 16 activation columns) produces the 16x16 block dots exactly.  The per-block scale product cannot
// be folded into integer operands, so every 16x16x32 tile needs a scale-accumulate of its 256 block sums.
// What that costs on gfx950 was measured first (scripts/ubench/coexec*.hip, profiles/r01_gemm_ablation.txt):
//
//   * on one SIMD the i8 MFMA (16 cycles) and the VALU barely overlap: ~2 scalar VALU ops hide under one MFMA,
//     every further one costs its full ~2 cycles.  A v_pk_*_f32 does not co-issue with an executing MFMA at all
//     and has no throughput edge over two scalar ops -> the kernel is compiled without packed-f32 selection.
//   * floor for this instruction mix = MFMA + 8 scalar ops (4 magic subtracts + 4 FMAs) + the scale product
//     ~= 18.5 ns per tile per SIMD (~900 TOP/s chip-wide); the loop below runs at ~70 % of that.
//
// Design points:
//   * int -> float without v_cvt: the MFMA's C input is the constant 0x4B400000 (= 1.5*2^23 as f32
//     bits); D = magic + isum reinterpreted as f32 IS 12582912 + isum exactly (|isum| < 2^22).
//   * d_w x d_x for FOUR tiles comes from one v_mfma_f32_16x16x1 (4-block outer product, C = 0): its D layout is
//     the i8 MFMA's D layout, so acc = fma(float(isum), P, acc) needs no shuffles.  For Q4_1 the m_w x s_x term is
//     the same instruction accumulating into its own registers (no VALU at all).
//   * the MFMA of tile t+1 is issued before the epilogue of tile t (software pipeline inside a wave,
//     pinned with sched_barrier because hipcc otherwise re-serialises MFMA -> s_nop -> own epilogue).
//   * block-granular operand pipeline: while block b is on the MFMA/VALU pipes, the LDS reads of block b+1 are in
//     flight, and the stage barrier + global_load_lds issue sit in the middle of a K-step under MFMAs in flight.
//   * several workgroup shapes (template WM x WN waves of TM x TN MFMA tiles) and a shape-driven choice among
//     them (pick_config): small outputs need many small wave tiles to occupy all 1024 SIMDs.
//   * QW16 / QA16 make every LDS fill a linear 16-byte copy -> global_load_lds (no staging VGPRs) into a
//     3-deep LDS ring with counted s_waitcnt vmcnt; per-lane source pointers are computed once and
//     advanced by a constant per K-step.  Fragment reads are bank-conflict free by construction
//     (q4_layout.h; SQ_LDS_BANK_CONFLICT = 0 in profiles/).
//   * XCD-aware bijective tile order: the N-tiles that share a W row panel run on one XCD's L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "q4_device.h"
#include "q4_kernels.h"
#include "gemm_epi.h"

namespace fl {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ const uint4 fl_zero_chunk[1] = {{0u, 0u, 0u, 0u}};  // source of out-of-range LDS fills

#ifndef FL_ABL
#define FL_ABL 0
#endif

// KS = quant blocks per K-step (LDS stage).  KS = 2 runs a 3-deep LDS ring, KS = 4 a 2-deep one (about the same LDS
// bytes and the same prefetch distance in blocks): half as many barriers and fill bookkeeping per MFMA.
template <int TYPE, int WM, int WN, int TM, int TN, int KS>
struct GemmCfg {
#ifndef FL_KS4_STAGES
#define FL_KS4_STAGES 2
#endif
    static constexpr int NSTAGE = KS == 2 ? 3 : FL_KS4_STAGES;   // LDS ring depth
    static constexpr int NW = WM * WN;                        // waves per workgroup
    static constexpr int MG = WM * TM;                        // W row groups (16 rows) per workgroup tile
    static constexpr int NG = WN * TN;                        // activation column groups per workgroup tile
    static constexpr int A_BYTES = MG * KS * 256;             // packed nibbles
    static constexpr int B_BYTES = NG * KS * 512;             // int8 activations
    static constexpr int A_PIECES = A_BYTES / 1024;           // 1-KiB global_load_lds pieces
    static constexpr int B_PIECES = B_BYTES / 1024;
    static constexpr int N_PLANES = TYPE == FL_TYPE_Q4_1 ? 4 : 2;   // dW, dX (, mW, sX)
    static constexpr int PLP = ((MG > NG ? MG : NG) * KS * 64 + 1023) / 1024;   // pieces per scale plane (padded)
    static constexpr int PL_STRIDE = PLP * 1024;
    static constexpr int PIECES = A_PIECES + B_PIECES + N_PLANES * PLP;
    static constexpr int LPW = (PIECES + NW - 1) / NW;        // pieces issued by EVERY wave per stage
    static constexpr int OFF_B = A_BYTES;
    static constexpr int OFF_PL = OFF_B + B_BYTES;            // planes, PL_STRIDE apart: dW | dX | mW | sX
    static constexpr int STAGE = OFF_PL + N_PLANES * PL_STRIDE;
    static constexpr int OFF_SINK = NSTAGE * STAGE;           // 1 KiB sink for padding pieces
    static constexpr int LDS_BYTES = OFF_SINK + 1024;
    static_assert(KS == 2 || KS == 4, "K-step of 2 or 4 quant blocks");
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "tile must be made of whole 1-KiB pieces");
};

// gfx950: a v_pk_*_f32 cannot issue while an MFMA is executing (scripts/ubench/coexec.hip: 8 x (mfma + 1 v_pk_fma) takes
// 96 ns against 57 ns for 8 x (mfma + 2 v_fma)), and packed f32 has no throughput edge over two scalar ops here.
// The whole kernel is therefore compiled without packed-f32 instruction selection.
#if defined(__HIP_DEVICE_COMPILE__)
#define FL_NOPK __attribute__((target("no-packed-fp32-ops")))
#else
#define FL_NOPK   /* the host pass only needs the launch stub */
#endif

template <int TYPE, int WM, int WN, int TM, int TN, int MINW, int KS>
__global__ __launch_bounds__(64 * WM * WN, (TYPE == FL_TYPE_Q4_1 && MINW == 4) ? 3 : MINW) FL_NOPK void gemm_q4_mfma_kernel(
    const uint4 *__restrict__ qs, const float *__restrict__ dW, const float *__restrict__ mW,
    const int8_t *__restrict__ xq, const float *__restrict__ xd, const float *__restrict__ xs, int N, int M,
    int MGT /* row groups total */, int NGT /* col groups total */, int KB, float *__restrict__ y, int ldy,
    const float *__restrict__ resid, int ldr, GemmSiluEpi epi) {
    using Cfg = GemmCfg<TYPE, WM, WN, TM, TN, KS>;
    constexpr int GM_KS = KS, GM_NSTAGE = Cfg::NSTAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- XCD-aware bijective remap of the tile id (cdna guide T1): block b runs on XCD b%8 ----
    const int tiles_m = (MGT + Cfg::MG - 1) / Cfg::MG, tiles_n = (NGT + Cfg::NG - 1) / Cfg::NG;
    int bid = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int mg0 = tm * Cfg::MG, ng0 = tn * Cfg::NG;

    // ---- LDS fill plan.  Piece ids: [0, A_PIECES) A, then B, then the scale planes.  Wave w owns pieces
    //      w, w+NW, ...; every wave issues exactly LPW global_load_lds per stage (missing ones go to a sink
    //      from a zero chunk) so one counted s_waitcnt vmcnt(LPW * stages_in_flight) is valid for all waves.
    //      The per-lane source pointer of each owned piece is computed ONCE; a K-step advances it by a constant.
    const unsigned char *src[Cfg::LPW];
    int step_bytes[Cfg::LPW], lds_off[Cfg::LPW], blk_of_lane[Cfg::LPW];
    const unsigned char *const zsrc = reinterpret_cast<const unsigned char *>(fl_zero_chunk);
#pragma unroll
    for (int s = 0; s < Cfg::LPW; ++s) {
        const int p = wave + s * Cfg::NW;
        src[s] = zsrc;
        step_bytes[s] = 0;
        lds_off[s] = -1;
        blk_of_lane[s] = 0;
        if (p < Cfg::A_PIECES) {
            const int c = p * 64 + lane, gi = c / (GM_KS * 16), e = c % (GM_KS * 16);
            lds_off[s] = p * 1024;
            blk_of_lane[s] = e / 16;
            if (mg0 + gi < MGT) {
                src[s] = reinterpret_cast<const unsigned char *>(qs + ((int64_t)(mg0 + gi) * KB) * 16 + e);
                step_bytes[s] = GM_KS * 256;
            }
        } else if (p < Cfg::A_PIECES + Cfg::B_PIECES) {
            const int pb = p - Cfg::A_PIECES;
            const int c = pb * 64 + lane, gi = c / (GM_KS * 32), e = c % (GM_KS * 32);
            lds_off[s] = Cfg::OFF_B + pb * 1024;
            blk_of_lane[s] = e / 32;
            if (ng0 + gi < NGT) {
                src[s] = reinterpret_cast<const unsigned char *>(xq) + ((int64_t)(ng0 + gi) * KB) * 512 + e * 16;
                step_bytes[s] = GM_KS * 512;
            }
        } else if (p < Cfg::PIECES) {
            const int idx = p - Cfg::A_PIECES - Cfg::B_PIECES;
            const int pl = idx / Cfg::PLP, pp = idx % Cfg::PLP;    // plane 0 dW, 1 dX, 2 mW, 3 sX; piece inside the plane
            const int c = pp * 64 + lane, gi = c / (GM_KS * 4), e = c % (GM_KS * 4);
            const bool wside = (pl & 1) == 0;
            const float *plane = pl == 0 ? dW : pl == 1 ? xd : pl == 2 ? mW : xs;
            const int g = (wside ? mg0 : ng0) + gi;
            lds_off[s] = Cfg::OFF_PL + pl * Cfg::PL_STRIDE + pp * 1024;
            blk_of_lane[s] = e / 4;
            if (wside ? (gi < Cfg::MG && g < MGT) : (gi < Cfg::NG && g < NGT)) {
                src[s] = reinterpret_cast<const unsigned char *>(plane + ((int64_t)g * KB) * 16 + e * 4);
                step_bytes[s] = GM_KS * 64;
            }
        }
    }
    auto fill = [&](int st, int kb0) FL_NOPK __attribute__((always_inline)) {
        // kb0 .. kb0+KS-1 are the blocks of this stage; blocks >= KB read the zero chunk (K tail / past the end)
        const bool tail = kb0 + GM_KS > KB;
#pragma unroll
        for (int s = 0; s < Cfg::LPW; ++s) {
            const unsigned char *p = src[s];
            if (tail && kb0 + blk_of_lane[s] >= KB) p = zsrc;
            unsigned char *dst = smem + (lds_off[s] >= 0 ? st * Cfg::STAGE + lds_off[s] : Cfg::OFF_SINK);
            __builtin_amdgcn_global_load_lds((glb_void *)p, (lds_void *)dst, 16, 0, 0);
            src[s] += step_bytes[s];
        }
    };

    // Accumulators live in groups of four 16x16 tiles (v16f = the D operand of one 4-block f32 MFMA).  Tiles of one
    // K-step are numbered flat = b*TT + i*TN + j; group u = flat/4 shares one "scale MFMA", lane group k = flat%4.
    // Every output accumulates its blocks in K order whatever the tile shape: results do not depend on the configuration.
    constexpr int TT = TM * TN;
    // TT = 2 (16x32 wave tile): one scale MFMA serves the two blocks of a K-step (lane groups {b0 j0, b0 j1, b1 j0, b1 j1});
    // both blocks still accumulate into the same two accumulator slots, in K order.
    static_assert(TT == 2 || TT % 4 == 0, "wave tile must be 2 or a multiple of 4 MFMA tiles");
    static_assert(TN == 2 || TN == 4, "TN must be 2 or 4");
    constexpr int G = TT == 2 ? 1 : TT / 4;            // accumulator groups
    v16f acc[G], msacc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f, msacc[g][e] = 0.f;

    const v4i magic = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};  // 12582912.0f = 1.5 * 2^23
    v4f negmagic = {-12582912.0f, -12582912.0f, -12582912.0f, -12582912.0f};
    asm volatile("" : "+v"(negmagic));   // keep it in VGPRs: with an SGPR operand hipcc splits half of the packed adds
    const int apos = qw16_pos(l15, lg);
    // per-lane LDS byte offsets inside a stage (constant over the kernel)
    const int a_off = ((wm * TM * GM_KS) * 16 + l15) * 16 + apos * 4;               // + (i*KS + b)*256
    const int b_off = Cfg::OFF_B + ((wn * TN * GM_KS) * 16 + l15) * 32 + apos * 8;  // + (j*KS + b)*512
    // scale operands of the 4-block outer-product MFMA: lane group lg = k supplies rows (A: d_w) / columns (B: d_x)
    // of tile flat = 4u + k.  Offset = lane part (below) + compile-time part of u.
    int ki, kj, kb;                                                                  // (i, j, b) contribution of k
    if constexpr (TT == 2) { ki = 0; kj = lg & 1; kb = lg >> 1; }
    else if constexpr (TN == 4) { ki = 0; kj = lg; kb = 0; }
    else { ki = lg >> 1; kj = lg & 1; kb = 0; }
    const int sa_off = Cfg::OFF_PL + (((wm * TM + ki) * GM_KS + kb) * 16 + l15) * 4;          // d_w (m_w: +2 planes)
    const int sb_off = Cfg::OFF_PL + Cfg::PL_STRIDE + (((wn * TN + kj) * GM_KS + kb) * 16 + l15) * 4;   // d_x (s_x: +2 planes)
    // TT = 2, Q4_1: the m_w x s_x MFMA is issued per block (lane groups {j0, j1, j0, j1} of THAT block) so that its
    // accumulator sees the blocks in K order like every other configuration
    const int ma_off = Cfg::OFF_PL + 2 * Cfg::PL_STRIDE + ((wm * TM) * GM_KS * 16 + l15) * 4;
    const int mb_off = Cfg::OFF_PL + 3 * Cfg::PL_STRIDE + ((wn * TN + (lg & 1)) * GM_KS * 16 + l15) * 4;

    // ---- block-granular software pipeline (GM_KS = 2 blocks per stage) --------------------------------------------
    //   registers hold the operands of two quant blocks: the one the MFMAs are consuming and the one whose LDS reads are
    //   in flight.  Step t:   read(t, b1) | tiles of (t, b0) | wait + barrier + fill(stage t+3) + read(t+1, b0) | tiles of (t, b1)
    //   so LDS latency, the barrier and the fill issue all sit under a full block of MFMA/VALU work of the same wave.
    static_assert(TT != 2 || GM_KS == 2, "the 16x32 wave tile pairs the two blocks of a 2-block K-step");
    constexpr int UB = TT == 2 ? 1 : TT / 4;           // scale MFMAs per block (TT == 2: one per K-step, read with block 0)
    struct Ops {
        uint32_t araw[TM];                             // packed nibbles as read from LDS (unpacked right before use)
        long bq[TN];
        float sa[UB], sb[UB], ma[UB], mb[UB];
    };
    auto read_ops = [&](Ops &o, const unsigned char *base, int b) FL_NOPK __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) o.araw[i] = *reinterpret_cast<const uint32_t *>(base + a_off + (i * GM_KS + b) * 256);
#pragma unroll
        for (int j = 0; j < TN; ++j) o.bq[j] = *reinterpret_cast<const long *>(base + b_off + (j * GM_KS + b) * 512);
        if constexpr (TT == 2) {
            if (TYPE == FL_TYPE_Q4_1) {
                o.ma[0] = *reinterpret_cast<const float *>(base + ma_off + b * 64);
                o.mb[0] = *reinterpret_cast<const float *>(base + mb_off + b * 64);
            }
            if (b != 0) return;
            o.sa[0] = *reinterpret_cast<const float *>(base + sa_off);
            o.sb[0] = *reinterpret_cast<const float *>(base + sb_off);
            return;
        }
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            const int cu_i = TN == 4 ? ub : ub * 2;    // compile-time part of i for group ub (lane part is in sa_off)
            const int oa = (cu_i * GM_KS + b) * 64, ob = b * 64;
            o.sa[ub] = *reinterpret_cast<const float *>(base + sa_off + oa);
            o.sb[ub] = *reinterpret_cast<const float *>(base + sb_off + ob);
            if (TYPE == FL_TYPE_Q4_1) {
                o.ma[ub] = *reinterpret_cast<const float *>(base + sa_off + 2 * Cfg::PL_STRIDE + oa);
                o.mb[ub] = *reinterpret_cast<const float *>(base + sb_off + 2 * Cfg::PL_STRIDE + ob);
            }
        }
    };
#ifdef FL_ABL_NOMFMA   // timing experiment only: the epilogue runs on garbage, no MFMA is issued
#define FL_MFMA(a, b, c) ({ v4i r_; asm volatile("; no mfma" : "=v"(r_) : "v"(a), "v"(b), "v"(c)); r_; })
#else
#define FL_MFMA(a, b, c) __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0)
#endif
    const int nsteps = (KB + GM_KS - 1) / GM_KS;
#pragma unroll
    for (int st = 0; st < GM_NSTAGE - 1; ++st) fill(st, st * GM_KS);   // past-the-end fills are all-zero pieces
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::LPW * (GM_NSTAGE - 2)) : "memory");
    __builtin_amdgcn_s_barrier();                      // stage 0 has landed for every wave
    fill(GM_NSTAGE - 1, (GM_NSTAGE - 1) * GM_KS);
    Ops ops[2];
    read_ops(ops[0], smem, 0);
    __builtin_amdgcn_sched_barrier(0);

    constexpr int NT = GM_KS * TT;                     // tiles per K-step
    const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        const unsigned char *base = smem + cur * Cfg::STAGE;
        int nxt = cur + 1 == GM_NSTAGE ? 0 : cur + 1;
        long afrag[2][TM];                              // unpacked A fragments of the block in flight / the next one
        // MFMA stream of the step: P(0) T(0) T(1) | E(0) T(2) | E(1) T(3) | E(2) P(1) T(4) | ...  (E = VALU scale-accumulate)
        // The d_w x d_x outer products of four tiles come from ONE v_mfma_f32_16x16x1 (4 blocks): 8 passes of the matrix
        // pipe instead of four VALU multiplies per tile.  Block b lives in ops[b & 1] / afrag[b & 1].
        auto unpack_block = [&](int b) FL_NOPK __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint32_t lo, hi;
                unpack_nibbles<TYPE>(ops[b & 1].araw[i], lo, hi);
                afrag[b & 1][i] = (long)(((uint64_t)hi << 32) | lo);
            }
        };
        auto tile_mfma = [&](int flat) FL_NOPK __attribute__((always_inline)) -> v4i {
            const int b = flat / TT, i = (flat % TT) / TN, j = flat % TN;   // (TT == 2: i = 0)
            return FL_MFMA(afrag[b & 1][i], ops[b & 1].bq[j], magic);
        };
        auto scale_mfma = [&](int u, v16f &P) FL_NOPK __attribute__((always_inline)) {   // group u of the step
            const int b = TT == 2 ? 0 : u / UB, ub = TT == 2 ? 0 : u % UB;
            P = __builtin_amdgcn_mfma_f32_16x16x1f32(ops[b & 1].sa[ub], ops[b & 1].sb[ub], zero16, 0, 0, 0);
            if (TYPE == FL_TYPE_Q4_1 && TT != 2)
                msacc[u % G] = __builtin_amdgcn_mfma_f32_16x16x1f32(ops[b & 1].ma[ub], ops[b & 1].mb[ub], msacc[u % G], 0, 0, 0);
        };
        // The operands of block nb+1 are requested when block nb starts (its first MFMA is out, so the registers of block
        // nb-1 are free).  For the last block of the step the next block is block 0 of the NEXT stage: that is where the
        // stage hand-over sits -- this wave's pieces of stage t+1 landed (vmcnt), everyone's did and everyone finished
        // reading stage t (barrier), the freed buffer is refilled, and the wave moves on with MFMAs already in flight.
        auto start_of_block = [&](int nb) FL_NOPK __attribute__((always_inline)) {
            if (nb < GM_KS - 1) {
#if !(FL_ABL & 2)
                read_ops(ops[(nb + 1) & 1], base, nb + 1);
#endif
            } else {
#if !(FL_ABL & 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::LPW * (GM_NSTAGE - 2)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(FL_ABL & 4)
                __builtin_amdgcn_s_barrier();
#endif
                fill(cur, (t + GM_NSTAGE) * GM_KS);    // stage t+NSTAGE into the buffer stage t occupied
#endif
#if !(FL_ABL & 2)
                read_ops(ops[(nb + 1) & 1], smem + nxt * Cfg::STAGE, 0);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        v16f P0, P1 = zero16;
        unpack_block(0);
        scale_mfma(0, P0);
        v4i r0 = tile_mfma(0), r1 = magic;
        __builtin_amdgcn_sched_barrier(0);
        // (block 1's LDS reads go out only now: hipcc waits with lgkmcnt(0) before the first use of block 0's operands --
        // the counter state is unknown across the back edge -- and that wait must not cover these reads)
        start_of_block(0);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if (tt + 1 < NT) {
                if ((tt + 1) % TT == 0) unpack_block((tt + 1) / TT);            // first tile of the next block
                if ((tt + 1) % 4 == 0) {
                    if (((tt + 1) / 4) & 1) scale_mfma((tt + 1) / 4, P1); else scale_mfma((tt + 1) / 4, P0);
                }
                const v4i rn = tile_mfma(tt + 1);
                if (tt & 1) r0 = rn; else r1 = rn;
            }
            __builtin_amdgcn_sched_barrier(0);
            const int u = tt / 4, k = tt % 4, g = u % G;
            const int ka = TT == 2 ? (k & 1) : k;                                      // accumulator slot of the tile
            if (TT == 2 && TYPE == FL_TYPE_Q4_1 && (tt % TT) == 0)                      // per block, slots {j0, j1}
                msacc[0] = __builtin_amdgcn_mfma_f32_16x16x1f32(ops[(tt / TT) & 1].ma[0], ops[(tt / TT) & 1].mb[0], msacc[0], 0, 0, 0);
            const v4f f = __builtin_bit_cast(v4f, (tt & 1) ? r1 : r0) + negmagic;   // exact: float(isum)
            const v16f &P = (u & 1) ? P1 : P0;
            const v4f p = {P[4 * k], P[4 * k + 1], P[4 * k + 2], P[4 * k + 3]};        // d_w*d_x (ggml.c:2452)
            v4f a = {acc[g][4 * ka], acc[g][4 * ka + 1], acc[g][4 * ka + 2], acc[g][4 * ka + 3]};
            a = __builtin_elementwise_fma(f, p, a);                                     // fma(d, isum, acc) (:2478)
            acc[g][4 * ka] = a[0]; acc[g][4 * ka + 1] = a[1]; acc[g][4 * ka + 2] = a[2]; acc[g][4 * ka + 3] = a[3];
            __builtin_amdgcn_sched_barrier(0);
            if (tt % TT == TT - 1 && tt + 1 < NT) start_of_block(tt / TT + 1);   // the next block's first MFMA is already out
        }
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the (zero) tail fills before the wave ends

    if (epi.silu_tab) {
        if constexpr (TM % 2 == 0) {
            // ---- silu(w1 x) * (w3 x) -> Q8_0 (QA16).  Tiles i (even) / i+1 hold the same 16 features of w1 / w3.
            constexpr int ACT_LD = Cfg::NG * 16, NFEAT = Cfg::MG * 8;
            static_assert(NFEAT % 32 == 0 && NFEAT * ACT_LD * 4 <= Cfg::LDS_BYTES, "activation tile must fit the operand ring");
            float *act = reinterpret_cast<float *>(smem);           // [NFEAT][ACT_LD] f32
            __syncthreads();                                         // every wave is done with the operand ring
#pragma unroll
            for (int i = 0; i < TM; i += 2)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int f1 = i * TN + j, f3 = (i + 1) * TN + j;
                    const int fl0 = ((wm * TM + i) >> 1) * 16 + lg * 4, nl = (wn * TN + j) * 16 + l15;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a1 = acc[f1 / 4][4 * (f1 % 4) + e], a3 = acc[f3 / 4][4 * (f3 % 4) + e];
                        if (TYPE == FL_TYPE_Q4_1) { a1 += msacc[f1 / 4][4 * (f1 % 4) + e]; a3 += msacc[f3 / 4][4 * (f3 % 4) + e]; }
                        const uint16_t hx = __half_as_ushort(__float2half_rn(a1));            // GGML_FP32_TO_FP16
                        const float sl = __half2float(__ushort_as_half(epi.silu_tab[hx]));   // table_silu_f16
                        act[(fl0 + e) * ACT_LD + nl] = __fmul_rn(sl, a3);                    // ggml_mul(silu, tmp)
                    }
                }
            __syncthreads();
            // one thread = one (token, 32-feature block): quantize_row_q8_0 arithmetic (lib/ggml.c:1341-1403, 1433-1440)
            for (int u = tid; u < (NFEAT / 32) * ACT_LD; u += 64 * WM * WN) {
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
        }
        return;
    }
    if (epi.rope_tab) {
        // lane holds features row0..row0+3 (two rope pairs) of token n
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = (mg0 + wm * TM + i) * 16 + lg * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = (ng0 + wn * TN + j) * 16 + l15;
                const int flat = i * TN + j, g = flat / 4, k = flat % 4;
                v4f o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = acc[g][4 * k + e];
                    if (TYPE == FL_TYPE_Q4_1) o[e] += msacc[g][4 * k + e];
                }
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
    // ---- store: lane holds rows m = 16*g + 4*lg + {0..3} of column n = 16*h + l15 -> one 16-byte store
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row0 = (mg0 + wm * TM + i) * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = (ng0 + wn * TN + j) * 16 + l15;
            v4f o;
            {
                const int flat = i * TN + j, g = flat / 4, k = flat % 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = acc[g][4 * k + e];
                    if (TYPE == FL_TYPE_Q4_1) o[e] += msacc[g][4 * k + e];   // + sum_b m_w*s_x (ggml.c:2651)
                }
            }
            if (n < N && row0 < M) {
                float *p = y + (int64_t)n * ldy + row0;
                const float *pr = resid ? resid + (int64_t)n * ldr + row0 : nullptr;
                if (row0 + 3 < M) {
                    if (pr) o += *reinterpret_cast<const v4f *>(pr);   // ggml_add(cur, inp) fused into the store
                    *reinterpret_cast<v4f *>(p) = o;
                } else {
                    for (int r = 0; r < 4 && row0 + r < M; ++r) p[r] = o[r] + (pr ? pr[r] : 0.f);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// configurations and the shape-driven choice
// ------------------------------------------------------------------------------------------------
//                      WM WN TM TN  minwaves/SIMD  KS     tile      waves
#define FL_GEMM_CONFIGS(X)                                                      \
    X(0, 2, 2, 4, 4, 2, 2)  /* 128x128   4 waves of 64x64                    */ \
    X(1, 2, 2, 2, 4, 4, 2)  /*  64x128   4 waves of 32x64                    */ \
    X(2, 4, 2, 2, 4, 2, 2)  /* 128x128   8 waves of 32x64                    */ \
    X(3, 4, 2, 1, 4, 2, 2)  /*  64x128   8 waves of 16x64                    */ \
    X(4, 4, 4, 2, 2, 1, 2)  /* 128x128  16 waves of 32x32                    */ \
    X(5, 2, 2, 2, 2, 4, 2)  /*  64x64    4 waves of 32x32                    */ \
    X(6, 2, 4, 2, 2, 2, 2)  /*  64x128   8 waves of 32x32                    */ \
    X(7, 4, 2, 2, 2, 2, 2)  /* 128x64    8 waves of 32x32                    */ \
    X(8, 4, 4, 1, 2, 1, 2)  /*  64x128  16 waves of 16x32                    */ \
    X(9, 4, 2, 2, 4, 2, 4)  /* 128x128   8 waves of 32x64, 4-block K-steps   */ \
    X(10, 2, 2, 2, 4, 3, 4) /*  64x128   4 waves of 32x64, 4-block K-steps   */ \
    X(11, 4, 2, 2, 2, 2, 4) /* 128x64    8 waves of 32x32, 4-block K-steps   */ \
    X(12, 2, 2, 2, 2, 4, 4) /*  64x64    4 waves of 32x32, 4-block K-steps   */ \
    X(13, 2, 4, 2, 2, 2, 4) /*  64x128   8 waves of 32x32, 4-block K-steps   */

int g_gemm_force_cfg = -1;  // debug / autotune hook: >= 0 forces one configuration (>= 100: the 32x32x32 kernel)
// (Round 3 also carried an fp6 block-scaled operand form of the 32x32 kernel -- v_mfma_scale_f32_32x32x64_f8f6f4 on E2M3 copies of the
// nibbles and of the Q8_0 quants, bit-identical, 24 fewer VALU operations per tile and block.  Measured -15 % .. +6 % of the int8 form's
// kernel time at +50 % weight bytes (profiles/r03_fp6_vs_i8_kernels.txt: the kernel is not issue-bound); removed in round 4.)

// gemm_q4_mfma32.hip
bool gemm32_supports(const fl_qtensor &W, int cfg, bool silu);
hipError_t gemm32_launch(int cfg, const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                         const float *resid, int ldr, const GemmSiluEpi &epi);

template <int TYPE, int WM, int WN, int TM, int TN, int MINW, int KS>
static hipError_t launch_gemm(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                              const float *resid, int ldr, const GemmSiluEpi &epi) {
    using Cfg = GemmCfg<TYPE, WM, WN, TM, TN, KS>;
    const int MGT = W.M16 / 16, NGT = fl_roundup(N, 16) / 16;
    const int tiles = ((MGT + Cfg::MG - 1) / Cfg::MG) * ((NGT + Cfg::NG - 1) / Cfg::NG);
    auto kern = gemm_q4_mfma_kernel<TYPE, WM, WN, TM, TN, MINW, KS>;
    if (Cfg::LDS_BYTES > 65536) {        // the attribute is per device: remember which devices have it (several GPUs in one process)
        static bool attr_set[64] = {false};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), Cfg::LDS_BYTES, st, reinterpret_cast<const uint4 *>(W.qs), W.d,
                       W.m, xq.q, xq.d, xq.s, N, W.M, MGT, NGT, W.KB, y, ldy, resid, ldr, epi);
    return hipGetLastError();
}

// Choose the workgroup shape from the problem shape (sweeps: scripts/sweep_cfg.py, profiles/r01_gemm_cfg_sweep.txt).
// Two things decide: (1) how evenly the workgroups cover the 256 CUs -- `quant` = work of the busiest CU relative to a
// perfect split, which is what makes e.g. M = 5120 or 27648 prefer the smaller tiles; (2) among equally balanced
// shapes the larger wave tile wins when there is plenty of work (fewer LDS bytes and barriers per MFMA), the many-small-
// waves shape when there is little (all 1024 SIMDs busy, latency hidden by occupancy).
static int pick_config(int MGT, int NGT, int type) {
    if (g_gemm_force_cfg >= 0) return g_gemm_force_cfg;
    if (g_gemm_force_cfg != -2) {
        // gemm_q4_mfma32.hip (round 2): 128x64 tiles of four 32x64 waves when that gives every CU at least two workgroups,
        // else 128x32 tiles of four 32x32 waves (profiles/r02_gemm32_cfg_sweep.txt).  -2 selects the 16x16 kernel below.
        const int64_t tiles106 = (int64_t)((MGT + 7) / 8) * ((NGT + 3) / 4);
        if (!(NGT > 2 && tiles106 >= 512)) return 101;
        // a launch lasts as long as its busiest SIMD: when the last round of 256 workgroups would be at most half full, its rows
        // get 128x32 tiles instead (cfg 116, gemm_q4_mfma32_mixed_kernel; LLaMA-7B w1|w3 at n_batch 512: 5.375 rounds).  -3: never.
        const int64_t part = tiles106 % 256;
        if (g_gemm_force_cfg != -3 && tiles106 >= 512 && part >= 16 && part <= 128) return 116;
        return 106;
    }
    const double tiles16 = (double)MGT * NGT;        // 16x16 output tiles
    const double per_simd = tiles16 / 1024;          // MI355X: 256 CUs x 4 SIMDs
    auto quant = [&](int MG, int NG) {
        const int64_t wgs = (int64_t)((MGT + MG - 1) / MG) * ((NGT + NG - 1) / NG);
        return (double)((wgs + 255) / 256) * 256.0 * MG * NG / tiles16;
    };
    const double q2 = quant(8, 8), q1 = quant(4, 8), q5 = quant(4, 4), q7 = quant(8, 4);
    if (type == FL_TYPE_Q4_1) return 12;                                         // every Q4_1 shape (measured): 64x64 tiles, 4 waves of 32x32
    if (per_simd < 12) return q1 > 1.1 * q5 ? 12 : 13;                           // small outputs: 32x32 wave tiles, 4-block K-steps
    int best = 2;
    double qb = q2;
    if (q1 < 0.96 * qb) { best = 10; qb = q1; }
    if (q5 < 0.93 * qb) { best = 12; qb = q5; }
    if (per_simd < 80 && best == 2 && q1 <= q2) best = 10;                        // 64x128 tiles of 4 waves, 4-block K-steps
    (void)q7;
    return best;
}

static hipError_t gemm_dispatch(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                                const float *resid, int ldr, const GemmSiluEpi &epi) {
    if (resid && ((ldr & 3) != 0 || (reinterpret_cast<uintptr_t>(resid) & 15) != 0)) return hipErrorInvalidValue;
    if ((ldy & 3) != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return hipErrorInvalidValue;
    const int MGT = W.M16 / 16, NGT = fl_roundup(N, 16) / 16;
    int cfg = pick_config(MGT, NGT, W.type);
    if (cfg >= 100) {
        if (gemm32_supports(W, cfg, epi.silu_tab != nullptr)) return gemm32_launch(cfg, W, xq, N, y, ldy, st, resid, ldr, epi);
        cfg = 12;
    }
    if (epi.silu_tab && (cfg == 3 || cfg == 8)) cfg = 12;   // the silu epilogue pairs two row groups per wave: TM must be even
#define X(ID, WM, WN, TM, TN, MINW, KS)                                                                     \
    if (cfg == ID)                                                                                          \
        return W.type == FL_TYPE_Q4_0 ? launch_gemm<FL_TYPE_Q4_0, WM, WN, TM, TN, MINW, KS>(W, xq, N, y, ldy, st, resid, ldr, epi) \
                                      : launch_gemm<FL_TYPE_Q4_1, WM, WN, TM, TN, MINW, KS>(W, xq, N, y, ldy, st, resid, ldr, epi);
    FL_GEMM_CONFIGS(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t gemm_q4_mfma(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                        const float *resid, int ldr) {
    return gemm_dispatch(W, xq, N, y, ldy, st, resid, ldr, GemmSiluEpi{});
}

// W = w1|w3 with 16-row groups interleaved (group 2p = w1 rows [16p, 16p+16), group 2p+1 = the same rows of w3);
// out <- Q8_0(silu(w1 x) * (w3 x)) in QA16, n_ff = W.M / 2 features
hipError_t gemm_q4_mfma_silu(const fl_qtensor &W, const fl_qact &xq, int N, const uint16_t *silu_tab, const fl_qact &out,
                             hipStream_t st) {
    if (!silu_tab || W.M % 64 != 0) return hipErrorInvalidValue;
    GemmSiluEpi epi{};
    epi.silu_tab = silu_tab; epi.oq = out.q; epi.od = out.d; epi.os = out.s; epi.KBo = W.M / 64;
    return gemm_dispatch(W, xq, N, nullptr, 4, st, nullptr, 0, epi);
}

// W = wq|wk|wv stacked ([3 El][K]); y <- rope(Q) rows ([N][ldy], only the first El columns are written), K cache rows
// n_past.. <- rope(K), transposed V cache columns n_past.. <- V
hipError_t gemm_q4_mfma_qkv(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, const float *rope_tab, float *kc,
                            float *vc, int El, int D, int n_past, int n_ctx, hipStream_t st) {
    if (!rope_tab || W.M != 3 * El || El % 4 != 0 || D % 4 != 0 || (ldy & 3) != 0) return hipErrorInvalidValue;
    GemmSiluEpi epi{};
    epi.rope_tab = reinterpret_cast<const float2 *>(rope_tab);
    epi.kc = kc; epi.vc = vc; epi.El = El; epi.D = D; epi.n_past = n_past; epi.n_ctx = n_ctx;
    return gemm_dispatch(W, xq, N, y, ldy, st, nullptr, 0, epi);
}

}  // namespace fl
