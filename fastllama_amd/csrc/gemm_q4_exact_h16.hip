// gemm_q4_exact_h16.hip -- the reference-order ("exact") Q4 x Q8_0 matmul for prefill (N >= 9), round 4 form.
//
// What must be reproduced (ggml_vec_dot_q4_{0,1}_q8_0, AVX2 branch, /root/reference/lib/ggml.c:2445-2487, :2639-2689): per output
// 8 f32 accumulators, accumulator j taking  acc_j = fma(d_w * d_x, float(sum of the products of elements 4j..4j+3), acc_j)
// block after block, then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) [+ the scalar chain summs = fma(m_w, s_x, summs) for Q4_1].
// The 8 fma per (output, block) are VALU work nothing can remove (64 v_pk_fma_f32 per 32x32 tile and block = 256 cycles) and the
// 8 lane sums cost four v_mfma_f32_32x32x4_2b_f16 (two lane sums of a 32x32 tile each, 64 cycles: the matrix pipe delivers 32
// results per cycle whatever the shape) = 256 cycles; on gfx950 the two pipes of a SIMD do not overlap, so 512 + 32 (the d_w x d_x
// outer product, one MFMA per block PAIR) is the floor of the reference's order.  Round 3's kernel (gemm_q4_exact_mfma.hip)
// spent 1020: it unpacked nibbles -> f16 for every 32-column tile again (40 VALU ops per tile and block), converted the
// activations while staging them, and kept a wave's weights private.  Here NOTHING but the fma chain is left on the VALU:
//   * both operands are read as ready-made f16 MFMA fragments (q4_layout.h "H16 copies": WH16 built once per tensor, XH16 by
//     whoever produces the Q8_0 activations), 2 KiB per (32-row tile, block), moved HBM/L2 -> LDS by buffer_load ... lds;
//   * a workgroup = 4 waves = 64 x 64 outputs (2 x 2 wave tiles): every fragment staged in LDS is read by two waves;
//     3-stage ring of block PAIRS, one barrier per pair, DMA counted with s_waitcnt vmcnt (never 0 inside the loop);
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
#include "q4_device.h"
#include "q4_kernels.h"
#include "gemm_epi.h"

#pragma clang fp contract(off)

namespace fl {

typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));

template <int TYPE>
struct XH {
    static constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    static constexpr int NSTAGE = 3;
    static constexpr int OFF_B = 8192;                         // A: [2 row tiles][2 blocks][2 parts][1 KiB], then B the same
    static constexpr int OFF_SC = 16384;                       // d_w [2 blocks][64 rows], d_x [2][64 cols] (, m_w, s_x)
    static constexpr int STAGE = OFF_SC + (Q41 ? 2048 : 1024);
    static constexpr int LPW = Q41 ? 6 : 5;                    // DMA instructions per wave and stage
    static constexpr int ACT_BYTES = 32 * 64 * 4;              // f32 tile of the silu epilogue (reuses the ring)
    static constexpr int LDS_BYTES = NSTAGE * STAGE;
};

enum { EPI_PLAIN = 0, EPI_ROPE = 1, EPI_SILU = 2 };

#if defined(__HIP_DEVICE_COMPILE__)
#ifdef XH_NOPK
#define XH_ATTR __attribute__((target("no-packed-fp32-ops")))
#else
#define XH_ATTR
#endif
template <int TYPE, int EPI>
__global__ __launch_bounds__(256, 2) XH_ATTR void gemm_q4_exact_h16_kernel(
    const uint16_t *wh, const float *dW, const float *mW, const uint16_t *xh, const float *xd, const float *xs, int N, int M,
    int MT32 /* 32-row tiles of wh */, int MGT /* 16-row groups of dW */, int NT32, int NGT, int KB, float *__restrict__ y, int ldy,
    const float *__restrict__ resid, int ldr, GemmSiluEpi epi) {
    using C = XH<TYPE>;
    constexpr bool Q41 = C::Q41;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;

    // ---- XCD-aware bijective remap of the tile id (as gemm_q4_mfma32.hip): workgroup b runs on XCD b % 8; the column tiles that
    //      share a weight row panel get consecutive ids on ONE XCD, so the panel comes from HBM once and is re-read from that L2.
    const int tiles_m = (MT32 + 1) >> 1, tiles_n = (NT32 + 1) >> 1;
    int bid = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int tn = bid % tiles_n, tm = bid / tiles_n;

    // ---- DMA plan.  Slots 0..3 of a wave: 1-KiB fragment pieces p = wave + 4 s (p < 8: A, else B); slot 4: one 256-byte scale
    //      piece (wave 0/1: d_w of block 0/1, wave 2/3: d_x); slot 5 (Q4_1): m_w / s_x likewise.  Everything but the lane offset is
    //      wave-uniform.  Out-of-range tiles / groups (and, on the activation side, blocks past K) read zeros through the bounds
    //      check of the buffer descriptor.
    v4i rsrc[C::LPW];
    uint32_t voff[C::LPW];
    int unit[C::LPW], loff[C::LPW], tailblk[C::LPW];
#pragma unroll
    for (int s = 0; s < C::LPW; ++s) {
        const void *base;
        uint32_t bytes;
        tailblk[s] = -1;
        if (s < 4) {
            const int p = wave + 4 * s, side = p >> 3, t = (p >> 2) & 1, blk = (p >> 1) & 1, pp = p & 1;
            const int tile = (side ? tn : tm) * 2 + t, ntile = side ? NT32 : MT32;
            base = side ? (const void *)xh : (const void *)wh;
            bytes = (uint32_t)ntile * (uint32_t)KB * 2048u;
            voff[s] = tile < ntile ? ((uint32_t)tile * (uint32_t)KB + (uint32_t)blk) * 2048u + (uint32_t)pp * 1024u + (uint32_t)lane * 16u
                                   : 0x80000000u;
            unit[s] = 2048;
            loff[s] = side * C::OFF_B + ((t * 2 + blk) * 2 + pp) * 1024;
        } else {
            const int side = wave >> 1, blk = wave & 1, pl = s - 4;            // pl 0: d planes, 1: m_w / s_x
            const int g = (side ? tn : tm) * 4 + (lane >> 4), ng = side ? NGT : MGT;
            base = side ? (const void *)(pl ? xs : xd) : (const void *)(pl ? mW : dW);
            bytes = (uint32_t)ng * (uint32_t)KB * 64u;
            voff[s] = g < ng ? (((uint32_t)g * (uint32_t)KB + (uint32_t)blk) * 16u + (uint32_t)(lane & 15)) * 4u : 0x80000000u;
            unit[s] = 64;
            loff[s] = C::OFF_SC + pl * 1024 + side * 512 + blk * 256;
            if (side) tailblk[s] = blk;                                        // a block past K must contribute nothing: d_x = s_x = 0
        }
        const uint64_t bp = (uint64_t)(uintptr_t)base;
        rsrc[s] = v4i{__builtin_amdgcn_readfirstlane((int)(uint32_t)bp), __builtin_amdgcn_readfirstlane((int)((bp >> 32) & 0xFFFF)),
                      __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
        unit[s] = __builtin_amdgcn_readfirstlane(unit[s]);
        loff[s] = __builtin_amdgcn_readfirstlane(loff[s]);
        tailblk[s] = __builtin_amdgcn_readfirstlane(tailblk[s]);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto fill = [&](int st, int kb0, int s_lo = 0, int s_hi = C::LPW) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < C::LPW; ++s) {
            if (s < s_lo || s >= s_hi) continue;
            uint32_t vo = voff[s];
            if (s >= 4 && tailblk[s] >= 0 && kb0 + tailblk[s] >= KB) vo = 0x80000000u;
            const uint32_t dst = lds0 + (uint32_t)(st * C::STAGE + loff[s]);
            if (s < 4)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(vo), "s"(rsrc[s]), "s"(kb0 * unit[s]) : "memory", "m0");
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(vo), "s"(rsrc[s]), "s"(kb0 * unit[s]) : "memory", "m0");
        }
    };

    v16f acc[8], summs;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) summs[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[j]));    // (opaque zeros: no peeled first trip with four D tiles alive)
    const v32f zero32 = {};

    // per-lane LDS offsets inside a stage: fragments of block u at + u * 2048 (+ 1024 for the second part), scales of block h
    const int a_off = wr * 4096 + lane * 16, b_off = C::OFF_B + wc * 4096 + lane * 16;
    const int sw_off = C::OFF_SC + h * 256 + (32 * wr + i) * 4, sx_off = C::OFF_SC + 512 + h * 256 + (32 * wc + i) * 4;

    const int nsteps = (KB + 1) >> 1;
    fill(0, 0);
    fill(1, 2);
    int cur = 0;
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::LPW) : "memory");          // this wave's pieces of stage t have landed
        __builtin_amdgcn_s_barrier();                                          // ... everyone's have, and everyone is done with stage t - 1
        const int nst = cur == 0 ? 2 : cur - 1;                                // block pair t + 2 goes into the stage pair t - 1 occupied
#if !defined(XH_SPREAD) && !defined(XH_NODMA)
        fill(nst, 2 * (t + 2));
#endif
        const unsigned char *base = smem + cur * C::STAGE;
        const float dw = *reinterpret_cast<const float *>(base + sw_off), dx = *reinterpret_cast<const float *>(base + sx_off);
        const v32f P = __builtin_amdgcn_mfma_f32_32x32x1f32(dw, dx, zero32, 0, 0, 0);       // rn(d_w d_x) of blocks 2t (regs 0..15), 2t+1
        if (Q41) {                                                             // summs: one chain, block 2t then 2t + 1 (k = 0, 1)
            const float mw = *reinterpret_cast<const float *>(base + sw_off + 1024), sx = *reinterpret_cast<const float *>(base + sx_off + 1024);
            summs = __builtin_amdgcn_mfma_f32_32x32x2f32(mw, sx, summs, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint4 a01 = *reinterpret_cast<const uint4 *>(base + a_off + u * 2048);
            const uint4 a23 = *reinterpret_cast<const uint4 *>(base + a_off + u * 2048 + 1024);
            const uint4 b01 = *reinterpret_cast<const uint4 *>(base + b_off + u * 2048);
            const uint4 b23 = *reinterpret_cast<const uint4 *>(base + b_off + u * 2048 + 1024);
            v2u32 af[4] = {{a01.x, a01.y}, {a01.z, a01.w}, {a23.x, a23.y}, {a23.z, a23.w}};
            const v2u32 bf[4] = {{b01.x, b01.y}, {b01.z, b01.w}, {b23.x, b23.y}, {b23.z, b23.w}};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#ifdef XH_NOCOMP
                v32f D = zero32;
                asm volatile("" : "+v"(D) : "v"(af[s]), "v"(bf[s]));
#else
                const v32f D = __builtin_amdgcn_mfma_f32_32x32x4f16(__builtin_bit_cast(v4h, af[s]), __builtin_bit_cast(v4h, bf[s]), zero32, 0, 0, 0);
#endif
#if defined(XH_SPREAD) && !defined(XH_NODMA)
                if (u * 4 + s < C::LPW) fill(nst, 2 * (t + 2), u * 4 + s, u * 4 + s + 1);   // one DMA piece in the shadow of each of the first MFMAs
#endif
#ifndef XH_NOCOMP
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[2 * s][e] = __builtin_fmaf(P[16 * u + e], D[e], acc[2 * s][e]);
                    acc[2 * s + 1][e] = __builtin_fmaf(P[16 * u + e], D[16 + e], acc[2 * s + 1][e]);
                }
#else
                acc[2 * s][0] += D[0];
#endif
                // Pin the order: these 32 fma are issued before the NEXT lane-sum MFMA (whose A operand passes through this
                // statement) -- left alone, the compiler sinks them to the next block's and keeps four D tiles (128 VGPRs) alive.
                if (s < 3) asm volatile("" : "+v"(acc[2 * s]), "+v"(acc[2 * s + 1]), "+v"(af[s + 1]));
                else asm volatile("" : "+v"(acc[2 * s]), "+v"(acc[2 * s + 1]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the fills past the last pair (zeros / unused)

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
        float *act = reinterpret_cast<float *>(smem);                          // [32 features][64 columns] f32
        __syncthreads();                                                       // every wave is done with the operand ring
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a1 = out[4 * g + e], a3 = out[4 * (g + 2) + e];
                const uint16_t hx = __half_as_ushort(__float2half_rn(a1));                      // GGML_FP32_TO_FP16
                const float sl = __half2float(__ushort_as_half(epi.silu_tab[hx]));              // table_silu_f16
                act[(wr * 16 + 8 * g + 4 * h + e) * 64 + wc * 32 + i] = __fmul_rn(sl, a3);     // ggml_mul(silu, tmp)
            }
        __syncthreads();
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
    static_assert(C::LDS_BYTES <= 65536 && C::ACT_BYTES <= C::LDS_BYTES, "no dynamic-LDS attribute needed");
    const int MT32 = (W.M16 + 31) / 32, NT32 = (N + 31) / 32;
    const int tiles = ((MT32 + 1) / 2) * ((NT32 + 1) / 2);
    hipLaunchKernelGGL((gemm_q4_exact_h16_kernel<TYPE, EPI>), dim3(tiles), dim3(256), C::LDS_BYTES, st, W.h16, W.d, W.m, xq.h16, xq.d, xq.s, N,
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
