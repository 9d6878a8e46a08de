// gemv_prologue.h -- the decode prologues of the exact-mode single-token matmul kernel (gemv1_q4_exact_kernel, exact_kernels.hip);
// the same code, statement for statement, as the prologue sections inside gemv_q4_kernel (q4_kernels.hip), which keeps its own
// copy: moving that kernel onto this struct changed its register allocation and cost the fast decode path 30 % (measured,
// round 3).  The activation arrives as f32 and the kernel builds its Q8_0 form in LDS itself,
// AFTER its weight loads are in flight -- one launch and one HBM round trip less per matmul; every workgroup redoes the small
// prologue from L2.
//   PRO = 1: rms_norm * weight -> Q8_0 (arithmetic and f64 sum order of rmsnorm_quant_kernel: first 256 threads)
//   PRO = 2: silu(w1 x) * (w3 x) -> Q8_0 (silu_mul_quant_kernel; xf = [w1 x (K) | w3 x (K)], aux = fp16 silu table)
//   PRO = 3: plain quantize_row_q8_0 of an f32 vector
// The LDS copy has the QA1 layout: q [KB][32] (k-groups at their natural position, bytes e0,e2,e4,e6,e1,e3,e5,e7), d [KB], s [KB].
// Usage (NT = threads of the workgroup, all of them call both):  GP_DECL(PRO);  GemvPrologue<PRO, NT>::issue(pv, pw, psl, psb, ...);
// <issue the weight loads>;  GemvPrologue<PRO, NT>::finish(pv, pw, psl, psb, ...)  -- ends with a __syncthreads(); vector-memory
// loads return in order, so the small activation loads must go out BEFORE the weight stream or the prologue would sit behind it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "q4_device.h"

namespace fl {

// quantize_row_q8_0 of one 8-element group (4 adjacent lanes = one block) into an LDS copy of the QA1 layout
// LXF: write the group straight into the lane view of gemv1_q4_exact_llc.hip instead -- LX [block >> 2][k-group][block & 3] 8-byte entries,
// bytes (e0..e3 | e4..e7) -- which saves that kernel a pass over LDS and a barrier (round 5)
template <bool LXF = false>
__device__ __forceinline__ void quantize_group_lds_x(const float o[8], int kg, int8_t *lq, float *ld_, float *ls_) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(o[i]));
    amax = quad_max_f32(amax);
    const float dd = __fdiv_rn(amax, 127.0f);
    const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
    int qi[8], isum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qi[i] = (int)rintf(__fmul_rn(o[i], id));
        isum += qi[i];
    }
    isum = quad_sum_i32(isum);
    auto pk = [](int a, int b, int c, int e) -> uint32_t {
        return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) |
               ((uint32_t)(e & 0xFF) << 24);
    };
    if (LXF) {
        const int b = kg >> 2, gg = kg & 3;
        *reinterpret_cast<uint2 *>(lq + (((b >> 2) * 4 + gg) * 4 + (b & 3)) * 8) = make_uint2(pk(qi[0], qi[1], qi[2], qi[3]), pk(qi[4], qi[5], qi[6], qi[7]));
    } else {
        *reinterpret_cast<uint2 *>(lq + kg * 8) = make_uint2(pk(qi[0], qi[2], qi[4], qi[6]), pk(qi[1], qi[3], qi[5], qi[7]));
    }
    if ((kg & 3) == 0) {
        ld_[kg >> 2] = dd;
        ls_[kg >> 2] = __fmul_rn(dd, (float)isum);
    }
}

// The register state between issue and finish lives in plain local arrays of the KERNEL (a struct holding them ended up in
// scratch memory: 172 bytes per lane and a 6 us prologue):
//   float pv[GP_MAXIT1][8], pw[GP_MAXIT1][8], psl[GP_SIT2][8], psb[GP_SIT2][8];   with the sizes below
#define GP_MAXIT 4               /* PRO == 1: 8-element groups per thread (256 threads: K <= 8192) */
#define GP_SIT 2                 /* PRO == 2: group-iterations whose loads precede the weight stream */
#define GP_QIT 4                 /* PRO == 3: group-iterations whose loads precede the weight stream (NT = 512: K <= 16384) */
#define GP_DECL(PRO) float pv[(PRO) == 1 ? GP_MAXIT : (PRO) == 3 ? GP_QIT : 1][8], pw[(PRO) == 1 ? GP_MAXIT : 1][8], psl[(PRO) == 2 ? GP_SIT : 1][8], psb[(PRO) == 2 ? GP_SIT : 1][8]

template <int PRO, int NT, bool LXF = false>
struct GemvPrologue {
    static constexpr int MAXIT = GP_MAXIT, SIT = GP_SIT, QIT = GP_QIT;
    template <typename A1, typename A1w, typename A2>
    static __device__ __forceinline__ void issue(A1 &v, A1w &ww, A2 &sl_, A2 &sb_, const float *__restrict__ xf,
                                                 const void *__restrict__ aux, int KB, int woven) {
        if constexpr (PRO == 1) {
            const float *nw = static_cast<const float *>(aux);
            const int gpr = KB * 4;
            if (threadIdx.x < 256) {
#pragma unroll
                for (int it = 0; it < MAXIT; ++it) {
                    const int kg = threadIdx.x + it * 256;
                    if (kg < gpr) {
                        const float4 a = *reinterpret_cast<const float4 *>(xf + kg * 8);
                        const float4 c = *reinterpret_cast<const float4 *>(xf + kg * 8 + 4);
                        v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
                        v[it][4] = c.x; v[it][5] = c.y; v[it][6] = c.z; v[it][7] = c.w;
                        const float4 wa = *reinterpret_cast<const float4 *>(nw + kg * 8);
                        const float4 wc = *reinterpret_cast<const float4 *>(nw + kg * 8 + 4);
                        ww[it][0] = wa.x; ww[it][1] = wa.y; ww[it][2] = wa.z; ww[it][3] = wa.w;
                        ww[it][4] = wc.x; ww[it][5] = wc.y; ww[it][6] = wc.z; ww[it][7] = wc.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[it][i] = 0.f, ww[it][i] = 0.f;
                    }
                }
            }
        }
        if constexpr (PRO == 3) {
            // round 5: the f32 vector's loads go out HERE, in front of the weight stream.  They used to be issued in finish(), behind it --
            // and vector-memory loads return in order: the prologue of the w2 matmul sat behind its whole 110 KB of weights
            // (6.1 us from entry to "prologue done" of a 11 us launch, profiles/r05_decode_timeline.md).
            const int gpr = KB * 4;
#pragma unroll
            for (int it = 0; it < QIT; ++it) {
                const int kg = threadIdx.x + it * NT;
                if (kg < gpr) {
                    const float4 a = *reinterpret_cast<const float4 *>(xf + kg * 8), c = *reinterpret_cast<const float4 *>(xf + kg * 8 + 4);
                    v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
                    v[it][4] = c.x; v[it][5] = c.y; v[it][6] = c.z; v[it][7] = c.w;
                }
            }
        }
        if constexpr (PRO == 2) {
            const uint16_t *silu_tab = static_cast<const uint16_t *>(aux);
            const int F = KB * 32, gpr = F >> 3;
            float sa_[SIT][8];
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int kg = threadIdx.x + it * NT;
                if (kg < gpr) {
                    // features 8kg..8kg+7 of w1 x and of w3 x (woven: 16-feature groups alternate, else the halves [F | F])
                    const float *pa = xf + (woven ? ((kg >> 1) << 5) + ((kg & 1) << 3) : kg * 8);
                    const int boff = woven ? 16 : F;
                    const float4 a0 = *reinterpret_cast<const float4 *>(pa), a1 = *reinterpret_cast<const float4 *>(pa + 4);
                    const float4 b0 = *reinterpret_cast<const float4 *>(pa + boff), b1 = *reinterpret_cast<const float4 *>(pa + boff + 4);
                    sa_[it][0] = a0.x; sa_[it][1] = a0.y; sa_[it][2] = a0.z; sa_[it][3] = a0.w;
                    sa_[it][4] = a1.x; sa_[it][5] = a1.y; sa_[it][6] = a1.z; sa_[it][7] = a1.w;
                    sb_[it][0] = b0.x; sb_[it][1] = b0.y; sb_[it][2] = b0.z; sb_[it][3] = b0.w;
                    sb_[it][4] = b1.x; sb_[it][5] = b1.y; sb_[it][6] = b1.z; sb_[it][7] = b1.w;
                }
            }
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int kg = threadIdx.x + it * NT;
                if (kg < gpr) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {   // the table gathers go out now; their results are used after the weight loads
                        const uint16_t hx = __half_as_ushort(__float2half_rn(sa_[it][i]));
                        sl_[it][i] = __half2float(__ushort_as_half(silu_tab[hx]));
                    }
                }
            }
        }
    }

    // lq [KB][32], ld_ [KB], ls_ [KB] in LDS; sh: 4 doubles of LDS scratch (PRO == 1).  ynorm (PRO == 1, optional): the f32
    // normalised vector, stored by the workgroup for which store_ynorm is set.
    template <typename A1, typename A1w, typename A2>
    static __device__ __forceinline__ void finish(A1 &v, A1w &ww, A2 &sl_, A2 &sb_, const float *__restrict__ xf,
                                                  const void *__restrict__ aux, int KB, int woven, int8_t *lq, float *ld_, float *ls_,
                                                  double *sh, float *__restrict__ ynorm, bool store_ynorm, long long *dbg = nullptr) {
        // (dbg: development builds only -- wall-clock stamps of the prologue's steps, scripts/dev/decode_timeline.py)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef GP_DIAG_SKIP   // timing-only diagnostic build (WRONG results): the prologue's arithmetic replaced by a constant fill of the LDS copy --
                      // what the norm / quantize COMPUTE costs a decode token end to end (profiles/r05_decode_exact.md)
        {
            const int n8 = KB * 4;
            for (int kg = threadIdx.x; kg < n8; kg += NT) {
                *reinterpret_cast<uint2 *>(lq + kg * 8) = make_uint2(0x01010101u, 0x01010101u);
                if ((kg & 3) == 0) { ld_[kg >> 2] = 1.f; ls_[kg >> 2] = 32.f; }
            }
            if (PRO == 1 && ynorm && store_ynorm)
                for (int i = threadIdx.x; i < KB * 32; i += NT) ynorm[i] = xf[i];
            asm volatile("" :: "v"(v[0][0]), "v"(ww[0][0]), "v"(sl_[0][0]), "v"(sb_[0][0]));      // (the loads stay: their round trips are still waited for)
            __syncthreads();
            return;
        }
#endif
        if constexpr (PRO == 1) {
            const int E = KB * 32, gpr = E >> 3;
            double sum = 0.0;
            if (threadIdx.x < 256) {
#pragma unroll
                for (int it = 0; it < MAXIT; ++it) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) sum += (double)__fmul_rn(v[it][i], v[it][i]);
                }
                if (dbg) { asm volatile("" :: "v"(sum)); dbg[0] = wall_clock64(); }      // x has arrived and is squared
                sum = wave_sum_f64(sum);                          // same order as block_sum_f64 (eval_kernels.hip)
                if (lane == 0) sh[wave] = sum;
            }
            __syncthreads();
            if (dbg) dbg[1] = wall_clock64();
            if (threadIdx.x < 256) {
                double t = 0.0;
                for (int i = 0; i < 4; ++i) t += sh[i];
                const float mean = (float)(t / (double)E);
                const float scale = __fdiv_rn(1.0f, sqrtf(mean + 1e-6f));
                if (dbg) { asm volatile("" :: "v"(scale), "v"(ww[0][0])); dbg[2] = wall_clock64(); }      // scale known, the norm weights have arrived
#pragma unroll
                for (int it = 0; it < MAXIT; ++it) {
                    const int kg = threadIdx.x + it * 256;
                    if (kg >= gpr) continue;   // whole quads leave together
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = __fmul_rn(ww[it][i], __fmul_rn(v[it][i], scale));
                    if (ynorm && store_ynorm) {
                        float4 *yp = reinterpret_cast<float4 *>(ynorm + kg * 8);
                        yp[0] = make_float4(o[0], o[1], o[2], o[3]);
                        yp[1] = make_float4(o[4], o[5], o[6], o[7]);
                    }
                    quantize_group_lds_x<LXF>(o, kg, lq, ld_, ls_);
                }
            }
            __syncthreads();
        } else if constexpr (PRO == 2) {
            const uint16_t *silu_tab = static_cast<const uint16_t *>(aux);
            const int F = KB * 32, gpr = F >> 3;
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int kg = threadIdx.x + it * NT;
                if (kg >= gpr) continue;                                   // gpr % 4 == 0: quads stay together
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = __fmul_rn(sl_[it][i], sb_[it][i]);
                quantize_group_lds_x<LXF>(o, kg, lq, ld_, ls_);
            }
            for (int kg = threadIdx.x + SIT * NT; kg < gpr; kg += NT) {   // very wide rows: the rest
                const float *pa = xf + (woven ? ((kg >> 1) << 5) + ((kg & 1) << 3) : kg * 8);
                const int boff = woven ? 16 : F;
                const float4 a0 = *reinterpret_cast<const float4 *>(pa), a1 = *reinterpret_cast<const float4 *>(pa + 4);
                const float4 b0 = *reinterpret_cast<const float4 *>(pa + boff), b1 = *reinterpret_cast<const float4 *>(pa + boff + 4);
                const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint16_t hx = __half_as_ushort(__float2half_rn(a[i]));
                    const float sl = __half2float(__ushort_as_half(silu_tab[hx]));
                    o[i] = __fmul_rn(sl, b[i]);
                }
                quantize_group_lds_x<LXF>(o, kg, lq, ld_, ls_);
            }
            __syncthreads();
        } else if constexpr (PRO == 3) {
            const int gpr = KB * 4;
#pragma unroll
            for (int it = 0; it < QIT; ++it) {
                const int kg = threadIdx.x + it * NT;
                if (kg >= gpr) continue;                                   // gpr % 4 == 0: quads stay together
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = v[it][i];
                quantize_group_lds_x<LXF>(o, kg, lq, ld_, ls_);
            }
            // very wide rows: the rest, behind the weights -- four group-iterations' loads at a time (one iteration at a time every one of them was a
            // dependent round trip behind the wave's weight requests: LLaMA-65B's w2, K = 22016 on 256 threads, took seven: round 5)
            // (only when more than three iterations are left: LLaMA-13B's w2, K = 13824, has 2.75 and lost 1.5 % to the batch's dummy loads and
            //  registers -- 307.9 -> 303.1 tok/s -- where 65B gained 1.4 %, 94.2 -> 95.6, alternating runs on one box, profiles/r05_decode_exact.md)
            constexpr int RB = 4;
            const bool batched = gpr - QIT * NT > 3 * NT;                  // (uniform)
            for (int kg = threadIdx.x + QIT * NT; !batched && kg < gpr; kg += NT) {
                const float4 a0 = *reinterpret_cast<const float4 *>(xf + kg * 8), a1 = *reinterpret_cast<const float4 *>(xf + kg * 8 + 4);
                const float o[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                quantize_group_lds_x<LXF>(o, kg, lq, ld_, ls_);
            }
            for (int kg0 = threadIdx.x + QIT * NT; batched && kg0 < gpr; kg0 += RB * NT) {
                float4 a[RB][2];
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int kg = min(kg0 + b * NT, gpr - 1);             // (past the row: a cache-hot dummy, never used -- unconditional loads)
                    a[b][0] = *reinterpret_cast<const float4 *>(xf + kg * 8);
                    a[b][1] = *reinterpret_cast<const float4 *>(xf + kg * 8 + 4);
                }
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int kg = kg0 + b * NT;
                    if (kg >= gpr) break;                                  // (gpr % 4 == 0 and NT % 4 == 0: quads stay together)
                    const float o[8] = {a[b][0].x, a[b][0].y, a[b][0].z, a[b][0].w, a[b][1].x, a[b][1].y, a[b][1].z, a[b][1].w};
                    quantize_group_lds_x<LXF>(o, kg, lq, ld_, ls_);
                }
            }
            __syncthreads();
        }
    }
};

}  // namespace fl
