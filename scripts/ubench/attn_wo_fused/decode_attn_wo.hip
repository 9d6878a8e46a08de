// decode_attn_wo.hip -- reference-order decode: the attention of a token and the wo matmul that follows it, ONE launch (round 6).
//
// The reference walks KQ .. KQV (lib/llama.cpp:364-398) and then wo + the residual add (:401-407) node after node.  As two launches the pair cost
// 8.6 + 5.3 us + a 1.4 us boundary for LLaMA-7B (profiles/r06_decode_kernels.md): during the attention -- 32 workgroups, a few round trips, no
// bandwidth -- HBM idles, and wo then starts cold: 2.2 us until its first weight byte arrives, its prologue behind that.  Here
//   * workgroups 0 .. H-1 are the attention's heads (decode_attention_body, decode_attention.h: the same code as decode_attention_kernel); each
//     writes the Q8_0 blocks of its head THROUGH to memory (agent-scope stores), waits for them (s_waitcnt) and adds one to a counter;
//   * workgroups H .. are the wo matmul's row groups in the K-sliced form of gemv1_q4_exact_llc.hip (NK waves split a 16-row group along K; lane =
//     (row, k-group) owns the chains 2g, 2g+1): they request their WEIGHTS and the residual at entry -- the 10.5 MB stream runs under the attention --
//     then one lane polls the counter (relaxed agent-scope loads, s_sleep between them), the activation's 5 KB come in through agent-scope loads
//     (served by memory / L2, never by this CU's L1) and the lane sums, the chains in block order and the store follow as in the llc kernel.
// Every workgroup of the launch is resident at once (H + row groups / TEAMS <= CUs, checked by the launcher; the heads have the lowest indices and
// depend on nothing in the launch), so the wait cannot starve its producers; it is bounded all the same (a trap after 5 s: a hung queue is worse
// than a failed eval).  The counters are back at zero when the launch ends: the last wo workgroup to pass the wait resets them.
// Arithmetic and order: exactly the two kernels' -- tests/test_exact_gpu.py compares the fused launch with them and with the oracle bit for bit.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <algorithm>
#include "q4_device.h"
#include "q4_kernels.h"
#include "eval_kernels.h"
#include "decode_attention.h"

#pragma clang fp contract(off)

namespace fl {

typedef unsigned int awv4u __attribute__((ext_vector_type(4)));

template <int SRC>
__device__ __forceinline__ float aw_bcast(float v) {      // value of lane (quad base + SRC) of every quad
    return dpp_f32<SRC | (SRC << 2) | (SRC << 4) | (SRC << 6)>(v);
}

// NK waves along K per row group, QPW quads a wave holds, 512 threads = TEAMS = 8 / NK row groups per wo workgroup
template <int TYPE, int NK, int QPW, int ORD>
__global__ __launch_bounds__(512) void decode_attn_wo_kernel(
    const int *__restrict__ dyn_past, const float *__restrict__ qkv, const float2 *__restrict__ rope_tab, float *__restrict__ kc, float *__restrict__ vc,
    int E, int D, int n_past, int n_ctx, const uint16_t *__restrict__ exp_tab, float scale, int8_t *__restrict__ aq, float *__restrict__ ad,
    float *__restrict__ as, int H, int M, int units, int KB, const uint32_t *__restrict__ qwd, const float *__restrict__ dW,
    const float *__restrict__ mW, float *__restrict__ y, const float *__restrict__ resid, unsigned *__restrict__ sync /* [0] heads done, [1] wo workgroups past the wait */) {
    constexpr bool Q41 = TYPE == FL_TYPE_Q4_1;
    constexpr int TEAMS = 8 / NK, NT = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    if ((int)blockIdx.x < H) {
        decode_attention_body<ORD, true>(dsm, blockIdx.x, dyn_past, qkv, rope_tab, kc, vc, E, D, n_past, n_ctx, exp_tab, scale, aq, ad, as, nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's written-through stores have been acknowledged ...
        __syncthreads();                                         // ... and every wave's
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ float accs_[TEAMS][64][3];                                    // the chains' state between the K slices: a_2g, a_2g+1, summs
    const int wg = (int)blockIdx.x - H, nwo = (int)gridDim.x - H;
    const int lane = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (slice k of team t is wave t NK + (k - t) mod NK: the waves of one chain phase sit on different SIMDs, as in the llc kernel)
    const int team = wave_ / NK, k = TEAMS == 1 ? wave_ % NK : (wave_ + team) % NK;
    float (*accs)[3] = accs_[team];
    const int NQ = (KB + 3) >> 2;
    unsigned char *lx = dsm;                                                 // [NQ][4 k-groups][4 blocks][8 B], d [4 NQ], s [4 NQ] (llc kernel's layout)
    float *ld_ = reinterpret_cast<float *>(dsm + (size_t)NQ * 128);
    float *ls_ = ld_ + 4 * NQ;
    const int unit = min(wg * TEAMS + team, units - 1);
    const bool live = wg * TEAMS + team < units;                             // (a team past the last row group redoes it and stores nothing)
    const int qlo = (k * NQ) / NK, nq = ((k + 1) * NQ) / NK - qlo;
    const int r = lane >> 2, g = lane & 3;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(qwd), 0, (int)((uint32_t)units * (uint32_t)NQ * 1024u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dW), 0, (int)((uint32_t)units * (uint32_t)KB * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q41 ? mW : dW), 0, (int)((uint32_t)units * (uint32_t)KB * 64u), 0x00020000);
    const int voff_w = lane * 16, voff_d = (g * 16 + r) * 4;
    // ---- this wave's slice of the weight stream, all of it: it arrives while the heads work (nontemporal: read once per token).  Not at once:
    // the heads' first round trips (position, q / k / v, the rope row, the K / V history) would queue behind 10 MB of weight requests
#ifndef AW_DELAY
#define AW_DELAY 60
#endif
    if (AW_DELAY > 0) __builtin_amdgcn_s_sleep(AW_DELAY);                    // (x 64 clocks)
    awv4u w[QPW];
    float dw[QPW], mw[QPW];
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int q = qlo + (i < nq ? i : 0);                                // (past the slice: a cache-hot dummy, never used)
        const int sd = (unit * KB + 4 * q) * 64;
        dw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, voff_d, sd, 2));
        if (Q41) mw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff_d, sd, 2));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int q = qlo + (i < nq ? i : 0);
        w[i] = __builtin_bit_cast(awv4u, __builtin_amdgcn_raw_buffer_load_b128(rW, voff_w, (unit * NQ + q) * 1024, 2 /* nt */));
    }
    float rsd = 0.f;
    if (resid && k == NK - 1) rsd = resid[min(unit * 16 + r, M - 1)];       // (wave-uniform condition, clamped address)
    __builtin_amdgcn_sched_barrier(0);
    // blocks past K in a partial last quad: zero quants, d_x = s_x = 0
    if ((int)threadIdx.x < (4 * NQ - KB) * 4) {
        const int b = KB + ((int)threadIdx.x >> 2), gg = threadIdx.x & 3;
        *reinterpret_cast<uint2 *>(lx + (((b >> 2) * 4 + gg) * 4 + (b & 3)) * 8) = make_uint2(0, 0);
        if (gg == 0) { ld_[b] = 0.f; ls_[b] = 0.f; }
    }
    // ---- the heads: one lane waits for all H of them (each has written its blocks through and seen the acknowledgements before it counted)
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)H) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 500000000ull) __builtin_trap();       // 5 s of the 100 MHz clock: the heads never ran (cannot happen on a healthy queue)
        }
        // the last wo workgroup past the wait puts both counters back (nobody polls [0] any more; the next launch starts from zero)
        if (__hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwo - 1) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    // ---- the activation: QA1 planes (k-group bytes e0,e2,e4,e6 | e1,e3,e5,e7) -> the lanes' view in LDS; agent-scope loads (never this CU's L1)
    for (int i = threadIdx.x; i < KB * 4; i += NT) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(aq) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        const int b = i >> 2, gg = i & 3;
        *reinterpret_cast<uint2 *>(lx + (((b >> 2) * 4 + gg) * 4 + (b & 3)) * 8) =
            make_uint2(__builtin_amdgcn_perm(hi, lo, 0x05010400u), __builtin_amdgcn_perm(hi, lo, 0x07030602u));
    }
    for (int i = threadIdx.x; i < KB; i += NT) {
        ld_[i] = __hip_atomic_load(ad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ls_[i] = Q41 ? __hip_atomic_load(as + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    __syncthreads();
    // ---- order-free part, all waves at once: per block the two lane sums of this lane's k-group as floats, rn(d_w d_x)
    const uint32_t m8 = 0xF0F0F0F0u;
    float f0[QPW][4], f1[QPW][4], dd[QPW][4];
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
            const float dwb[4] = {aw_bcast<0>(dw[i]), aw_bcast<1>(dw[i]), aw_bcast<2>(dw[i]), aw_bcast<3>(dw[i])};
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                uint32_t wa, wb;
                if (TYPE == FL_TYPE_Q4_0) { wa = (wv[blk] << 4) & m8; wb = wv[blk] & m8; }     // 16 (nib - 8): elements 0..3 | 4..7
                else { wa = wv[blk] & 0x0F0F0F0Fu; wb = (wv[blk] >> 4) & 0x0F0F0F0Fu; }
                f0[i][blk] = (float)__builtin_amdgcn_sdot4((int)wa, (int)xa[blk], 0, false);
                f1[i][blk] = (float)__builtin_amdgcn_sdot4((int)wb, (int)xb[blk], 0, false);
                dd[i][blk] = __fmul_rn(dwb[blk], dxv[blk]);                 // rn(d_w d_x); a block past K: d_x = 0
            }
        }
    }
    // ---- the chains, slice after slice: wave k continues from the state wave k - 1 left in LDS
    float a0 = 0.f, a1 = 0.f, summs = 0.f;
#pragma unroll 1
    for (int ph = 0; ph < NK; ++ph) {
        if (k == ph) {                                                      // (wave-uniform)
            if (ph > 0) { a0 = accs[lane][0]; a1 = accs[lane][1]; if (Q41) summs = accs[lane][2]; }
#pragma unroll
            for (int i = 0; i < QPW; ++i) {
                if (i < nq) {
                    float sxv[4] = {0.f, 0.f, 0.f, 0.f}, msb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (Q41) {
                        msb[0] = aw_bcast<0>(mw[i]); msb[1] = aw_bcast<1>(mw[i]); msb[2] = aw_bcast<2>(mw[i]); msb[3] = aw_bcast<3>(mw[i]);
                        const float4 sx4 = *reinterpret_cast<const float4 *>(ls_ + 4 * (qlo + i));
                        sxv[0] = sx4.x; sxv[1] = sx4.y; sxv[2] = sx4.z; sxv[3] = sx4.w;
                    }
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk) {
                        a0 = __fmaf_rn(dd[i][blk], f0[i][blk], a0);
                        a1 = __fmaf_rn(dd[i][blk], f1[i][blk], a1);
                        if (Q41) summs = __fmaf_rn(msb[blk], sxv[blk], summs);
                    }
                }
            }
            if (ph < NK - 1) { accs[lane][0] = a0; accs[lane][1] = a1; if (Q41) accs[lane][2] = summs; }
        }
        if (ph < NK - 1) __syncthreads();
    }
    // ---- the row group is complete in its last wave: ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) over the quad of lanes that holds the row
    if (k == NK - 1) {
        float e = a0, o = a1;
        e = __fadd_rn(e, dpp_f32<DPP_XOR2>(e));
        o = __fadd_rn(o, dpp_f32<DPP_XOR2>(o));
        e = __fadd_rn(e, dpp_f32<DPP_XOR1>(e));
        o = __fadd_rn(o, dpp_f32<DPP_XOR1>(o));
        float v = __fadd_rn(e, o);
        if (Q41) v = __fadd_rn(v, summs);
        const int row = unit * 16 + r;
        if (g == 0 && row < M && live) {
            if (resid) v = __fadd_rn(v, rsd);
            y[row] = v;
        }
    }
}

// hipErrorInvalidValue: a shape outside the fused launch's reach -> the caller launches the attention and wo one after the other
hipError_t decode_attn_wo(const float *qkv, int E, int D, int H, int n_past, int n_ctx, const float *rope_tab, float *kc, float *vc,
                          const uint16_t *exp_tab, float scale, const fl_qact &act, const fl_qtensor &W, float *y, const float *resid,
                          unsigned *sync, hipStream_t st, const int *dyn_past, bool exact) {
    if (!exact || !W.qwd || !sync) return hipErrorInvalidValue;
    if (D % 32 != 0 || D > 128 || n_ctx % 4 != 0 || E % 4 != 0 || W.K != E || H * D != E) return hipErrorInvalidValue;
    const int KB = W.KB, NQ = (KB + 3) / 4, units = W.M16 / 16;
    if (units < 1 || NQ > 44) return hipErrorInvalidValue;                                     // (K <= 5632: the slices a wave holds in registers)
    if (qwd_bytes(W) >= (1ull << 31) || (size_t)W.M16 * (size_t)KB * 4 >= (1ull << 31)) return hipErrorInvalidValue;
    static const int n_cus = [] { int d = 0, c = 0; return (hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && c > 0) ? c : 0; }();
    const size_t lds_att = (size_t)(4 * D + n_ctx + 4) * 4 + 8 * 8 + 8 * 4, lds_wo = (size_t)NQ * 160;
    const size_t lds = std::max(lds_att, lds_wo);
    if (lds > 60 * 1024) return hipErrorInvalidValue;
#define FL_AW_GO(TYPE, NK, QPW)                                                                                                            \
    do {                                                                                                                                   \
        const int grid = H + (units + (8 / NK) - 1) / (8 / NK);                                                                            \
        if (grid > n_cus) return hipErrorInvalidValue;      /* every workgroup resident at once: one 512-thread workgroup per CU */         \
        hipLaunchKernelGGL((decode_attn_wo_kernel<TYPE, NK, QPW, 1>), dim3(grid), dim3(512), lds, st, dyn_past, qkv,                       \
                           reinterpret_cast<const float2 *>(rope_tab), kc, vc, E, D, n_past, n_ctx, exp_tab, scale, act.q, act.d, act.s, H, \
                           W.M, units, KB, W.qwd, W.d, W.m, y, resid, sync);                                                               \
        return hipGetLastError();                                                                                                          \
    } while (0)
    if (W.type == FL_TYPE_Q4_0) {
        if (NQ <= 32) FL_AW_GO(FL_TYPE_Q4_0, 4, 8);
        else FL_AW_GO(FL_TYPE_Q4_0, 4, 11);
    } else {          // (the llc kernel runs Q4_1 as 8 x 4 to stay under 128 registers; here one workgroup per CU has 256 either way)
        if (NQ <= 32) FL_AW_GO(FL_TYPE_Q4_1, 4, 8);
        else FL_AW_GO(FL_TYPE_Q4_1, 4, 11);
    }
#undef FL_AW_GO
}

}  // namespace fl
