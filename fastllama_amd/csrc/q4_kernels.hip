// q4_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for fastLLaMa's Q4_0/Q4_1 x Q8_0
// "quantize activations, integer block dot, scale" matmul path.
//
// Reference semantics being reproduced (all file:line into /root/reference):
//   quantize_row_q8_0                  lib/ggml.c:1299-1441   (AVX2 flavour :1341-1403, s :1433-1440)
//   ggml_vec_dot_q4_0_q8_0             lib/ggml.c:2368-2559
//   ggml_vec_dot_q4_1_q8_0             lib/ggml.c:2561-2714
//   dequantize_row_q4_0 / _q4_1        lib/ggml.c:1443-1665
//   ggml_compute_forward_mul_mat_q_f32 lib/ggml.c:7928-8176
//
// Exactness contract: everything integer (Q8_0 quants, block dots) and every scale (d, s) is
// bit-identical to the reference; the only freedom taken is the ORDER in which the per-block f32
// terms d_w*d_x*isum are added (the reference itself adds them in 8 interleaved partial sums).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "q4_kernels.h"
#include "q4_device.h"

namespace fl {

// ------------------------------------------------------------------------------------------------
// weights: AoS <-> QW16
// ------------------------------------------------------------------------------------------------
template <int TYPE>
__global__ void repack_qw16_kernel(const uint32_t *__restrict__ aos, int M, int M16, int KB,
                                   uint32_t *__restrict__ qs, float *__restrict__ d, float *__restrict__ mm,
                                   int *bad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (row, block)
    if (idx >= (int64_t)M16 * KB) return;
    // iterate in OUTPUT order so the 16-byte stores coalesce: idx = ((grp*KB + b)*16 + r)
    const int r = (int)(idx & 15);
    const int64_t gb = idx >> 4;
    const int b = (int)(gb % KB);
    const int grp = (int)(gb / KB);
    const int row = grp * 16 + r;
    constexpr int WPB = TYPE == FL_TYPE_Q4_0 ? 5 : 6;  // 32-bit words per AoS block
    uint32_t w[4] = {0, 0, 0, 0};
    float dv = 0.f, mv = 0.f;
    if (row < M) {
        const uint32_t *p = aos + ((int64_t)row * KB + b) * WPB;
        dv = __uint_as_float(p[0]);
        if (TYPE == FL_TYPE_Q4_1) mv = __uint_as_float(p[1]);
        // (a non-finite block scale is no model -- and the single-token kernels rely on finite scales where a partial last quad reads
        //  one scale past its row against d_x = 0, gemv1_q4_exact_llc.hip)
        if (!isfinite(dv) || (TYPE == FL_TYPE_Q4_1 && !isfinite(mv))) atomicOr(bad, 2);
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = p[WPB - 4 + g];
        if (TYPE == FL_TYPE_Q4_0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g] ^= 0x88888888u;
            if (dv != 0.f && fabsf(dv) < 1.8807909613156600e-37f /* 2^-122 */) atomicOr(bad, 1);
            dv *= 0.0625f;
        }
    }
    uint4 o;
    uint32_t *op = &o.x;
#pragma unroll
    for (int g = 0; g < 4; ++g) op[qw16_pos(r, g)] = w[g];
    reinterpret_cast<uint4 *>(qs)[idx] = o;
    d[idx] = dv;
    if (TYPE == FL_TYPE_Q4_1) mm[idx] = mv;
}

template <int TYPE>
__global__ void unpack_qw16_kernel(const uint32_t *__restrict__ qs, const float *__restrict__ d,
                                   const float *__restrict__ mm, int M, int KB, uint32_t *__restrict__ aos) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (row, block)
    if (idx >= (int64_t)M * KB) return;
    const int b = (int)(idx % KB);
    const int row = (int)(idx / KB);
    const int grp = row >> 4, r = row & 15;
    const int64_t src = ((int64_t)grp * KB + b) * 16 + r;
    constexpr int WPB = TYPE == FL_TYPE_Q4_0 ? 5 : 6;
    uint32_t *p = aos + idx * WPB;
    const uint4 v = reinterpret_cast<const uint4 *>(qs)[src];
    const uint32_t *vp = &v.x;
    float dv = d[src];
    if (TYPE == FL_TYPE_Q4_0) dv *= 16.0f;
    p[0] = __float_as_uint(dv);
    if (TYPE == FL_TYPE_Q4_1) p[1] = __float_as_uint(mm[src]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t w = vp[qw16_pos(r, g)];
        if (TYPE == FL_TYPE_Q4_0) w ^= 0x88888888u;
        p[WPB - 4 + g] = w;
    }
}

hipError_t repack_to_qw16(int type, const void *aos, int M, int K, uint32_t *qs, float *d, float *m,
                          int *bad, hipStream_t st) {
    const int M16 = fl_roundup(M, 16), KB = K / FL_QK;
    const int64_t total = (int64_t)M16 * KB;
    const int grid = (int)((total + 255) / 256);
    if (type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(repack_qw16_kernel<FL_TYPE_Q4_0>, dim3(grid), dim3(256), 0, st,
                           (const uint32_t *)aos, M, M16, KB, qs, d, m, bad);
    else
        hipLaunchKernelGGL(repack_qw16_kernel<FL_TYPE_Q4_1>, dim3(grid), dim3(256), 0, st,
                           (const uint32_t *)aos, M, M16, KB, qs, d, m, bad);
    return hipGetLastError();
}

hipError_t unpack_from_qw16(int type, const uint32_t *qs, const float *d, const float *m, int M, int K,
                            void *aos, hipStream_t st) {
    const int KB = K / FL_QK;
    const int64_t total = (int64_t)M * KB;
    const int grid = (int)((total + 255) / 256);
    if (type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(unpack_qw16_kernel<FL_TYPE_Q4_0>, dim3(grid), dim3(256), 0, st, qs, d, m, M, KB,
                           (uint32_t *)aos);
    else
        hipLaunchKernelGGL(unpack_qw16_kernel<FL_TYPE_Q4_1>, dim3(grid), dim3(256), 0, st, qs, d, m, M, KB,
                           (uint32_t *)aos);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a4: quantize_row_q8_0.  One thread per 8 consecutive elements, 4 adjacent lanes per 32-block.
//   amax -> d = amax/127, id = amax ? 127/amax : 0 -> q = rint(x*id) (half-even) -> s = d*sum(q)
// ------------------------------------------------------------------------------------------------
enum { Q8_AOS = 0, Q8_QA16 = 1, Q8_QA1 = 2 };

struct Q8Quad {
    uint32_t w_nat[2];   // q0..q3 | q4..q7            (reference order)
    uint32_t w_perm[2];  // q0,q2,q4,q6 | q1,q3,q5,q7  (MFMA / dot4 order)
    float d, s;
};

__device__ __forceinline__ Q8Quad quantize_group8(const float v[8]) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = quad_max_f32(amax);
    const float d = __fdiv_rn(amax, 127.0f);
    const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
    int q[8];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        q[i] = (int)rintf(__fmul_rn(v[i], id));
        sum += q[i];
    }
    sum = quad_sum_i32(sum);
    Q8Quad o;
    o.d = d;
    o.s = __fmul_rn(d, (float)sum);
    auto pk = [](int a, int b, int c, int e) -> uint32_t {
        return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) |
               ((uint32_t)(e & 0xFF) << 24);
    };
    o.w_nat[0] = pk(q[0], q[1], q[2], q[3]);
    o.w_nat[1] = pk(q[4], q[5], q[6], q[7]);
    o.w_perm[0] = pk(q[0], q[2], q[4], q[6]);
    o.w_perm[1] = pk(q[1], q[3], q[5], q[7]);
    return o;
}

template <int LAYOUT>
__global__ void quantize_q8_kernel(const float *__restrict__ x, int ldx, int N, int NP, int K,
                                   int8_t *__restrict__ q, float *__restrict__ d, float *__restrict__ s,
                                   uint32_t *__restrict__ aos, uint16_t *__restrict__ h16) {
    const int gpr = K >> 3;  // 8-element groups per row
    const int KB = K >> 5;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)NP * gpr;  // multiple of 4; whole quads are in or out of range
    const bool live = gid < total;
    const int n = live ? (int)(gid / gpr) : 0;
    const int kg = live ? (int)(gid % gpr) : 0;
    float v[8];
    if (live && n < N) {
        const float4 a = *reinterpret_cast<const float4 *>(x + (int64_t)n * ldx + kg * 8);
        const float4 c = *reinterpret_cast<const float4 *>(x + (int64_t)n * ldx + kg * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    const Q8Quad o = quantize_group8(v);
    if (!live) return;
    const int b = kg >> 2, g = kg & 3;
    if (LAYOUT == Q8_AOS) {
        // block_q8_0 {float d; float s; int8 qs[32]} = 10 words, lib/ggml.c:620-626
        uint32_t *p = aos + ((int64_t)n * KB + b) * 10;
        p[2 + g * 2] = o.w_nat[0];
        p[3 + g * 2] = o.w_nat[1];
        if (g == 0) {
            p[0] = __float_as_uint(o.d);
            p[1] = __float_as_uint(o.s);
        }
    } else if (LAYOUT == Q8_QA16) {
        const int grp = n >> 4, c = n & 15;
        const int64_t cb = ((int64_t)grp * KB + b) * 16 + c;  // (group, block, col)
        uint2 w = make_uint2(o.w_perm[0], o.w_perm[1]);
        *reinterpret_cast<uint2 *>(q + cb * 32 + qw16_pos(c, g) * 8) = w;
        if (g == 0) {
            d[cb] = o.d;
            s[cb] = o.s;
        }
        if (h16) {   // the XH16 copy (q4_layout.h): w_nat holds the group's int8 in element order (0..3 | 4..7)
            auto hb = [](uint32_t wv, int i) -> uint32_t { return (uint32_t)__half_as_ushort(__int2half_rn((int)(int8_t)((wv >> (8 * i)) & 0xFF))); };
            unsigned char *dst = reinterpret_cast<unsigned char *>(h16) + ((((int64_t)(n >> 5) * KB + b) * 2 + (g >> 1)) * 64 + (n & 31)) * 16 + (g & 1) * 8;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(hb(o.w_nat[0], 0) | (hb(o.w_nat[0], 1) << 16), hb(o.w_nat[0], 2) | (hb(o.w_nat[0], 3) << 16));
            *reinterpret_cast<uint2 *>(dst + 512) = make_uint2(hb(o.w_nat[1], 0) | (hb(o.w_nat[1], 1) << 16), hb(o.w_nat[1], 2) | (hb(o.w_nat[1], 3) << 16));
        }
    } else {
        const int64_t vb = (int64_t)n * KB + b;
        uint2 w = make_uint2(o.w_perm[0], o.w_perm[1]);
        *reinterpret_cast<uint2 *>(q + vb * 32 + g * 8) = w;
        if (g == 0) {
            d[vb] = o.d;
            s[vb] = o.s;
        }
    }
}

size_t qact_bytes_q(int N, int K) { return (size_t)fl_roundup(N, 16) * (size_t)K; }
size_t qact_bytes_scale(int N, int K) { return (size_t)fl_roundup(N, 16) * (size_t)(K / FL_QK) * sizeof(float); }

template <int LAYOUT>
static hipError_t launch_quantize(const float *x, int ldx, int N, int NP, int K, int8_t *q, float *d, float *s,
                                  void *aos, hipStream_t st, uint16_t *h16 = nullptr) {
    const int64_t total = (int64_t)NP * (K >> 3);
    const int grid = (int)((total + 255) / 256);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_q8_kernel<LAYOUT>, dim3(grid), dim3(256), 0, st, x, ldx, N, NP, K, q, d, s,
                       (uint32_t *)aos, h16);
    return hipGetLastError();
}

hipError_t quantize_q8_aos(const float *x, int ldx, int N, int K, void *aos, hipStream_t st) {
    return launch_quantize<Q8_AOS>(x, ldx, N, N, K, nullptr, nullptr, nullptr, aos, st);
}
hipError_t quantize_q8_qa16(const float *x, int ldx, int N, int K, const fl_qact &o, hipStream_t st, bool with_h16) {
    return launch_quantize<Q8_QA16>(x, ldx, N, fl_roundup(N, 16), K, o.q, o.d, o.s, nullptr, st, with_h16 ? o.h16 : nullptr);
}
hipError_t quantize_q8_qa1(const float *x, int ldx, int N, int K, const fl_qact &o, hipStream_t st) {
    return launch_quantize<Q8_QA1>(x, ldx, N, N, K, o.q, o.d, o.s, nullptr, st);
}

template <int LAYOUT>
__global__ void export_q8_kernel(const int8_t *__restrict__ q, const float *__restrict__ d,
                                 const float *__restrict__ s, int N, int K, uint32_t *__restrict__ aos) {
    const int gpr = K >> 3, KB = K >> 5;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)N * gpr) return;
    const int n = (int)(gid / gpr), kg = (int)(gid % gpr);
    const int b = kg >> 2, g = kg & 3;
    int64_t sb;
    uint2 w;
    if (LAYOUT == Q8_QA16) {
        const int grp = n >> 4, c = n & 15;
        sb = ((int64_t)grp * KB + b) * 16 + c;
        w = *reinterpret_cast<const uint2 *>(q + sb * 32 + qw16_pos(c, g) * 8);
    } else {
        sb = (int64_t)n * KB + b;
        w = *reinterpret_cast<const uint2 *>(q + sb * 32 + g * 8);
    }
    // undo the even/odd split: w.x = e0,e2,e4,e6  w.y = e1,e3,e5,e7
    auto by = [](uint32_t v, int i) -> uint32_t { return (v >> (8 * i)) & 0xFFu; };
    const uint32_t n0 = by(w.x, 0) | (by(w.y, 0) << 8) | (by(w.x, 1) << 16) | (by(w.y, 1) << 24);
    const uint32_t n1 = by(w.x, 2) | (by(w.y, 2) << 8) | (by(w.x, 3) << 16) | (by(w.y, 3) << 24);
    uint32_t *p = aos + ((int64_t)n * KB + b) * 10;
    p[2 + g * 2] = n0;
    p[3 + g * 2] = n1;
    if (g == 0) {
        p[0] = __float_as_uint(d[sb]);
        p[1] = __float_as_uint(s[sb]);
    }
}

hipError_t export_qa16_to_aos(const fl_qact &in, int N, int K, void *aos, hipStream_t st) {
    const int64_t total = (int64_t)N * (K >> 3);
    hipLaunchKernelGGL(export_q8_kernel<Q8_QA16>, dim3((int)((total + 255) / 256)), dim3(256), 0, st, in.q, in.d,
                       in.s, N, K, (uint32_t *)aos);
    return hipGetLastError();
}
hipError_t export_qa1_to_aos(const fl_qact &in, int N, int K, void *aos, hipStream_t st) {
    const int64_t total = (int64_t)N * (K >> 3);
    hipLaunchKernelGGL(export_q8_kernel<Q8_QA1>, dim3((int)((total + 255) / 256)), dim3(256), 0, st, in.q, in.d,
                       in.s, N, K, (uint32_t *)aos);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a7: dequantize_row_q4_{0,1} on AoS blocks.  One thread per nibble dword (8 weights).
//   Q4_0: (nib-8)*d          lib/ggml.c:1449-1482
//   Q4_1: fma(nib, d, m)     lib/ggml.c:1567-1597 (the reference's gcc build fuses mul+add)
// ------------------------------------------------------------------------------------------------
template <int TYPE>
__global__ void dequantize_aos_kernel(const uint32_t *__restrict__ aos, float *__restrict__ y, int64_t ngroups) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ngroups) return;
    constexpr int WPB = TYPE == FL_TYPE_Q4_0 ? 5 : 6;
    const int64_t blk = gid >> 2;
    const int g = (int)(gid & 3);
    const uint32_t *p = aos + blk * WPB;
    const float d = __uint_as_float(p[0]);
    const float m = TYPE == FL_TYPE_Q4_1 ? __uint_as_float(p[1]) : 0.f;
    const uint32_t v = p[WPB - 4 + g];
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lo = (v >> (8 * j)) & 0xF, hi = (v >> (8 * j + 4)) & 0xF;
        if (TYPE == FL_TYPE_Q4_0) {
            o[2 * j] = __fmul_rn((float)(lo - 8), d);
            o[2 * j + 1] = __fmul_rn((float)(hi - 8), d);
        } else {
            o[2 * j] = __fmaf_rn((float)lo, d, m);
            o[2 * j + 1] = __fmaf_rn((float)hi, d, m);
        }
    }
    float4 *yp = reinterpret_cast<float4 *>(y + gid * 8);
    yp[0] = make_float4(o[0], o[1], o[2], o[3]);
    yp[1] = make_float4(o[4], o[5], o[6], o[7]);
}

hipError_t dequantize_aos(int type, const void *aos, float *y, int64_t k, hipStream_t st) {
    const int64_t ngroups = k >> 3;
    const int grid = (int)((ngroups + 255) / 256);
    if (grid == 0) return hipSuccess;
    if (type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(dequantize_aos_kernel<FL_TYPE_Q4_0>, dim3(grid), dim3(256), 0, st,
                           (const uint32_t *)aos, y, ngroups);
    else
        hipLaunchKernelGGL(dequantize_aos_kernel<FL_TYPE_Q4_1>, dim3(grid), dim3(256), 0, st,
                           (const uint32_t *)aos, y, ngroups);
    return hipGetLastError();
}

// get_rows_q (token embedding): dequantize rows[i] of a QW16 tensor.  lib/ggml.c:8333-8360
template <int TYPE>
__global__ void get_rows_qw16_kernel(const uint32_t *__restrict__ qs, const float *__restrict__ d,
                                     const float *__restrict__ mm, const int *__restrict__ rows, int nrows,
                                     int M, int KB, float *__restrict__ y, int ldy) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = KB * 4;
    if (gid >= (int64_t)nrows * gpr) return;
    const int i = (int)(gid / gpr), kg = (int)(gid % gpr);
    const int b = kg >> 2, g = kg & 3;
    int row = rows[i];
    row = row < 0 ? 0 : (row >= M ? M - 1 : row);
    const int grp = row >> 4, r = row & 15;
    const int64_t src = ((int64_t)grp * KB + b) * 16 + r;
    const uint32_t v = qs[src * 4 + qw16_pos(r, g)];
    float o[8];
    if (TYPE == FL_TYPE_Q4_0) {
        const float dv = d[src] * 16.0f;  // undo the exact /16 of the repack
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lo = ((int)(v << (28 - 8 * j))) >> 28;  // stored nibble is two's-complement (nib-8)
            const int hi = ((int)(v << (24 - 8 * j))) >> 28;
            o[2 * j] = __fmul_rn((float)lo, dv);
            o[2 * j + 1] = __fmul_rn((float)hi, dv);
        }
    } else {
        const float dv = d[src], mv = mm[src];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lo = (v >> (8 * j)) & 0xF, hi = (v >> (8 * j + 4)) & 0xF;
            o[2 * j] = __fmaf_rn((float)lo, dv, mv);
            o[2 * j + 1] = __fmaf_rn((float)hi, dv, mv);
        }
    }
    float4 *yp = reinterpret_cast<float4 *>(y + (int64_t)i * ldy + kg * 8);
    yp[0] = make_float4(o[0], o[1], o[2], o[3]);
    yp[1] = make_float4(o[4], o[5], o[6], o[7]);
}

hipError_t get_rows_qw16(const fl_qtensor &W, const int *rows, int nrows, float *y, int ldy, hipStream_t st) {
    const int64_t total = (int64_t)nrows * W.KB * 4;
    const int grid = (int)((total + 255) / 256);
    if (grid == 0) return hipSuccess;
    if (W.type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(get_rows_qw16_kernel<FL_TYPE_Q4_0>, dim3(grid), dim3(256), 0, st, W.qs, W.d, W.m, rows,
                           nrows, W.M, W.KB, y, ldy);
    else
        hipLaunchKernelGGL(get_rows_qw16_kernel<FL_TYPE_Q4_1>, dim3(grid), dim3(256), 0, st, W.qs, W.d, W.m, rows,
                           nrows, W.M, W.KB, y, ldy);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a5/a6 on AoS operands: the quantize_fns_t::vec_dot_q mirror (one dot, one workgroup).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int TYPE>
__global__ __launch_bounds__(256) void vec_dot_aos_kernel(int nb, float *__restrict__ out,
                                                          const uint32_t *__restrict__ xw,
                                                          const uint32_t *__restrict__ yq) {
    constexpr int WPB = TYPE == FL_TYPE_Q4_0 ? 5 : 6;
    float acc = 0.f;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t *pw = xw + (int64_t)b * WPB;
        const uint32_t *px = yq + (int64_t)b * 10;
        const float dw = __uint_as_float(pw[0]);
        const float dx = __uint_as_float(px[0]);
        int isum = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t v = pw[WPB - 4 + g];
            const uint32_t x0 = px[2 + 2 * g], x1 = px[3 + 2 * g];  // natural order q0..q3 | q4..q7
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lo = (v >> (8 * j)) & 0xF, hi = (v >> (8 * j + 4)) & 0xF;
                const uint32_t xs = j < 2 ? x0 : x1;
                const int q0 = (int)(int8_t)((xs >> (16 * (j & 1))) & 0xFF);
                const int q1 = (int)(int8_t)((xs >> (16 * (j & 1) + 8)) & 0xFF);
                if (TYPE == FL_TYPE_Q4_0) isum += (lo - 8) * q0 + (hi - 8) * q1;
                else isum += lo * q0 + hi * q1;
            }
        }
        acc = __fmaf_rn(__fmul_rn(dw, dx), (float)isum, acc);
        if (TYPE == FL_TYPE_Q4_1) acc = __fmaf_rn(__uint_as_float(pw[1]), __uint_as_float(px[1]), acc);
    }
    __shared__ float part[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1]) + (part[2] + part[3]);
}

hipError_t vec_dot_aos(int type, int n, float *s, const void *x, const void *y, hipStream_t st) {
    if (type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(vec_dot_aos_kernel<FL_TYPE_Q4_0>, dim3(1), dim3(256), 0, st, n / FL_QK, s,
                           (const uint32_t *)x, (const uint32_t *)y);
    else
        hipLaunchKernelGGL(vec_dot_aos_kernel<FL_TYPE_Q4_1>, dim3(1), dim3(256), 0, st, n / FL_QK, s,
                           (const uint32_t *)x, (const uint32_t *)y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a9 decode path (N <= 8): wavefront-dot GEMV, HBM-bound.
//
// One 256-thread workgroup (4 waves) owns one 16-row group of W.  A wave-wide 16-byte load covers
// (16 rows) x (4 consecutive blocks) = 1 KiB contiguous of QW16; lane = row + 16*(block&3).  The four
// waves stride over the block-quads of the row, every lane keeps one f32 partial per column, then
// lanes {r, r+16, r+32, r+48} and the four waves are summed.
// Per lane and block: 12 VALU unpack + NC*(8 v_dot4 + cvt + mul + fma).
// ------------------------------------------------------------------------------------------------
// quantize_row_q8_0 of one 8-element group (4 adjacent lanes = one block) into an LDS copy of the QA1 layout
__device__ __forceinline__ void quantize_group_lds(const float o[8], int kg, int8_t *lq, float *ld_, float *ls_) {
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
    *reinterpret_cast<uint2 *>(lq + kg * 8) = make_uint2(pk(qi[0], qi[2], qi[4], qi[6]), pk(qi[1], qi[3], qi[5], qi[7]));
    if ((kg & 3) == 0) {
        ld_[kg >> 2] = dd;
        ls_[kg >> 2] = __fmul_rn(dd, (float)isum);
    }
}

// Decode prologues (NC = 1): the activation arrives as f32 and the kernel builds its Q8_0 form in LDS itself, AFTER
// its weight loads are in flight -- one launch and one HBM round trip less per matmul; every workgroup redoes the
// small prologue from L2.
//   PRO = 1: rms_norm * weight -> Q8_0 (arithmetic and f64 sum order of rmsnorm_quant_kernel: first 256 threads)
//   PRO = 2: silu(w1 x) * (w3 x) -> Q8_0 (silu_mul_quant_kernel; xf = [w1 x (K) | w3 x (K)], aux = fp16 silu table)
// U = block-quads in flight per wave; the launcher picks U so that NWAVES * U covers the row when it can, i.e. all of
// a workgroup's weight bytes are requested before anything waits.
//   PRO = 3: plain quantize_row_q8_0 of an f32 vector (the activation of the w2 matmul after a PAIR = 1 launch)
// PAIR = 1 (NC = 1): the workgroup owns TWO consecutive 16-row groups -- w1 rows and the same rows of w3 in the woven
// w1|w3 matrix (model.cpp) -- and stores silu(w1 x) * (w3 x) for its 16 features instead of the two dots (aux2 = fp16
// SiLU table): ggml_silu + ggml_mul of lib/llama.cpp:428-431 as the epilogue of the matmul.  Slot u of a wave's U loads
// belongs to group u & 1.
template <int TYPE, int NC, int NWAVES, int PRO, int U, int PAIR>
// (argument order: what the first weight / activation requests need comes first -- the leading 12 dwords of the kernarg segment
//  are preloaded into SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count, build.sh), so those requests go out
//  without waiting for the s_load round trip of the rest)
__global__ __launch_bounds__(64 * NWAVES) void gemv_q4_kernel(int N, int M, int KB, int woven,
                                                      const uint4 *__restrict__ qs, const float *__restrict__ dW,
                                                      const float *__restrict__ xf, const void *__restrict__ aux,
                                                      const float *__restrict__ mW,
                                                      const int8_t *__restrict__ xq, const float *__restrict__ xd,
                                                      const float *__restrict__ xs,
                                                      float *__restrict__ y, int ldy,
                                                      const float *__restrict__ resid, int ldr,
                                                      float *__restrict__ ynorm,
                                                      const uint16_t *__restrict__ aux2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    static_assert(!PAIR || (NC == 1 && U % 2 == 0), "pair mode: single column, even number of load slots");
    constexpr int G2 = PAIR ? 2 : 1;
    constexpr int UQ = PAIR ? U / 2 : U;                       // block-quads per wave and pass (of each group)
    const int grp = blockIdx.x * G2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, bq = lane >> 4;
    const int xswap = ((r >> 3) & 1) * 16;  // rows 8..15 keep k-groups {2,3} first (qw16_pos)
    float acc[G2][NC];
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[g][c] = 0.f;

    const int nquads = (KB + 3) >> 2;
    const int64_t gbase = (int64_t)grp * KB;
    uint4 w[U];
    float dw[U], mw[U];
    bool ok[U];
    auto load = [&](int q0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g2 = PAIR ? (u & 1) : 0, uq = PAIR ? (u >> 1) : u;
            const int b = (q0 + uq * NWAVES) * 4 + bq;
            ok[u] = (q0 + uq * NWAVES) < nquads && b < KB;
            const int64_t idx = (gbase + (int64_t)g2 * KB + (ok[u] ? b : 0)) * 16 + r;
#ifndef FL_GEMV_NO_NT  // streamed once, by one CU: nontemporal loads (1.541 -> 1.446 ms per 7B token, profiles/r03_decode_nt.txt)
            {
                typedef unsigned int nt_v4u __attribute__((ext_vector_type(4)));
                const nt_v4u t = __builtin_nontemporal_load(reinterpret_cast<const nt_v4u *>(&qs[idx]));
                w[u] = make_uint4(t.x, t.y, t.z, t.w);
            }
            dw[u] = __builtin_nontemporal_load(&dW[idx]);
            mw[u] = TYPE == FL_TYPE_Q4_1 ? __builtin_nontemporal_load(&mW[idx]) : 0.f;
#else
            w[u] = qs[idx];
            dw[u] = dW[idx];
            mw[u] = TYPE == FL_TYPE_Q4_1 ? mW[idx] : 0.f;
#endif
        }
    };
    // Vector-memory loads return in order: the small activation loads of the prologue are issued BEFORE the weight
    // stream, or the prologue would sit behind all of it.
    constexpr int MAXIT = 4;
    float v[PRO == 1 ? MAXIT : 1][8];
    float ww[PRO == 1 ? MAXIT : 1][8];
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
    constexpr int SIT = 2;                       // PRO == 2: group-iterations whose loads precede the weight stream
    float sl_[PRO == 2 ? SIT : 1][8], sb_[PRO == 2 ? SIT : 1][8];
    if constexpr (PRO == 2) {
        const uint16_t *silu_tab = static_cast<const uint16_t *>(aux);
        const int F = KB * 32, gpr = F >> 3;
        float sa_[SIT][8];
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int kg = threadIdx.x + it * 64 * NWAVES;
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
            const int kg = threadIdx.x + it * 64 * NWAVES;
            if (kg < gpr) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {   // the table gathers go out now; their results are used after the weight loads
                    const uint16_t hx = __half_as_ushort(__float2half_rn(sa_[it][i]));
                    sl_[it][i] = __half2float(__ushort_as_half(silu_tab[hx]));
                }
            }
        }
    }
    // block-quad q is owned by wave q % NWAVES: every wave streams, whatever K is
    int q0 = wave;
    if (q0 < nquads) load(q0);

    int8_t *lq = reinterpret_cast<int8_t *>(gsm);                   // [KB][32]
    float *ld_ = reinterpret_cast<float *>(gsm + (size_t)KB * 32);  // [KB] d
    float *ls_ = ld_ + KB;                                          // [KB] s
    if constexpr (PRO == 1) {
        __shared__ double sh[4];
        const int E = KB * 32, gpr = E >> 3;
        double sum = 0.0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += (double)__fmul_rn(v[it][i], v[it][i]);
            }
            sum = wave_sum_f64(sum);                          // same order as block_sum_f64 (eval_kernels.hip)
            if (lane == 0) sh[wave] = sum;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            double t = 0.0;
            for (int i = 0; i < 4; ++i) t += sh[i];
            const float mean = (float)(t / (double)E);
            const float scale = __fdiv_rn(1.0f, sqrtf(mean + 1e-6f));
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int kg = threadIdx.x + it * 256;
                if (kg >= gpr) continue;   // whole quads leave together
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = __fmul_rn(ww[it][i], __fmul_rn(v[it][i], scale));
                if (ynorm && grp == 0) {
                    float4 *yp = reinterpret_cast<float4 *>(ynorm + kg * 8);
                    yp[0] = make_float4(o[0], o[1], o[2], o[3]);
                    yp[1] = make_float4(o[4], o[5], o[6], o[7]);
                }
                quantize_group_lds(o, kg, lq, ld_, ls_);
            }
        }
        __syncthreads();
    } else if constexpr (PRO == 2) {
        const uint16_t *silu_tab = static_cast<const uint16_t *>(aux);
        const int F = KB * 32, gpr = F >> 3;
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int kg = threadIdx.x + it * 64 * NWAVES;
            if (kg >= gpr) continue;                                   // gpr % 4 == 0: quads stay together
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = __fmul_rn(sl_[it][i], sb_[it][i]);
            quantize_group_lds(o, kg, lq, ld_, ls_);
        }
        for (int kg = threadIdx.x + SIT * 64 * NWAVES; kg < gpr; kg += 64 * NWAVES) {   // very wide rows: the rest
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
            quantize_group_lds(o, kg, lq, ld_, ls_);
        }
        __syncthreads();
    } else if constexpr (PRO == 3) {
        const int gpr = KB * 4;
        for (int kg = threadIdx.x; kg < gpr; kg += 64 * NWAVES) {   // gpr % 4 == 0: quads stay together
            const float4 a0 = *reinterpret_cast<const float4 *>(xf + kg * 8), a1 = *reinterpret_cast<const float4 *>(xf + kg * 8 + 4);
            const float o[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            quantize_group_lds(o, kg, lq, ld_, ls_);
        }
        __syncthreads();
    }

    while (q0 < nquads) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int g2 = PAIR ? (u & 1) : 0, uq = PAIR ? (u >> 1) : u;
            const int b = (q0 + uq * NWAVES) * 4 + bq;
            uint32_t lo[4], hi[4];
            unpack_nibbles<TYPE>(w[u].x, lo[0], hi[0]);
            unpack_nibbles<TYPE>(w[u].y, lo[1], hi[1]);
            unpack_nibbles<TYPE>(w[u].z, lo[2], hi[2]);
            unpack_nibbles<TYPE>(w[u].w, lo[3], hi[3]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int n = c < N ? c : 0;
                uint4 xa, xc;
                float dx, sx;
                if constexpr (PRO != 0) {
                    xa = *reinterpret_cast<const uint4 *>(lq + b * 32 + xswap);
                    xc = *reinterpret_cast<const uint4 *>(lq + b * 32 + (16 - xswap));
                    dx = ld_[b];
                    sx = ls_[b];
                } else {
                    const int8_t *xb = xq + ((int64_t)n * KB + b) * 32;
                    xa = *reinterpret_cast<const uint4 *>(xb + xswap);
                    xc = *reinterpret_cast<const uint4 *>(xb + (16 - xswap));
                    dx = xd[(int64_t)n * KB + b];
                    sx = TYPE == FL_TYPE_Q4_1 ? xs[(int64_t)n * KB + b] : 0.f;
                }
                int isum = 0;
                isum = dot8(lo[0], hi[0], xa.x, xa.y, isum);
                isum = dot8(lo[1], hi[1], xa.z, xa.w, isum);
                isum = dot8(lo[2], hi[2], xc.x, xc.y, isum);
                isum = dot8(lo[3], hi[3], xc.z, xc.w, isum);
                acc[g2][c] = __fmaf_rn(__fmul_rn(dw[u], dx), (float)isum, acc[g2][c]);
                if (TYPE == FL_TYPE_Q4_1) acc[g2][c] = __fmaf_rn(mw[u], sx, acc[g2][c]);
            }
        }
        q0 += NWAVES * UQ;
        if (q0 < nquads) load(q0);
    }
    __shared__ float part[NWAVES][G2 * NC][16];
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float v = acc[g][c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) part[wave][g * NC + c][lane] = v;
        }
    __syncthreads();
    auto total = [&](int slot, int rr) -> float {
        float t[NWAVES];
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) t[w] = part[w][slot][rr];
#pragma unroll
        for (int st = 1; st < NWAVES; st <<= 1)           // fixed pairwise tree: deterministic
#pragma unroll
            for (int w = 0; w + st < NWAVES; w += 2 * st) t[w] += t[w + st];
        return t[0];
    };
    if constexpr (PAIR) {
        if (threadIdx.x < 16) {
            const int rr = threadIdx.x, f = blockIdx.x * 16 + rr;          // feature index; rows 2*16*blockIdx.x + {rr, 16+rr}
            if (grp * 16 + 16 + rr < M) {
                const float y1 = total(0, rr), y3 = total(1, rr);
                const uint16_t hx = __half_as_ushort(__float2half_rn(y1));                // GGML_FP32_TO_FP16
                const float sl = __half2float(__ushort_as_half(aux2[hx]));               // table_silu_f16
                y[f] = __fmul_rn(sl, y3);                                                 // ggml_mul(silu, tmp)
            }
        }
    } else if (threadIdx.x < 16 * NC) {
        const int c = threadIdx.x >> 4, rr = threadIdx.x & 15;
        const int row = grp * 16 + rr;
        if (c < N && row < M) {
            float v = total(c, rr);
            if (resid) v += resid[(int64_t)c * ldr + row];
            y[(int64_t)c * ldy + row] = v;
        }
    }
}

int g_gemv_force_waves = 0;  // debug / tuning hook (fl_debug_set(1, n))
static inline int gemv_waves(int groups, int KB = 0) {
    if (g_gemv_force_waves == 4 || g_gemv_force_waves == 8 || g_gemv_force_waves == 16) return g_gemv_force_waves;
    if (groups >= 640) return 4;
    if (groups >= 512) return 8;
    return KB >= 256 ? 8 : 16;          // few row groups: 16 waves each, unless the rows are long enough to feed 8 deeply
}

// single-token launches: (NWAVES by M, U by K) so that one pass covers the row when NWAVES * 8 >= K/128
template <int TYPE, int PRO, int PAIR = 0>
static hipError_t launch_gemv1(const fl_qtensor &W, const fl_qact *xq, float *y, hipStream_t st, const float *resid,
                               const float *xf, const void *aux, float *ynorm, int woven = 0, const uint16_t *aux2 = nullptr) {
    const int groups = W.M16 / 16 / (PAIR ? 2 : 1);             // workgroups
    const dim3 grid(groups);
    const uint4 *qs = reinterpret_cast<const uint4 *>(W.qs);
    const int nw = gemv_waves(groups, W.KB);
    const int nquads = (W.KB + 3) / 4, per_wave = ((nquads + nw - 1) / nw) * (PAIR ? 2 : 1);   // load slots per wave
    const int u = per_wave <= 2 ? 2 : per_wave <= 4 ? 4 : per_wave <= 6 ? 6 : 8;
    const size_t lds = PRO ? (size_t)W.KB * 40 : 0;
#define FL_GEMV(NW, UU)                                                                                              \
    hipLaunchKernelGGL((gemv_q4_kernel<TYPE, 1, NW, PRO, UU, PAIR>), grid, dim3(64 * NW), lds, st, 1, W.M, W.KB, woven, qs, W.d, \
                       xf, aux, W.m, xq ? xq->q : nullptr, xq ? xq->d : nullptr, xq ? xq->s : nullptr, y, 0, resid, 0,        \
                       ynorm, aux2)
#define FL_GEMV_U(NW)                   \
    do {                                \
        if (u == 2) FL_GEMV(NW, 2);     \
        else if (u == 4) FL_GEMV(NW, 4);\
        else if (u == 6) FL_GEMV(NW, 6);\
        else FL_GEMV(NW, 8);            \
    } while (0)
    if (nw == 4) FL_GEMV_U(4);
    else if (nw == 8) FL_GEMV_U(8);
    else FL_GEMV_U(16);
#undef FL_GEMV_U
#undef FL_GEMV
    return hipGetLastError();
}

template <int TYPE>
static hipError_t launch_gemv(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                              const float *resid, int ldr) {
    if (N == 1) return launch_gemv1<TYPE, 0>(W, &xq, y, st, resid, nullptr, nullptr, nullptr);
    const dim3 grid(W.M16 / 16);
    const uint4 *qs = reinterpret_cast<const uint4 *>(W.qs);
    // HBM-bound: what matters is bytes in flight per CU.  One workgroup streams one 16-row group; small M gets
    // more waves per group (each takes every NWAVES-th block-quad) so that >= ~16 waves per CU are loading.
    const int nw = gemv_waves(W.M16 / 16);
#define FL_GEMV(NC, NW, UU)                                                                                             \
    hipLaunchKernelGGL((gemv_q4_kernel<TYPE, NC, NW, 0, UU, 0>), grid, dim3(64 * NW), 0, st, N, W.M, W.KB, 0, qs, W.d,      \
                       static_cast<const float *>(nullptr), static_cast<const void *>(nullptr), W.m, xq.q, xq.d, xq.s, y, ldy,  \
                       resid, ldr, static_cast<float *>(nullptr), static_cast<const uint16_t *>(nullptr))
#define FL_GEMV_NW(NC, UU)                       \
    do {                                         \
        if (nw == 4) FL_GEMV(NC, 4, UU);         \
        else if (nw == 8) FL_GEMV(NC, 8, UU);    \
        else FL_GEMV(NC, 16, UU);                \
    } while (0)
    if (N == 2) FL_GEMV_NW(2, 4);
    else if (N <= 4) FL_GEMV_NW(4, 2);
    else FL_GEMV_NW(8, 2);
#undef FL_GEMV_NW
#undef FL_GEMV
    return hipGetLastError();
}

// y[M] = W . Q8_0(norm_w * rms_norm(x))   -- decode: rms_norm + mul + quantize_row_q8_0 + mul_mat in one launch
hipError_t gemv_q4_norm(const fl_qtensor &W, const float *x, const float *norm_w, float *ynorm, float *y, hipStream_t st) {
    if (W.K % 32 != 0 || W.K > 8192) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_gemv1<FL_TYPE_Q4_0, 1>(W, nullptr, y, st, nullptr, x, norm_w, ynorm)
                                  : launch_gemv1<FL_TYPE_Q4_1, 1>(W, nullptr, y, st, nullptr, x, norm_w, ynorm);
}

// y[M] = W . Q8_0(silu(h13[0:K]) * h13[K:2K]) (+ resid)   -- decode feed-forward down projection in one launch
hipError_t gemv_q4_silu(const fl_qtensor &W, const float *h13, const uint16_t *silu_tab, float *y, const float *resid,
                        hipStream_t st, bool woven) {
    if (W.K % 32 != 0 || W.K > 32768) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_gemv1<FL_TYPE_Q4_0, 2>(W, nullptr, y, st, resid, h13, silu_tab, nullptr, woven ? 1 : 0)
                                  : launch_gemv1<FL_TYPE_Q4_1, 2>(W, nullptr, y, st, resid, h13, silu_tab, nullptr, woven ? 1 : 0);
}

// act[n_ff] = silu(w1 . q) * (w3 . q),  q = Q8_0(norm_w * rms_norm(x));  W = w1|w3 woven by 16-row groups (W.M = 2 n_ff)
hipError_t gemv_q4_norm_silu(const fl_qtensor &W, const float *x, const float *norm_w, const uint16_t *silu_tab, float *act,
                             hipStream_t st) {
    if (W.K % 32 != 0 || W.K > 8192 || W.M % 32 != 0) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_gemv1<FL_TYPE_Q4_0, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab)
                                  : launch_gemv1<FL_TYPE_Q4_1, 1, 1>(W, nullptr, act, st, nullptr, x, norm_w, nullptr, 0, silu_tab);
}

// y[M] = W . Q8_0(x) (+ resid) for an f32 vector x: quantize_row_q8_0 in the prologue
hipError_t gemv_q4_quant(const fl_qtensor &W, const float *x, float *y, const float *resid, hipStream_t st) {
    if (W.K % 32 != 0 || W.K > 32768) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_gemv1<FL_TYPE_Q4_0, 3>(W, nullptr, y, st, resid, x, nullptr, nullptr)
                                  : launch_gemv1<FL_TYPE_Q4_1, 3>(W, nullptr, y, st, resid, x, nullptr, nullptr);
}

hipError_t gemv_q4(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st,
                   const float *resid, int ldr) {
    if (N < 1 || N > 8) return hipErrorInvalidValue;
    return W.type == FL_TYPE_Q4_0 ? launch_gemv<FL_TYPE_Q4_0>(W, xq, N, y, ldy, st, resid, ldr)
                                  : launch_gemv<FL_TYPE_Q4_1>(W, xq, N, y, ldy, st, resid, ldr);
}

// ------------------------------------------------------------------------------------------------
// a9, straightforward device version (one thread per output) -- used by tests to cross-check the
// MFMA kernel on the device itself.  QW16 x QA16.
// ------------------------------------------------------------------------------------------------
template <int TYPE>
__global__ void gemm_q4_naive_kernel(const uint32_t *__restrict__ qs, const float *__restrict__ dW,
                                     const float *__restrict__ mW, const int8_t *__restrict__ xq,
                                     const float *__restrict__ xd, const float *__restrict__ xs, int N, int M,
                                     int KB, float *__restrict__ y, int ldy) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (row >= M || n >= N) return;
    const int grp = row >> 4, r = row & 15, ng = n >> 4, c = n & 15;
    float acc = 0.f;
    for (int b = 0; b < KB; ++b) {
        const int64_t wi = ((int64_t)grp * KB + b) * 16 + r;
        const int64_t xi = ((int64_t)ng * KB + b) * 16 + c;
        int isum = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t lo, hi;
            unpack_nibbles<TYPE>(qs[wi * 4 + qw16_pos(r, g)], lo, hi);
            const uint2 xv = *reinterpret_cast<const uint2 *>(xq + xi * 32 + qw16_pos(c, g) * 8);
            isum = dot8(lo, hi, xv.x, xv.y, isum);
        }
        acc = __fmaf_rn(__fmul_rn(dW[wi], xd[xi]), (float)isum, acc);
        if (TYPE == FL_TYPE_Q4_1) acc = __fmaf_rn(mW[wi], xs[xi], acc);
    }
    y[(int64_t)n * ldy + row] = acc;
}

hipError_t gemm_q4_naive(const fl_qtensor &W, const fl_qact &xq, int N, float *y, int ldy, hipStream_t st) {
    const dim3 grid((W.M + 255) / 256, N), block(256);
    if (W.type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(gemm_q4_naive_kernel<FL_TYPE_Q4_0>, grid, block, 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s,
                           N, W.M, W.KB, y, ldy);
    else
        hipLaunchKernelGGL(gemm_q4_naive_kernel<FL_TYPE_Q4_1>, grid, block, 0, st, W.qs, W.d, W.m, xq.q, xq.d, xq.s,
                           N, W.M, W.KB, y, ldy);
    return hipGetLastError();
}

}  // namespace fl
