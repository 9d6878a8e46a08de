// q4_device.h -- device-side helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "q4_layout.h"

namespace fl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int qw16_pos(int r, int g) { return g ^ (((r >> 3) & 1) << 1); }

// unpack one stored nibble dword (8 weights of k-group g) into two int8x4 dwords:
//   lo = elements 0,2,4,6   hi = elements 1,3,5,7 of the group
// Q4_0: values are 16*(nib-8) (stored nibbles are nib^8, see q4_layout.h); Q4_1: values are nib.
template <int TYPE>
__device__ __forceinline__ void unpack_nibbles(uint32_t v, uint32_t &lo, uint32_t &hi) {
    if (TYPE == FL_TYPE_Q4_0) {
        lo = (v << 4) & 0xF0F0F0F0u;
        hi = v & 0xF0F0F0F0u;
    } else {
        lo = v & 0x0F0F0F0Fu;
        hi = (v >> 4) & 0x0F0F0F0Fu;
    }
}

__device__ __forceinline__ int dot8(uint32_t wlo, uint32_t whi, uint32_t xlo, uint32_t xhi, int acc) {
    acc = __builtin_amdgcn_sdot4((int)wlo, (int)xlo, acc, false);
    acc = __builtin_amdgcn_sdot4((int)whi, (int)xhi, acc, false);
    return acc;
}


// ---- cross-lane reductions on DPP (ds_bpermute, which __shfl_xor compiles to, costs an LDS round trip per step) ----
// quad_perm xor 1 / xor 2, row_half_mirror (lane i <-> 7-i of each 8) and row_mirror (i <-> 15-i of each 16) pair every
// lane with one that holds the other half of the running result, exactly like the xor butterfly.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)b), hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

__device__ __forceinline__ float quad_max_f32(float m) {      // over 4 adjacent lanes
    m = fmaxf(m, dpp_f32<DPP_XOR1>(m));
    return fmaxf(m, dpp_f32<DPP_XOR2>(m));
}
__device__ __forceinline__ int quad_sum_i32(int s) {
    s += dpp_i32<DPP_XOR1>(s);
    return s + dpp_i32<DPP_XOR2>(s);
}
__device__ __forceinline__ float group8_sum_f32(float a) {    // over 8 adjacent lanes: ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7))
    a += dpp_f32<DPP_XOR1>(a);
    a += dpp_f32<DPP_XOR2>(a);
    return a + dpp_f32<DPP_HALF_MIRROR>(a);
}
__device__ __forceinline__ float wave_max_f32(float m) {
    m = fmaxf(m, dpp_f32<DPP_XOR1>(m));
    m = fmaxf(m, dpp_f32<DPP_XOR2>(m));
    m = fmaxf(m, dpp_f32<DPP_HALF_MIRROR>(m));
    m = fmaxf(m, dpp_f32<DPP_MIRROR>(m));
    m = fmaxf(m, __shfl_xor(m, 16));
    return fmaxf(m, __shfl_xor(m, 32));
}
__device__ __forceinline__ double wave_sum_f64(double s) {
    s += dpp_f64<DPP_XOR1>(s);
    s += dpp_f64<DPP_XOR2>(s);
    s += dpp_f64<DPP_HALF_MIRROR>(s);
    s += dpp_f64<DPP_MIRROR>(s);
    s += __shfl_xor(s, 16);
    return s + __shfl_xor(s, 32);
}

// ------------------------------------------------------------------------------------------------
// shared: quantize one 8-element group (4 adjacent lanes = one Q8_0 block) and store it
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quantize_store_group(const float v[8], int n, int kg, int KB, int layout,
                                                     int8_t *__restrict__ q, float *__restrict__ d,
                                                     float *__restrict__ s, uint16_t *__restrict__ h16 = nullptr) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = quad_max_f32(amax);
    const float dd = __fdiv_rn(amax, 127.0f);
    const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
    int qi[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qi[i] = (int)rintf(__fmul_rn(v[i], id));
        sum += qi[i];
    }
    sum = quad_sum_i32(sum);
    auto pk = [](int a, int b, int c, int e) -> uint32_t {
        return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) |
               ((uint32_t)(e & 0xFF) << 24);
    };
    const uint2 w = make_uint2(pk(qi[0], qi[2], qi[4], qi[6]), pk(qi[1], qi[3], qi[5], qi[7]));
    const int b = kg >> 2, g = kg & 3;
    if (layout == 16) {
        const int grp = n >> 4, c = n & 15;
        const int64_t cb = ((int64_t)grp * KB + b) * 16 + c;
        *reinterpret_cast<uint2 *>(q + cb * 32 + qw16_pos(c, g) * 8) = w;
        if (g == 0) {
            d[cb] = dd;
            s[cb] = __fmul_rn(dd, (float)sum);
        }
        if (h16) {   // the XH16 copy (q4_layout.h): k-group g = MFMA step g, elements 0..3 in lane (n & 31), elements 4..7 in lane + 32
            auto hb = [](int x) -> uint32_t { return (uint32_t)__half_as_ushort(__int2half_rn(x)); };
            unsigned char *dst = reinterpret_cast<unsigned char *>(h16) + ((((int64_t)(n >> 5) * KB + b) * 2 + (g >> 1)) * 64 + (n & 31)) * 16 + (g & 1) * 8;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(hb(qi[0]) | (hb(qi[1]) << 16), hb(qi[2]) | (hb(qi[3]) << 16));
            *reinterpret_cast<uint2 *>(dst + 512) = make_uint2(hb(qi[4]) | (hb(qi[5]) << 16), hb(qi[6]) | (hb(qi[7]) << 16));
        }
    } else {
        const int64_t vb = (int64_t)n * KB + b;
        *reinterpret_cast<uint2 *>(q + vb * 32 + g * 8) = w;
        if (g == 0) {
            d[vb] = dd;
            s[vb] = __fmul_rn(dd, (float)sum);
        }
    }
}

}  // namespace fl
