// q4_device.h -- device-side helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
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


}  // namespace fl
