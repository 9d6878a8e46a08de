// fp6_convert.hip -- the fp6 (E2M3) operand copies of the prefill path (q4_layout.h, "F6 copies"): QW16 -> QW16F6 once per
// weight tensor (at load and after a LoRA merge), QA16 -> QA16F6 once per quantized activation batch.  Pure re-encoding of
// integers that are already final (nibbles of ggml's block_q4_0/1, int8 of quantize_row_q8_0: /root/reference/lib/ggml.c:590-626,
// :1341-1403): nothing here rounds.  HBM-bound byte work: one thread per (row | column, block), 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "q4_device.h"
#include "q4_kernels.h"

namespace fl {

// E2M3 code of x/2 for an integer x in [-15, 15]
__device__ __forceinline__ uint32_t fp6_half_code(int x) {
    const uint32_t m = (uint32_t)(x < 0 ? -x : x);
    const uint32_t c = m < 4 ? m << 2 : m < 8 ? 8 + 2 * m : 16 + m;
    return c | (x < 0 ? 32u : 0u);
}

// 32 codes -> 24 bytes, code k at bits [6k, 6k + 6)
struct F6Pack {
    uint64_t w[3] = {0, 0, 0};
    __device__ __forceinline__ void put(int k, uint32_t code) {
        const int bit = 6 * k, word = bit >> 6, sh = bit & 63;
        w[word] |= (uint64_t)code << sh;
        if (sh > 58) w[word + 1] |= (uint64_t)code >> (64 - sh);
    }
};

template <int TYPE>
__global__ __launch_bounds__(256) void qw16_to_f6_kernel(const uint4 *__restrict__ qs, uint2 *__restrict__ f6, int64_t n_rows /* M16/16 * KB * 16 */) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (group, block, row)
    if (u >= n_rows) return;
    const int row = (int)(u & 15);
    const uint4 raw = qs[u];
    const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
    F6Pack pk;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int g = p ^ (((row >> 3) & 1) << 1);                    // logical dword stored at position p (qw16_pos is an involution)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t b = (dw[p] >> (8 * j)) & 0xFF;
            int lo = (int)(b & 15), hi = (int)(b >> 4);
            if (TYPE == FL_TYPE_Q4_0) {                               // stored nibble = 4-bit two's complement of nib - 8
                lo = (lo ^ 8) - 8;
                hi = (hi ^ 8) - 8;
            }
            // positions inside the block as QA16 has them: 8g + t, t = 0..3 the even elements (low nibbles), 4..7 the odd ones
            if (g == 0) { pk.put(j, fp6_half_code(lo)); pk.put(4 + j, fp6_half_code(hi)); }
            if (g == 1) { pk.put(8 + j, fp6_half_code(lo)); pk.put(12 + j, fp6_half_code(hi)); }
            if (g == 2) { pk.put(16 + j, fp6_half_code(lo)); pk.put(20 + j, fp6_half_code(hi)); }
            if (g == 3) { pk.put(24 + j, fp6_half_code(lo)); pk.put(28 + j, fp6_half_code(hi)); }
        }
    }
    uint2 *dst = f6 + u * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[i] = make_uint2((uint32_t)pk.w[i], (uint32_t)(pk.w[i] >> 32));
}

hipError_t qw16_to_f6(const fl_qtensor &W, uint8_t *f6, hipStream_t st) {
    const int64_t n = (int64_t)W.M16 * W.KB;
    if (n == 0) return hipSuccess;
    const int64_t nb = (n + 255) / 256;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    if (W.type == FL_TYPE_Q4_0)
        hipLaunchKernelGGL(qw16_to_f6_kernel<FL_TYPE_Q4_0>, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const uint4 *>(W.qs),
                           reinterpret_cast<uint2 *>(f6), n);
    else
        hipLaunchKernelGGL(qw16_to_f6_kernel<FL_TYPE_Q4_1>, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const uint4 *>(W.qs),
                           reinterpret_cast<uint2 *>(f6), n);
    return hipGetLastError();
}

// the two planes of 32 int8 (positions as stored: slot p of 8 bytes = logical group p ^ (((col >> 3) & 1) << 1))
__device__ __forceinline__ void qa_block_to_f6(const uint4 &r0, const uint4 &r1, int col, F6Pack &hi, F6Pack &lo) {
    const uint32_t dw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int g = p ^ (((col >> 3) & 1) << 1);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = (int)(int8_t)((dw[2 * p + (t >> 2)] >> (8 * (t & 3))) & 0xFF);
            const uint32_t ch = fp6_half_code(q >> 4), cl = fp6_half_code(q & 15);
            if (g == 0) { hi.put(t, ch); lo.put(t, cl); }
            if (g == 1) { hi.put(8 + t, ch); lo.put(8 + t, cl); }
            if (g == 2) { hi.put(16 + t, ch); lo.put(16 + t, cl); }
            if (g == 3) { hi.put(24 + t, ch); lo.put(24 + t, cl); }
        }
    }
}

__global__ __launch_bounds__(256) void qa16_to_f6_kernel(const uint4 *__restrict__ q, uint4 *__restrict__ q6, int64_t n_cols /* N16/16 * KB * 16 */, int KB) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (column group, block, column)
    if (u >= n_cols) return;
    const int col = (int)(u & 15);
    const int grp = (int)((u >> 4) / KB);
    F6Pack hi, lo;
    qa_block_to_f6(q[2 * u], q[2 * u + 1], col, hi, lo);
    unsigned char *blk = reinterpret_cast<unsigned char *>(q6) + (u >> 4) * 768;
    const int sx = (col >> 3) & 1, sy = grp & 1;
    uint4 *X = reinterpret_cast<uint4 *>(blk + col * 32);
    uint2 *Y = reinterpret_cast<uint2 *>(blk + 512 + col * 16);
    X[0 ^ sx] = make_uint4((uint32_t)hi.w[0], (uint32_t)(hi.w[0] >> 32), (uint32_t)hi.w[1], (uint32_t)(hi.w[1] >> 32));
    X[1 ^ sx] = make_uint4((uint32_t)lo.w[0], (uint32_t)(lo.w[0] >> 32), (uint32_t)lo.w[1], (uint32_t)(lo.w[1] >> 32));
    Y[0 ^ sy] = make_uint2((uint32_t)hi.w[2], (uint32_t)(hi.w[2] >> 32));
    Y[1 ^ sy] = make_uint2((uint32_t)lo.w[2], (uint32_t)(lo.w[2] >> 32));
}

hipError_t qa16_to_f6(const fl_qact &xq, int N, hipStream_t st) {
    if (!xq.q6) return hipErrorInvalidValue;
    const int64_t n = (int64_t)fl_roundup(N, 16) * xq.KB;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qa16_to_f6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(xq.q),
                       reinterpret_cast<uint4 *>(xq.q6), n, xq.KB);
    return hipGetLastError();
}

}  // namespace fl
