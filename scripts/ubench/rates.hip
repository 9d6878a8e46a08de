// rates.hip -- instruction-rate microbenchmarks that decide the GEMM epilogue design (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 rates.hip -o rates ; run: ./rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define ITERS 32768

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int n) {
    const long long t_begin = clock64();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    long fa = threadIdx.x * 0x0101010101010101L, fb = 0x0102030405060708L;
    v4i acc0 = {0,0,0,0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    v4f f0 = {0,0,0,0}, f1 = f0, f2 = f0, f3 = f0;
    v8s ha = {1,2,3,4,5,6,7,8}, hb = {8,7,6,5,4,3,2,1};
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {  // 4 independent v_pk_fma_f32 (8 elements)
            v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            a0 = p0[0]; a1 = p0[1]; a2 = p1[0]; a3 = p1[1]; a4 = p2[0]; a5 = p2[1]; a6 = p3[0]; a7 = p3[1];
        } else if (MODE == 2) {  // 8 v_cvt_f32_i32
            asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n"
                         "v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));
        } else if (MODE == 3) {  // 4 independent mfma i32 16x16x32 i8
            acc0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, fb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, fb, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, fb, acc3, 0, 0, 0);
        } else if (MODE == 4) {  // 4 independent mfma f32 16x16x32 bf16
            f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f2, 0, 0, 0);
            f3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f3, 0, 0, 0);
        } else if (MODE == 5) {  // i8 mfma with zero C each time + 12 VALU epilogue, 2-deep pipelined by hand
            v4i z = {0,0,0,0};
            v4i r0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, fb, z, 0, 0, 0);
            v4i r1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fb, fa, z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { f0[r] = __builtin_fmaf((float)r0[r], b * f1[r], f0[r]); }
#pragma unroll
            for (int r = 0; r < 4; ++r) { f2[r] = __builtin_fmaf((float)r1[r], c * f3[r], f2[r]); }
        } else if (MODE == 6) {  // bf16 mfma (f32 out) + 8 VALU epilogue
            v4f z = {0,0,0,0};
            v4f r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, z, 0, 0, 0);
            v4f r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, ha, z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { f0[r] = __builtin_fmaf(r0[r], b * f1[r], f0[r]); }
#pragma unroll
            for (int r = 0; r < 4; ++r) { f2[r] = __builtin_fmaf(r1[r], c * f3[r], f2[r]); }
        } else if (MODE == 7) {  // 4 independent mfma i32 16x16x64 i8 (gfx950 double-rate form)
            v4i wa = {(int)fa, (int)fb, (int)fa, (int)fb}, wb = {(int)fb, (int)fa, 3, 4};
            acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, wb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, wb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, wb, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, wb, acc3, 0, 0, 0);
        } else if (MODE == 8) {  // 8 plain v_mul + 8 plain v_fma... measure mixed scalar VALU: 4 cvt, 4 mul, 4 fma
            asm volatile("v_cvt_f32_i32 %0, %8\n v_cvt_f32_i32 %1, %9\n v_cvt_f32_i32 %2, %10\n v_cvt_f32_i32 %3, %11\n"
                         "v_mul_f32 %4, %4, %12\n v_mul_f32 %5, %5, %12\n v_mul_f32 %6, %6, %12\n v_mul_f32 %7, %7, %12\n"
                         "v_fma_f32 %0, %0, %4, %13\n v_fma_f32 %1, %1, %5, %13\n v_fma_f32 %2, %2, %6, %13\n v_fma_f32 %3, %3, %7, %13\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(b), "v"(c));
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[1] + acc2[2] + acc3[3] + f0[0] + f1[1] + f2[2] + f3[3];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - t_begin;
}

static double g_cycles = 0;
template <int MODE>
double run(int blocks_per_cu, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g(256 * blocks_per_cu), b(256);
    static long long *cyc = nullptr;
    if (!cyc) hipMalloc(&cyc, 8 * 4 * 256 * 8);
    hipLaunchKernelGGL(k<MODE>, g, b, 0, 0, d, cyc, ITERS);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, g, b, 0, 0, d, cyc, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(4 * 256 * blocks_per_cu);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    g_cycles = avg;
    return ms * 1e-3;
}

int main() {
    float *d; hipMalloc(&d, 1024);
    const char *names[] = {"v_fma_f32 x8", "v_pk_fma_f32 x4 (8 elts)", "v_cvt_f32_i32 x8", "mfma_i32_16x16x32_i8 x4",
                           "mfma_f32_16x16x32_bf16 x4", "i8 mfma x2 + 12-op epilogue x2", "bf16 mfma x2 + 8-op epilogue x2",
                           "mfma_i32_16x16x64_i8 x4", "4cvt+4mul+4fma scalar"};
    for (int bpc : {1, 2, 4}) {
        printf("--- %d workgroups (256 thr) per CU => %d wave(s)/SIMD\n", bpc, bpc);
        double t[9], c[9];
        t[0] = run<0>(bpc, d); c[0] = g_cycles; t[1] = run<1>(bpc, d); c[1] = g_cycles; t[2] = run<2>(bpc, d); c[2] = g_cycles;
        t[3] = run<3>(bpc, d); c[3] = g_cycles; t[4] = run<4>(bpc, d); c[4] = g_cycles; t[5] = run<5>(bpc, d); c[5] = g_cycles;
        t[6] = run<6>(bpc, d); c[6] = g_cycles; t[7] = run<7>(bpc, d); c[7] = g_cycles; t[8] = run<8>(bpc, d); c[8] = g_cycles;
        const double waves = 256.0 * bpc * 4;  // total waves
        const double insts[] = {8, 4, 8, 4, 4, 2, 2, 4, 12};
        for (int m = 0; m < 9; ++m) {
            // cycles per instruction per SIMD at 2.4 GHz: time * 2.4e9 / (ITERS * insts * waves_per_simd)
            const double cyc = c[m] / ((double)ITERS * insts[m] * bpc);
            printf("%-34s %8.3f ms   %6.2f cyc/inst/SIMD (s_memtime)   clock %.2f GHz\n", names[m], t[m] * 1e3, cyc, c[m] / t[m] * 1e-9);
        }
        (void)waves;
    }
    return 0;
}
