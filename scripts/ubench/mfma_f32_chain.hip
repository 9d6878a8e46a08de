// mfma_f32_chain.hip -- latency of a DEPENDENT chain of v_mfma_f32_32x32x2_f32 (the attention kernels accumulate one
// output tile through such a chain) versus two interleaved chains.  build: hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
             "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"
#define M1 "v_mfma_f32_32x32x2_f32 v[0:15], v32, v33, v[0:15]\n"
#define M2 "v_mfma_f32_32x32x2_f32 v[16:31], v34, v35, v[16:31]\n"
template <int MODE>
__global__ void k(float *out, int n) {
    asm volatile("v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n" ::: CLOB);
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) asm volatile(M1 M1 M1 M1 M1 M1 M1 M1 ::: CLOB);                 // 8 dependent
        else asm volatile(M1 M2 M1 M2 M1 M2 M1 M2 ::: CLOB);                           // 2 chains of 4
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v16" : "=v"(r)::CLOB);
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int MODE> static void run(const char *name, float *out) {
    const int n = 20000;
    printf("%-34s", name);
    for (int wps = 1; wps <= 2; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d wave/SIMD: %6.1f ns per MFMA per SIMD", wps, ms * 1e6 / n / 8 / wps);
    }
    printf("\n");
}
int main() { float *out; hipMalloc(&out, 4096); run<0>("8 dependent 32x32x2 f32", out); run<1>("2 interleaved chains", out); return 0; }
