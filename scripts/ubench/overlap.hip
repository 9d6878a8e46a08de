// overlap.hip -- would "attention + wo matmul in ONE launch" pay on MI355X?  Stand-ins with the real sizes:
//   A  32 workgroups x 512 threads: a dependent chain of memory round trips (~ the latency chain of decode_attention),
//      then 128 floats of the activation vector each
//   B  256 workgroups x 1024 threads: 10.5 MB of weights (40 B per thread, wo at 7B) against the 4096 activations
// mode 0: A then B as two kernels (kernel boundary between them)
// mode 1: one kernel, workgroups [0,32) = A, [32,288) = B: B requests its weights into registers FIRST, then waits for a
//         counter the A workgroups bump when their part of the vector is out (one-way signal, agent-scope relaxed
//         atomics for the vector itself: no fences), then computes.  A's workgroups have the lowest ids, so they are
//         resident before any waiter; spins are bounded anyway (a lost signal sets `bad`).
// 200 iterations captured in a hipGraph each; prints us per iteration.   build: hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define AG __HIP_MEMORY_SCOPE_AGENT
constexpr int NA = 32, NB = 256, CHASE_N = 1 << 20;

__device__ __forceinline__ void body_a(int wg, int tid, int iter, const int *chase, float *act, int chain) {
    int p = (wg * 9973 + tid * 131) & (CHASE_N - 1);
    for (int i = 0; i < chain; ++i) p = chase[p];             // dependent round trips through a 4 MB table (L2)
    if (tid < 128) __hip_atomic_store(&act[wg * 128 + tid], (float)iter + (float)(wg * 128 + tid) * 0.001f + (p == -7 ? 1.f : 0.f), __ATOMIC_RELAXED, AG);
}

__device__ __forceinline__ float dot_b(const uint4 (&w)[3], const float *act, int tid) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const float x = __hip_atomic_load(&act[(tid * 3 + u) & 4095], __ATOMIC_RELAXED, AG);
        a += x * (float)((w[u].x & 15) + (w[u].y & 15) + (w[u].z & 15) + (w[u].w & 15));
    }
    return a;
}

__device__ __forceinline__ void finish_b(float a, int wgb, int tid, float *y) {
    __shared__ float red[16];
    for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int i = 0; i < 16; ++i) s += red[i]; y[wgb] = s; }
}

__global__ __launch_bounds__(512) void ka(int iter, const int *chase, float *act, int chain) { body_a(blockIdx.x, threadIdx.x, iter, chase, act, chain); }

__global__ __launch_bounds__(1024) void kb(const uint4 *w, const float *act, float *y) {
    const int tid = threadIdx.x, wgb = blockIdx.x;
    uint4 r[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) r[u] = w[((size_t)wgb * 3 + u) * 1024 + tid];
    finish_b(dot_b(r, act, tid), wgb, tid, y);
}

__global__ __launch_bounds__(1024) void kab(int iter, const int *chase, float *act, int chain, const uint4 *w, float *y, unsigned *ctr, int *bad) {
    const int tid = threadIdx.x;
    if (blockIdx.x < NA) {
        if (tid < 512) body_a(blockIdx.x, tid, iter, chase, act, chain);
        __syncthreads();                                       // the stores of this workgroup are out (vmcnt(0) before the barrier)
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, AG);
        return;
    }
    const int wgb = blockIdx.x - NA;
    uint4 r[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) r[u] = w[((size_t)wgb * 3 + u) * 1024 + tid];
    if (tid == 0) {
        const unsigned target = (unsigned)(iter + 1) * NA;
        unsigned n = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, AG) < target) if (++n > (1u << 14)) { atomicAdd(bad, 1); break; }
    }
    __syncthreads();
    finish_b(dot_b(r, act, tid), wgb, tid, y);
}

int main() {
    int *chase; float *act, *y; uint4 *w; unsigned *ctr; int *bad;
    const size_t wn = (size_t)NB * 3 * 1024;
    hipMalloc(&chase, (size_t)CHASE_N * 4); hipMalloc(&act, 4096 * 4); hipMalloc(&y, NB * 4); hipMalloc(&w, wn * 16); hipMalloc(&ctr, 4); hipMalloc(&bad, 4);
    std::vector<int> hc(CHASE_N);
    for (int i = 0; i < CHASE_N; ++i) hc[i] = (int)(((long long)i * 1103515245LL + 12345) & (CHASE_N - 1));
    hipMemcpy(chase, hc.data(), (size_t)CHASE_N * 4, hipMemcpyHostToDevice);
    hipMemset(w, 0x35, wn * 16);
    hipStream_t st; hipStreamCreate(&st);
    const int iters = 200;
    std::vector<float> y0(NB), y1(NB);
    for (int chain : {4, 8, 16}) {
        float us[2];
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4); hipMemset(act, 0, 4096 * 4);
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int it = 0; it < iters; ++it) {
                if (mode == 0) {
                    hipLaunchKernelGGL(ka, dim3(NA), dim3(512), 0, st, it, chase, act, chain);
                    hipLaunchKernelGGL(kb, dim3(NB), dim3(1024), 0, st, w, act, y);
                } else {
                    hipLaunchKernelGGL(kab, dim3(NA + NB), dim3(1024), 0, st, it, chase, act, chain, w, y, ctr, bad);
                }
            }
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, st); hipStreamSynchronize(st);          // warm-up (counter keeps growing: reset it)
            hipMemset(ctr, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            us[mode] = ms * 1e3f / iters;
            hipMemcpy(mode ? y1.data() : y0.data(), y, NB * 4, hipMemcpyDeviceToHost);
            int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            if (hb) printf("  mode %d: %d lost signals\n", mode, hb);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
        int diff = 0;
        for (int i = 0; i < NB; ++i) diff += y0[i] != y1[i];
        printf("chain %2d: two kernels %.2f us/iter   one kernel with a one-way signal %.2f us/iter   (results differ in %d of %d)\n", chain, us[0], us[1], diff, NB);
    }
    return 0;
}
