// gridbar.hip -- cost of a software grid barrier (agent-scope release/acquire, 8 XCDs) and of streaming reads from a
// persistent grid on MI355X.  build: hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned &target, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// mode 0: barriers only.  mode 1: every WG writes a value before the barrier and reads its neighbour's after (checks
// cross-XCD visibility).  mode 2: stream `bytes` from `w` between barriers (one phase = bytes/phases), no prefetch.
// mode 3: like 2, but the first 16-byte-per-thread chunk of the NEXT phase is requested before the barrier.
__global__ __launch_bounds__(1024) void k(unsigned *ctr, int nbar, int mode, unsigned *mail, int *bad, const uint4 *w,
                                          size_t n16_per_phase, unsigned *sink) {
    unsigned target = 0;
    const unsigned nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    unsigned acc = 0;
    uint4 pre = {0, 0, 0, 0};
    const size_t stride = (size_t)nwg * 1024;
    if (mode == 3) pre = w[(size_t)wg * 1024 + tid];
    for (int b = 0; b < nbar; ++b) {
        if (mode == 1) {
            if (tid == 0) mail[wg] = (unsigned)b * 1000003u + wg;
        } else if (mode >= 2) {
            const uint4 *p = w + (size_t)b * n16_per_phase;
            size_t i = (size_t)wg * 1024 + tid;
            if (mode == 3) { acc ^= pre.x ^ pre.y ^ pre.z ^ pre.w; i += stride; }
            uint4 v[4];
            for (; i + 3 * stride < n16_per_phase; i += 4 * stride) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = p[i + u * stride];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            }
            for (; i < n16_per_phase; i += stride) { uint4 t = p[i]; acc ^= t.x ^ t.y ^ t.z ^ t.w; }
            if (mode == 3 && b + 1 < nbar) pre = (p + n16_per_phase)[(size_t)wg * 1024 + tid];
        }
        grid_barrier(ctr, target, nwg);
        if (mode == 1 && tid == 0) {
            const unsigned other = (wg + 37) % nwg;
            const unsigned got = __hip_atomic_load(&mail[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (unsigned)b * 1000003u + other) atomicAdd(bad, 1);
        }
        if (mode == 1) grid_barrier(ctr, target, nwg);   // nobody overwrites before everyone has read
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int nwg = prop.multiProcessorCount;
    printf("CUs %d\n", nwg);
    unsigned *ctr, *mail, *sink;
    int *bad;
    hipMalloc(&ctr, 4); hipMalloc(&mail, 4096 * 4); hipMalloc(&bad, 4); hipMalloc(&sink, 4);
    const size_t total = (size_t)3600 << 20;      // 3.6 GB like the 7B weights
    uint4 *w;
    hipMalloc(&w, total);
    hipMemset(w, 1, total);
    for (int mode = 0; mode < 4; ++mode) {
        const int nbar = mode < 2 ? 2000 : 160;
        const size_t n16 = total / 16 / nbar;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            void *args[] = {&ctr, (void *)&nbar, &mode, &mail, &bad, &w, (void *)&n16, &sink};
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((const void *)k, dim3(nwg), dim3(1024), args, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            if (rep == 1) {
                if (mode < 2) printf("mode %d: %s  %.2f us per barrier%s  bad=%d\n", mode, hipGetErrorString(e), ms * 1e3 / nbar / (mode == 1 ? 2 : 1), mode == 1 ? " (with mailbox check)" : "", hb);
                else printf("mode %d: %s  %d phases of %.1f MB: %.3f ms total, %.2f TB/s, %.2f us per phase\n", mode, hipGetErrorString(e), nbar, n16 * 16 / 1e6, ms, total / ms / 1e9, ms * 1e3 / nbar);
            }
        }
    }
    return 0;
}
