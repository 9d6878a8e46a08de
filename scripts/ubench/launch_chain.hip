// launch_chain.hip -- what a chain of dependent kernel launches costs on this part, whatever the kernels do: N tiny kernels (one
// workgroup / 256 workgroups, each reading the value the previous one wrote) per hipGraph replay and as plain stream launches.
// hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void link(const float *in, float *out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = *in + 1.f; }
int main() {
    const int N = 161;
    float *buf;
    hipMalloc(&buf, 4 * (N + 1));
    hipMemset(buf, 0, 4 * (N + 1));
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int wgs : {1, 256, 768}) {
        for (int threads : {64, 256}) {
            hipGraph_t g;
            hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(link, dim3(wgs), dim3(threads), 0, st, buf + i, buf + i + 1);
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            const int reps = 50;
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            const double tg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r)
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(link, dim3(wgs), dim3(threads), 0, st, buf + i, buf + i + 1);
            hipStreamSynchronize(st);
            const double ts = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
            printf("%3d workgroups x %3d threads: graph replay %.3f ms = %.2f us per kernel   stream launches %.3f ms = %.2f us per kernel\n", wgs,
                   threads, tg * 1e3, tg * 1e6 / N, ts * 1e3, ts * 1e6 / N);
            hipGraphExecDestroy(ge);
            hipGraphDestroy(g);
        }
    }
    return 0;
}
