// stream_gemv.hip -- round 4, VERDICT r3 item 4: what does the decode GEMV's access pattern cost WITHOUT the GEMV?
// A bare nontemporal streaming read of the same QW16 bytes (nibbles: 1-KiB pieces of (16-row group, 4 blocks); scales: 256-byte
// pieces) with the same grid, workgroup size, piece -> wave assignment and loads in flight as gemv_q4_kernel (q4_kernels.hip,
// launch_gemv1) uses for each LLaMA-7B shape, reduced with XORs into one dword per workgroup.  If this runs at the guide's 6.3-6.8
// TB/s while the GEMV fits 3.5 us + bytes / 5 TB/s, the GEMV adds something; if it runs at ~5 TB/s, that is the pattern's ceiling.
// build: hipcc --offload-arch=gfx950 -O3 stream_gemv.hip -o stream_gemv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// one workgroup per 16-row group (PAIR: two groups); wave w takes block-quads w, w + NW, ...; U quads in flight per wave
template <int NW, int U, int PAIR, bool NT>
__global__ __launch_bounds__(64 * NW) void stream_kernel(const v4u *__restrict__ qs, const float *__restrict__ d, int KB, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nquads = KB / 4;
    unsigned acc = 0;
    for (int g2 = 0; g2 < (PAIR ? 2 : 1); ++g2) {
        const int grp = blockIdx.x * (PAIR ? 2 : 1) + g2;
        const v4u *gq = qs + (int64_t)grp * KB * 16;
        const float *gd = d + (int64_t)grp * KB * 16;
        for (int q0 = wave; q0 < nquads; q0 += NW * U) {
            v4u w[U];
            float s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NW;
                const int qq = q < nquads ? q : wave;
                w[u] = NT ? __builtin_nontemporal_load(gq + (int64_t)qq * 64 + lane) : gq[(int64_t)qq * 64 + lane];
                s[u] = NT ? __builtin_nontemporal_load(gd + (int64_t)qq * 64 + lane) : gd[(int64_t)qq * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w ^ __float_as_uint(s[u]);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// the guide's reference point: a plain grid-stride 16-byte read of the same bytes, enough workgroups to fill the chip
template <bool NT>
__global__ __launch_bounds__(256) void flat_kernel(const v4u *__restrict__ p, int64_t n16, unsigned *out) {
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const v4u v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

static void *g_flush;
static double time_us(void (*launch)(void *), void *ctx, int reps = 20) {
    std::vector<float> ts;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < reps + 3; ++r) {
        (void)hipMemsetAsync(g_flush, r, 512u << 20, 0);       // 512 MiB: evict the 256 MiB memory-side cache between launches
        (void)hipEventRecord(e0, 0);
        launch(ctx);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

struct Ctx { const v4u *qs; const float *d; int KB, groups; unsigned *out; int64_t n16; };
template <int NW, int U, int PAIR, bool NT> static void L(void *c_) { Ctx *c = (Ctx *)c_; hipLaunchKernelGGL((stream_kernel<NW, U, PAIR, NT>), dim3(c->groups / (PAIR ? 2 : 1)), dim3(64 * NW), 0, 0, c->qs, c->d, c->KB, c->out); }
template <bool NT> static void F(void *c_) { Ctx *c = (Ctx *)c_; hipLaunchKernelGGL((flat_kernel<NT>), dim3(256 * 8), dim3(256), 0, 0, c->qs, c->n16, c->out); }

int main() {
    (void)hipMalloc(&g_flush, 512u << 20);
    struct Shape { const char *name; int M, K, nw, u, pair; } shapes[] = {
        {"wq|wk|wv 12288x4096", 12288, 4096, 4, 8, 0}, {"wo 4096x4096", 4096, 4096, 16, 2, 0}, {"w1|w3 22016x4096 (pair)", 22016, 4096, 4, 8, 1},
        {"w2 4096x11008", 4096, 11008, 8, 8, 0}, {"lm-head 32000x4096", 32000, 4096, 4, 8, 0}};
    printf("%-26s %9s %8s | %-34s | %-34s | %-22s\n", "shape", "MB", "WGs", "gemv pattern, nt loads", "gemv pattern, plain loads", "flat 16-B read (nt / plain)");
    for (auto &s : shapes) {
        const int KB = s.K / 32, groups = s.M / 16;
        const size_t nq = (size_t)groups * KB * 256, nd = (size_t)groups * KB * 64;
        void *qs, *d; unsigned *out;
        (void)hipMalloc(&qs, nq); (void)hipMalloc(&d, nd); (void)hipMalloc((void **)&out, 1 << 16);
        (void)hipMemset(qs, 0x5a, nq); (void)hipMemset(d, 0, nd);
        Ctx c{(const v4u *)qs, (const float *)d, KB, groups, out, (int64_t)(nq / 16)};
        double t_nt, t_pl;
#define RUN(NW, UU, P) (t_nt = time_us(L<NW, UU, P, true>, &c), t_pl = time_us(L<NW, UU, P, false>, &c))
        if (s.nw == 4 && !s.pair) RUN(4, 8, 0);
        else if (s.nw == 4) RUN(4, 8, 1);
        else if (s.nw == 8) RUN(8, 8, 0);
        else RUN(16, 2, 0);
        const double f_nt = time_us(F<true>, &c), f_pl = time_us(F<false>, &c);
        const double mb = (nq + nd) / 1e6;
        printf("%-26s %9.2f %8d | %7.2f us %6.2f TB/s              | %7.2f us %6.2f TB/s              | %6.2f / %6.2f us  %5.2f / %5.2f TB/s (nibbles only)\n", s.name, mb,
               groups / (s.pair ? 2 : 1), t_nt, mb / t_nt, t_pl, mb / t_pl, f_nt, f_pl, nq / 1e6 / f_nt, nq / 1e6 / f_pl);
        (void)hipFree(qs); (void)hipFree(d); (void)hipFree(out);
    }
    // what more loads in flight / more waves would buy on the worst shape (wo: 256 workgroups = one per CU)
    {
        const int KB = 128, groups = 256;
        const size_t nq = (size_t)groups * KB * 256, nd = (size_t)groups * KB * 64;
        void *qs, *d; unsigned *out;
        (void)hipMalloc(&qs, nq); (void)hipMalloc(&d, nd); (void)hipMalloc((void **)&out, 1 << 16);
        Ctx c{(const v4u *)qs, (const float *)d, KB, groups, out, (int64_t)(nq / 16)};
        printf("wo, other (waves, in flight): 16x2 %.2f us  16x1 %.2f us  8x4 %.2f us  4x8 %.2f us\n", time_us(L<16, 2, 0, true>, &c), time_us(L<16, 1, 0, true>, &c),
               time_us(L<8, 4, 0, true>, &c), time_us(L<4, 8, 0, true>, &c));
    }
    return 0;
}
