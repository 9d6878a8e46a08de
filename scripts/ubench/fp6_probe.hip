// fp6_probe.hip -- what does v_mfma_scale_f32_32x32x64_f8f6f4 compute with fp6 (e2m3) operands?  Checks, against a host model:
//   * packing: element k of a lane's 32 at bits [6k, 6k + 6) of its six operand dwords,
//   * operand lanes: A lane (i, h) = row i, K elements 32 h .. 32 h + 31;  B lane (j, h) likewise for column j,
//   * block scales: byte 0 (op_sel 0) of the lane's scale VGPR, E8M0, applies to that lane's 32 elements,
//   * D layout as the other 32x32 MFMAs, and EXACT integer results for the Q4 x Q8 split (w/2, hi/2 | lo/2, scales 2^1, 2^5 / 2^1).
// build: hipcc --offload-arch=gfx950 -O2 fp6_probe.hip -o fp6_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned *a6, const unsigned *b6, const int *sa, const int *sb, float *d) {
    const int l = threadIdx.x;
    v8i a = {}, b = {};
    for (int i = 0; i < 6; ++i) { a[i] = (int)a6[l * 6 + i]; b[i] = (int)b6[l * 6 + i]; }
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, sa[l], 0, sb[l]);
    for (int e = 0; e < 16; ++e) d[l * 16 + e] = c[e];
}

static unsigned enc_e2m3(float v) {      // exact encode of multiples of 0.125 .. 7.5 that e2m3 can hold
    unsigned s = v < 0 ? 32u : 0u;
    float m = std::fabs(v);
    for (unsigned code = 0; code < 32; ++code) {
        const unsigned e = code >> 3, f = code & 7;
        const float val = e == 0 ? f / 8.0f : (1.0f + f / 8.0f) * (float)(1 << (e - 1));
        if (val == m) return s | code;
    }
    fprintf(stderr, "not representable: %f\n", v);
    exit(1);
}

int main() {
    static float A[32][64], B[32][64];        // A[row][K], B[col][K]: the values the hardware should see BEFORE the block scales
    static int q[32][32], w[32][32];
    srand(5);
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 32; ++k) {
            w[i][k] = rand() % 16 - 8;          // nib - 8
            q[i][k] = rand() % 255 - 127;       // int8 activation of column i
        }
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 32; ++k) {
            A[i][k] = w[i][k] / 2.0f;
            A[i][32 + k] = w[i][k] / 2.0f;
            const int hi = (q[i][k] + 128) / 16 - 8, lo = q[i][k] - 16 * hi;      // q = 16 hi + lo, hi in [-8, 7], lo in [0, 15]
            B[i][k] = hi / 2.0f;
            B[i][32 + k] = lo / 2.0f;
        }
    unsigned ha[64 * 6] = {0}, hb[64 * 6] = {0};
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
        const int i = l & 31, h = l >> 5;
        for (int k = 0; k < 32; ++k) {
            const unsigned ca = enc_e2m3(A[i][32 * h + k]), cb = enc_e2m3(B[i][32 * h + k]);
            const int bit = 6 * k;
            for (int t = 0; t < 6; ++t) {
                if (ca >> t & 1) ha[l * 6 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
                if (cb >> t & 1) hb[l * 6 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
            }
        }
        hsa[l] = 127 + 1;                     // w/2 -> w
        hsb[l] = h == 0 ? 127 + 5 : 127 + 1;  // hi/2 -> 16 hi ; lo/2 -> lo
    }
    unsigned *da, *db; int *dsa, *dsb; float *dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    static float hd[64 * 16];
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 16; ++e) {
            const int col = l & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            long isum = 0;
            for (int kk = 0; kk < 32; ++kk) isum += (long)w[row][kk] * q[col][kk];
            if (hd[l * 16 + e] != (float)isum) {
                if (bad < 8) printf("mismatch lane %d e %d (row %d col %d): got %f want %ld\n", l, e, row, col, hd[l * 16 + e], isum);
                ++bad;
            }
        }
    printf("fp6 probe: %d mismatches of 1024 (D = exact integer block dots: %s)\n", bad, bad ? "NO" : "yes");
    return bad != 0;
}
