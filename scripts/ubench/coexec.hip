// coexec.hip -- does the VALU run under an MFMA on gfx950?  Fixed-register inline asm, no compiler scheduling.
// build: hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec ; run: ./coexec
// Each "block" below is executed `n` times by every wave; the table printed is ns per block per SIMD at 1, 2 and 4
// waves per SIMD (the SIMD is shared, so perfect overlap of two pipes shows as max(), none as sum()).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
             "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
             "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"

#define MF(i) "v_mfma_i32_16x16x32_i8 v[" #i "], v[64:65], v[66:67], v[" #i "]\n"
#define S4(a, b, c, d) "v_fma_f32 v" #a ", v68, v69, v" #a "\n v_fma_f32 v" #b ", v68, v69, v" #b "\n v_fma_f32 v" #c ", v68, v69, v" #c "\n v_fma_f32 v" #d ", v68, v69, v" #d "\n"
#define P2(a, b) "v_pk_fma_f32 v[" #a "], v[68:69], v[70:71], v[" #a "]\n v_pk_fma_f32 v[" #b "], v[68:69], v[70:71], v[" #b "]\n"
#define A4(a, b, c, d) "v_add_f32 v" #a ", v68, v" #a "\n v_add_f32 v" #b ", v68, v" #b "\n v_add_f32 v" #c ", v68, v" #c "\n v_add_f32 v" #d ", v68, v" #d "\n"
#define FM(i) "v_mfma_f32_16x16x1_4b_f32 v[" #i "], v68, v69, v[" #i "]\n"

template <int MODE>
__global__ void k(float *out, int n) {
    asm volatile("v_mov_b32 v64, 0x01010101\n v_mov_b32 v65, 0x01010101\n v_mov_b32 v66, 0x01010101\n v_mov_b32 v67, 0x01010101\n"
                 "v_mov_b32 v68, 1.0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 1.0\n v_mov_b32 v71, 0\n" ::: CLOB);
    for (int it = 0; it < n; ++it) {
        if (MODE == 0)       // 8 i8 MFMA
            asm volatile(MF(0:3) MF(4:7) MF(8:11) MF(12:15) MF(16:19) MF(20:23) MF(24:27) MF(28:31) ::: CLOB);
        else if (MODE == 1)  // 32 scalar fma
            asm volatile(S4(32,33,34,35) S4(36,37,38,39) S4(40,41,42,43) S4(44,45,46,47) S4(48,49,50,51) S4(52,53,54,55) S4(56,57,58,59) S4(60,61,62,63) ::: CLOB);
        else if (MODE == 2)  // 16 packed fma (same flops as mode 1)
            asm volatile(P2(32:33,34:35) P2(36:37,38:39) P2(40:41,42:43) P2(44:45,46:47) P2(48:49,50:51) P2(52:53,54:55) P2(56:57,58:59) P2(60:61,62:63) ::: CLOB);
        else if (MODE == 3)  // 8 x (MFMA + 4 scalar fma)
            asm volatile(MF(0:3) S4(32,33,34,35) MF(4:7) S4(36,37,38,39) MF(8:11) S4(40,41,42,43) MF(12:15) S4(44,45,46,47)
                         MF(16:19) S4(48,49,50,51) MF(20:23) S4(52,53,54,55) MF(24:27) S4(56,57,58,59) MF(28:31) S4(60,61,62,63) ::: CLOB);
        else if (MODE == 4)  // 8 x (MFMA + 2 packed fma)
            asm volatile(MF(0:3) P2(32:33,34:35) MF(4:7) P2(36:37,38:39) MF(8:11) P2(40:41,42:43) MF(12:15) P2(44:45,46:47)
                         MF(16:19) P2(48:49,50:51) MF(20:23) P2(52:53,54:55) MF(24:27) P2(56:57,58:59) MF(28:31) P2(60:61,62:63) ::: CLOB);
        else if (MODE == 5)  // 2 f32 16x16x1 4-block MFMA
            asm volatile(FM(0:15) FM(16:31) ::: CLOB);
        else if (MODE == 6)  // 8 x (MFMA + 2 scalar fma): VALU well under the MFMA time
            asm volatile(MF(0:3) "v_fma_f32 v32, v68, v69, v32\n v_fma_f32 v33, v68, v69, v33\n" MF(4:7) "v_fma_f32 v34, v68, v69, v34\n v_fma_f32 v35, v68, v69, v35\n"
                         MF(8:11) "v_fma_f32 v36, v68, v69, v36\n v_fma_f32 v37, v68, v69, v37\n" MF(12:15) "v_fma_f32 v38, v68, v69, v38\n v_fma_f32 v39, v68, v69, v39\n"
                         MF(16:19) "v_fma_f32 v40, v68, v69, v40\n v_fma_f32 v41, v68, v69, v41\n" MF(20:23) "v_fma_f32 v42, v68, v69, v42\n v_fma_f32 v43, v68, v69, v43\n"
                         MF(24:27) "v_fma_f32 v44, v68, v69, v44\n v_fma_f32 v45, v68, v69, v45\n" MF(28:31) "v_fma_f32 v46, v68, v69, v46\n v_fma_f32 v47, v68, v69, v47\n" ::: CLOB);
        else if (MODE == 7)  // 8 x (MFMA + 1 packed fma)
            asm volatile(MF(0:3) "v_pk_fma_f32 v[32:33], v[68:69], v[70:71], v[32:33]\n" MF(4:7) "v_pk_fma_f32 v[34:35], v[68:69], v[70:71], v[34:35]\n"
                         MF(8:11) "v_pk_fma_f32 v[36:37], v[68:69], v[70:71], v[36:37]\n" MF(12:15) "v_pk_fma_f32 v[38:39], v[68:69], v[70:71], v[38:39]\n"
                         MF(16:19) "v_pk_fma_f32 v[40:41], v[68:69], v[70:71], v[40:41]\n" MF(20:23) "v_pk_fma_f32 v[42:43], v[68:69], v[70:71], v[42:43]\n"
                         MF(24:27) "v_pk_fma_f32 v[44:45], v[68:69], v[70:71], v[44:45]\n" MF(28:31) "v_pk_fma_f32 v[46:47], v[68:69], v[70:71], v[46:47]\n" ::: CLOB);
        else if (MODE == 8)  // 32 scalar add
            asm volatile(A4(32,33,34,35) A4(36,37,38,39) A4(40,41,42,43) A4(44,45,46,47) A4(48,49,50,51) A4(52,53,54,55) A4(56,57,58,59) A4(60,61,62,63) ::: CLOB);
        else if (MODE == 9)  // epilogue-like dependent use: MFMA(t+1) then 4 scalar fma READING the result of MFMA(t)
            asm volatile(MF(0:3) MF(4:7)
                         "v_fma_f32 v32, v0, v69, v32\n v_fma_f32 v33, v1, v69, v33\n v_fma_f32 v34, v2, v69, v34\n v_fma_f32 v35, v3, v69, v35\n" MF(8:11)
                         "v_fma_f32 v36, v4, v69, v36\n v_fma_f32 v37, v5, v69, v37\n v_fma_f32 v38, v6, v69, v38\n v_fma_f32 v39, v7, v69, v39\n" MF(12:15)
                         "v_fma_f32 v40, v8, v69, v40\n v_fma_f32 v41, v9, v69, v41\n v_fma_f32 v42, v10, v69, v42\n v_fma_f32 v43, v11, v69, v43\n" MF(16:19)
                         "v_fma_f32 v44, v12, v69, v44\n v_fma_f32 v45, v13, v69, v45\n v_fma_f32 v46, v14, v69, v46\n v_fma_f32 v47, v15, v69, v47\n" MF(20:23)
                         "v_fma_f32 v48, v16, v69, v48\n v_fma_f32 v49, v17, v69, v49\n v_fma_f32 v50, v18, v69, v50\n v_fma_f32 v51, v19, v69, v51\n" MF(24:27)
                         "v_fma_f32 v52, v20, v69, v52\n v_fma_f32 v53, v21, v69, v53\n v_fma_f32 v54, v22, v69, v54\n v_fma_f32 v55, v23, v69, v55\n" MF(28:31)
                         "v_fma_f32 v56, v24, v69, v56\n v_fma_f32 v57, v25, v69, v57\n v_fma_f32 v58, v26, v69, v58\n v_fma_f32 v59, v27, v69, v59\n"
                         "s_nop 7\n v_fma_f32 v60, v28, v69, v60\n v_fma_f32 v61, v29, v69, v61\n v_fma_f32 v62, v30, v69, v62\n v_fma_f32 v63, v31, v69, v63\n" ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v32" : "=v"(r)::CLOB);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int MODE>
static void run(const char *name, float *out) {
    const int n = 20000;
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw/SIMD: %7.1f ns/blk/SIMD", wps, ms * 1e6 / n / wps);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    run<0>("8 mfma_i32_16x16x32_i8", out);
    run<1>("32 v_fma_f32", out);
    run<8>("32 v_add_f32", out);
    run<2>("16 v_pk_fma_f32", out);
    run<5>("2 mfma_f32_16x16x1_4b", out);
    run<3>("8 x (mfma + 4 v_fma_f32)", out);
    run<4>("8 x (mfma + 2 v_pk_fma_f32)", out);
    run<6>("8 x (mfma + 2 v_fma_f32)", out);
    run<7>("8 x (mfma + 1 v_pk_fma_f32)", out);
    run<9>("8 x (mfma(t+1) + 4 v_fma reading mfma(t))", out);
    return 0;
}
