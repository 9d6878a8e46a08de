// coexec2.hip -- how many scalar VALU ops hide under one i8 MFMA on gfx950, and does the op kind matter?
// build: hipcc --offload-arch=gfx950 -O3 coexec2.hip -o coexec2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
             "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
             "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99", \
             "v100","v101","v102","v103"

// the kernel body is generated as a string at host side? no: keep it static -- variants via template + constexpr strings
#define MF(d) "v_mfma_i32_16x16x32_i8 v[" d "], v[96:97], v[98:99], v[100:103]\n"
#define MFZ(d) "v_mfma_i32_16x16x32_i8 v[" d "], v[96:97], v[98:99], 0\n"
#define ADD(r) "v_add_f32_e32 v" r ", v95, v" r "\n"
#define FMAC(r) "v_fmac_f32_e32 v" r ", v94, v95\n"
#define FMA3(r) "v_fma_f32 v" r ", v94, v95, v" r "\n"
#define CVT(r) "v_cvt_f32_i32_e32 v" r ", v" r "\n"
#define PM(d) "v_mfma_f32_16x16x1_4b_f32 v[" d "], v94, v95, 0\n"
// 8 accumulator quads v[0:31]; VALU playground v32..v93
#define V2(OP, a, b) OP(#a) OP(#b)
#define V4(OP, a, b, c, d) OP(#a) OP(#b) OP(#c) OP(#d)

template <int MODE>
__global__ void k(float *out, int n) {
    asm volatile("v_mov_b32 v96, 0x01010101\n v_mov_b32 v97, 0x01010101\n v_mov_b32 v98, 0x01010101\n v_mov_b32 v99, 0x01010101\n"
                 "v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v94, 1.0\n v_mov_b32 v95, 0\n" ::: CLOB);
    for (int it = 0; it < n; ++it) {
        if (MODE == 0)
            asm volatile(MF("0:3") MF("4:7") MF("8:11") MF("12:15") MF("16:19") MF("20:23") MF("24:27") MF("28:31") ::: CLOB);
#define BLK(MFX, E0, E1, E2, E3, E4, E5, E6, E7) \
            asm volatile(MFX("0:3") E0 MFX("4:7") E1 MFX("8:11") E2 MFX("12:15") E3 MFX("16:19") E4 MFX("20:23") E5 MFX("24:27") E6 MFX("28:31") E7 ::: CLOB)
        else if (MODE == 1)   // + 2 add
            BLK(MF, V2(ADD,32,33), V2(ADD,34,35), V2(ADD,36,37), V2(ADD,38,39), V2(ADD,40,41), V2(ADD,42,43), V2(ADD,44,45), V2(ADD,46,47));
        else if (MODE == 2)   // + 4 add
            BLK(MF, V4(ADD,32,33,34,35), V4(ADD,36,37,38,39), V4(ADD,40,41,42,43), V4(ADD,44,45,46,47), V4(ADD,48,49,50,51), V4(ADD,52,53,54,55), V4(ADD,56,57,58,59), V4(ADD,60,61,62,63));
        else if (MODE == 3)   // + 8 add
            BLK(MF, V4(ADD,32,33,34,35) V4(ADD,64,65,66,67), V4(ADD,36,37,38,39) V4(ADD,68,69,70,71), V4(ADD,40,41,42,43) V4(ADD,72,73,74,75), V4(ADD,44,45,46,47) V4(ADD,76,77,78,79),
                V4(ADD,48,49,50,51) V4(ADD,80,81,82,83), V4(ADD,52,53,54,55) V4(ADD,84,85,86,87), V4(ADD,56,57,58,59) V4(ADD,88,89,90,91), V4(ADD,60,61,62,63) V4(ADD,32,33,34,35));
        else if (MODE == 4)   // + 8 fmac (VOP2)
            BLK(MF, V4(FMAC,32,33,34,35) V4(FMAC,64,65,66,67), V4(FMAC,36,37,38,39) V4(FMAC,68,69,70,71), V4(FMAC,40,41,42,43) V4(FMAC,72,73,74,75), V4(FMAC,44,45,46,47) V4(FMAC,76,77,78,79),
                V4(FMAC,48,49,50,51) V4(FMAC,80,81,82,83), V4(FMAC,52,53,54,55) V4(FMAC,84,85,86,87), V4(FMAC,56,57,58,59) V4(FMAC,88,89,90,91), V4(FMAC,60,61,62,63) V4(FMAC,32,33,34,35));
        else if (MODE == 5)   // + 8 fma (VOP3)
            BLK(MF, V4(FMA3,32,33,34,35) V4(FMA3,64,65,66,67), V4(FMA3,36,37,38,39) V4(FMA3,68,69,70,71), V4(FMA3,40,41,42,43) V4(FMA3,72,73,74,75), V4(FMA3,44,45,46,47) V4(FMA3,76,77,78,79),
                V4(FMA3,48,49,50,51) V4(FMA3,80,81,82,83), V4(FMA3,52,53,54,55) V4(FMA3,84,85,86,87), V4(FMA3,56,57,58,59) V4(FMA3,88,89,90,91), V4(FMA3,60,61,62,63) V4(FMA3,32,33,34,35));
        else if (MODE == 6)   // + 8 cvt
            BLK(MF, V4(CVT,32,33,34,35) V4(CVT,64,65,66,67), V4(CVT,36,37,38,39) V4(CVT,68,69,70,71), V4(CVT,40,41,42,43) V4(CVT,72,73,74,75), V4(CVT,44,45,46,47) V4(CVT,76,77,78,79),
                V4(CVT,48,49,50,51) V4(CVT,80,81,82,83), V4(CVT,52,53,54,55) V4(CVT,84,85,86,87), V4(CVT,56,57,58,59) V4(CVT,88,89,90,91), V4(CVT,60,61,62,63) V4(CVT,32,33,34,35));
        else if (MODE == 7)   // C = 0 inline, + 8 add
            BLK(MFZ, V4(ADD,32,33,34,35) V4(ADD,64,65,66,67), V4(ADD,36,37,38,39) V4(ADD,68,69,70,71), V4(ADD,40,41,42,43) V4(ADD,72,73,74,75), V4(ADD,44,45,46,47) V4(ADD,76,77,78,79),
                V4(ADD,48,49,50,51) V4(ADD,80,81,82,83), V4(ADD,52,53,54,55) V4(ADD,84,85,86,87), V4(ADD,56,57,58,59) V4(ADD,88,89,90,91), V4(ADD,60,61,62,63) V4(ADD,32,33,34,35));
        else if (MODE == 8)   // 8 add only per slot (no mfma): VALU baseline for 64 adds
            asm volatile(V4(ADD,32,33,34,35) V4(ADD,64,65,66,67) V4(ADD,36,37,38,39) V4(ADD,68,69,70,71) V4(ADD,40,41,42,43) V4(ADD,72,73,74,75) V4(ADD,44,45,46,47) V4(ADD,76,77,78,79)
                         V4(ADD,48,49,50,51) V4(ADD,80,81,82,83) V4(ADD,52,53,54,55) V4(ADD,84,85,86,87) V4(ADD,56,57,58,59) V4(ADD,88,89,90,91) V4(ADD,60,61,62,63) V4(ADD,32,33,34,35) ::: CLOB);
        else if (MODE == 9)   // C = 0 inline only
            asm volatile(MFZ("0:3") MFZ("4:7") MFZ("8:11") MFZ("12:15") MFZ("16:19") MFZ("20:23") MFZ("24:27") MFZ("28:31") ::: CLOB);
        else if (MODE == 10)  // + 12 add
            BLK(MF, V4(ADD,32,33,34,35) V4(ADD,64,65,66,67) V4(ADD,48,49,50,51), V4(ADD,36,37,38,39) V4(ADD,68,69,70,71) V4(ADD,52,53,54,55), V4(ADD,40,41,42,43) V4(ADD,72,73,74,75) V4(ADD,56,57,58,59), V4(ADD,44,45,46,47) V4(ADD,76,77,78,79) V4(ADD,60,61,62,63),
                V4(ADD,48,49,50,51) V4(ADD,80,81,82,83) V4(ADD,32,33,34,35), V4(ADD,52,53,54,55) V4(ADD,84,85,86,87) V4(ADD,36,37,38,39), V4(ADD,56,57,58,59) V4(ADD,88,89,90,91) V4(ADD,40,41,42,43), V4(ADD,60,61,62,63) V4(ADD,32,33,34,35) V4(ADD,44,45,46,47));
        else if (MODE == 11)  // realistic: mfma(t+1); 4 add reading mfma(t); 4 fmac; 2 scale MFMAs per 8 tiles
            asm volatile(PM("64:79") MF("0:3") MF("4:7")
                         "v_add_f32_e32 v32, v95, v0\n v_add_f32_e32 v33, v95, v1\n v_add_f32_e32 v34, v95, v2\n v_add_f32_e32 v35, v95, v3\n" "v_fmac_f32_e32 v40, v32, v64\n v_fmac_f32_e32 v41, v33, v65\n v_fmac_f32_e32 v42, v34, v66\n v_fmac_f32_e32 v43, v35, v67\n" MF("8:11")
                         "v_add_f32_e32 v32, v95, v4\n v_add_f32_e32 v33, v95, v5\n v_add_f32_e32 v34, v95, v6\n v_add_f32_e32 v35, v95, v7\n" "v_fmac_f32_e32 v44, v32, v68\n v_fmac_f32_e32 v45, v33, v69\n v_fmac_f32_e32 v46, v34, v70\n v_fmac_f32_e32 v47, v35, v71\n" MF("12:15")
                         "v_add_f32_e32 v32, v95, v8\n v_add_f32_e32 v33, v95, v9\n v_add_f32_e32 v34, v95, v10\n v_add_f32_e32 v35, v95, v11\n" "v_fmac_f32_e32 v48, v32, v72\n v_fmac_f32_e32 v49, v33, v73\n v_fmac_f32_e32 v50, v34, v74\n v_fmac_f32_e32 v51, v35, v75\n" PM("80:95") MF("16:19")
                         "v_add_f32_e32 v32, v95, v12\n v_add_f32_e32 v33, v95, v13\n v_add_f32_e32 v34, v95, v14\n v_add_f32_e32 v35, v95, v15\n" "v_fmac_f32_e32 v52, v32, v76\n v_fmac_f32_e32 v53, v33, v77\n v_fmac_f32_e32 v54, v34, v78\n v_fmac_f32_e32 v55, v35, v79\n" MF("20:23")
                         "v_add_f32_e32 v32, v95, v16\n v_add_f32_e32 v33, v95, v17\n v_add_f32_e32 v34, v95, v18\n v_add_f32_e32 v35, v95, v19\n" "v_fmac_f32_e32 v40, v32, v80\n v_fmac_f32_e32 v41, v33, v81\n v_fmac_f32_e32 v42, v34, v82\n v_fmac_f32_e32 v43, v35, v83\n" MF("24:27")
                         "v_add_f32_e32 v32, v95, v20\n v_add_f32_e32 v33, v95, v21\n v_add_f32_e32 v34, v95, v22\n v_add_f32_e32 v35, v95, v23\n" "v_fmac_f32_e32 v44, v32, v84\n v_fmac_f32_e32 v45, v33, v85\n v_fmac_f32_e32 v46, v34, v86\n v_fmac_f32_e32 v47, v35, v87\n" MF("28:31")
                         "v_add_f32_e32 v32, v95, v24\n v_add_f32_e32 v33, v95, v25\n v_add_f32_e32 v34, v95, v26\n v_add_f32_e32 v35, v95, v27\n" "v_fmac_f32_e32 v48, v32, v88\n v_fmac_f32_e32 v49, v33, v89\n v_fmac_f32_e32 v50, v34, v90\n v_fmac_f32_e32 v51, v35, v91\n"
                         "s_nop 7\n"
                         "v_add_f32_e32 v32, v95, v28\n v_add_f32_e32 v33, v95, v29\n v_add_f32_e32 v34, v95, v30\n v_add_f32_e32 v35, v95, v31\n" "v_fmac_f32_e32 v52, v32, v92\n v_fmac_f32_e32 v53, v33, v93\n v_fmac_f32_e32 v54, v34, v94\n v_fmac_f32_e32 v55, v35, v94\n" ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v32" : "=v"(r)::CLOB);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int MODE>
static void run(const char *name, float *out) {
    const int n = 20000;
    printf("%-40s", name);
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
        printf("  %dw: %6.1f ns", wps, ms * 1e6 / n / wps);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("   (per block of 8 MFMA, per SIMD)\n");
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
    run<0>("8 mfma (C in VGPRs)", out);
    run<9>("8 mfma (C = 0 inline)", out);
    run<8>("64 v_add_f32 alone", out);
    run<1>("8 x (mfma + 2 add)", out);
    run<2>("8 x (mfma + 4 add)", out);
    run<3>("8 x (mfma + 8 add)", out);
    run<10>("8 x (mfma + 12 add)", out);
    run<4>("8 x (mfma + 8 fmac_e32)", out);
    run<5>("8 x (mfma + 8 fma vop3)", out);
    run<6>("8 x (mfma + 8 cvt_f32_i32)", out);
    run<7>("8 x (mfma C=0 + 8 add)", out);
    run<11>("realistic: 2 P-mfma + 8 x (mfma + 4 add + 4 fmac, dependent)", out);
    return 0;
}
