#!/usr/bin/env python3
"""Generates coexec4.hip (round 3): fixed-register inline-asm instruction mixes, same harness and units as coexec3
(ns per loop trip and wave, 1..2 waves per SIMD), for two questions:

 A. VERDICT r02 item 3 -- does the block-scaled fp6 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, fp6 x fp6) beat the i8 form in the
    FAST kernel's mix?  One K = 64 instruction is one 32-element quant block of a 32x32 tile: A = [w/2, w/2] (e2m3, exact), B =
    [hi/2, lo/2] with q = 16 hi + lo, E8M0 block scales 2^6 / 2^2: D = float(isum) directly, no magic subtract.  Rows: the bare
    instruction, the realistic mix (P outer product + MFMA + 16 fmac per tile), the same with the nibble -> fp6 repack in the loop,
    and the row coexec3 lacked (f16-32 form + 14 unpack ops).
 B. exact mode (reference summation order): the 8 lane sums of every block are needed separately.  K = 4 MFMA forms
    (v_mfma_f32_32x32x4_2b_f16: two 4-element groups per instruction, D = float(lane sum)) against v_dot4 + v_cvt on the VALU,
    each followed by the 8 fma per output and block that the reference's order fixes.

register map: v[0:31] D0  v[32:63] P  v[64:95] acc  v[96:103] A/B i8  v[104:119] magic C  v[120:135] temps  v136/v137 scales
              v[140:147] f16 operands  v[148:179] D1  v[180:185] fp6 A  v[186:191] fp6 B  v192/v193 E8M0 scales
"""
import sys

NV = 196
clob = ",".join('"v%d"' % i for i in range(NV))


def mf32(d, c="104:119"):
    return f"v_mfma_i32_32x32x32_i8 v[{d}:{d+15}], v[96:99], v[100:103], v[{c}]\n"


def pm32():
    return "v_mfma_f32_32x32x1_2b_f32 v[32:63], v136, v137, 0\n"


def f6(d):
    return f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{d}:{d+15}], v[180:185], v[186:191], 0, v192, v193 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"


def f8(d):
    return f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{d}:{d+15}], v[96:103], v[180:187], 0, v192, v193 op_sel_hi:[0,0,0] cbsz:0 blgp:0\n"


def f16_32(d, c):
    return f"v_mfma_f32_32x32x16_f16 v[{d}:{d+15}], v[140:143], v[144:147], {c}\n"


def k4(d):      # two 32x32x4 blocks: 2 lane-sum groups of a 32x32 tile
    return f"v_mfma_f32_32x32x4_2b_f16 v[{d}:{d+31}], v[140:141], v[144:145], 0\n"


def k4_16(d):   # four 16x16x4 blocks
    return f"v_mfma_f32_16x16x4_4b_f16 v[{d}:{d+15}], v[140:141], v[144:145], 0\n"


def add(r, src=None):
    return f"v_add_f32_e32 v{r}, v136, v{r if src is None else src}\n"


def fmac(acc, a, b):
    return f"v_fmac_f32_e32 v{acc}, v{a}, v{b}\n"


def unpack(n):
    return "".join(f"v_and_b32_e32 v{120 + (i % 16)}, v137, v{96 + (i % 4)}\n" for i in range(n))


modes = []


def mode(name, body):
    modes.append((name, body))


# ---------------------------------------------------------------- A: fp6 block-scaled MFMA in the fast mix
mode("2 mfma_i32_32x32x32_i8                       (reference)", mf32(0) + mf32(16))
mode("2 mfma_scale_f32_32x32x64_f8f6f4 fp6 x fp6", f6(0) + f6(16))
mode("2 mfma_scale_f32_32x32x64_f8f6f4 fp8 x fp8", f8(0) + f8(16))
real = pm32() + mf32(0) + mf32(16)
real += "".join(add(120 + i, i) for i in range(16)) + "".join(fmac(64 + i, 120 + i, 32 + i) for i in range(16))
real += "s_nop 7\ns_nop 7\n"
real += "".join(add(120 + i, 16 + i) for i in range(16)) + "".join(fmac(80 + i, 120 + i, 48 + i) for i in range(16))
mode("realistic i8-32: P2b + 2 x (mfma32 + 16 sub + 16 fmac)  (shipped)", real)
real6 = pm32() + f6(0) + f6(16)
real6 += "".join(fmac(64 + i, i, 32 + i) for i in range(16))
real6 += "s_nop 7\ns_nop 7\n"
real6 += "".join(fmac(80 + i, 16 + i, 48 + i) for i in range(16))
mode("realistic fp6: P2b + 2 x (mfma_scale fp6 + 16 fmac)", real6)
mode("realistic i8-32 + 6 unpack ops   (QW16 nibbles -> i8, shipped)", unpack(6) + real)
mode("realistic fp6 + 0 unpack ops     (pre-packed fp6 weight copy)", real6)
mode("realistic fp6 + 24 unpack ops    (QW16 nibbles -> fp6 codes in the loop, optimistic)", unpack(24) + real6)
mode("realistic fp6 + 40 unpack ops    (the same, realistic bit packing)", unpack(40) + real6)
real32 = pm32() + f16_32(0, "0") + f16_32(0, "v[0:15]") + f16_32(16, "0") + "".join(fmac(64 + i, i, 32 + i) for i in range(16)) \
    + f16_32(16, "v[16:31]") + "s_nop 7\ns_nop 7\n" + "".join(fmac(80 + i, 16 + i, 48 + i) for i in range(16))
mode("realistic f16-32: P2b + 2 x (2 mfma 32x32x16 + 16 fmac)", real32)
mode("realistic f16-32 + 14 unpack ops (the row r02 lacked)", unpack(14) + real32)

# ---------------------------------------------------------------- B: exact mode, 8 lane sums per output and block
mode("4 mfma_f32_32x32x4_2b_f16 (the 8 lane sums of a 32x32 tile, bare)", k4(0) + k4(148) + k4(0) + k4(148))
mode("8 mfma_f32_16x16x4_4b_f16 (the same work, 16x16 form)", "".join(k4_16(16 * (i % 2)) for i in range(8)))
mode("128 v_fmac_f32 (the reference's 8 fma per output and block, bare)", "".join(fmac(64 + i % 32, 136, 137) for i in range(128)))
# MFMA route: dd (half a P2b per block) + 4 x (K4 MFMA + 32 fmac reading its D)
xm = pm32()
for blk in range(2):
    for s in range(4):
        d = 0 if s % 2 == 0 else 148
        xm += k4(d)
        xm += "".join(fmac(64 + (i % 32), d + i, 32 + (i % 16) + 16 * blk) for i in range(32))
mode("exact, MFMA route, TWO blocks: P2b + 8 x (K4 mfma + 32 fmac)  (halve it)", xm)
# the same, MFMA of the next group issued before the fmacs of the current one
xp = pm32() + k4(0)
for s in range(8):
    d, dn = (0, 148) if s % 2 == 0 else (148, 0)
    if s < 7:
        xp += k4(dn)
    xp += "".join(fmac(64 + (i % 32), d + i, 32 + (i % 16) + 16 * (s // 4)) for i in range(32))
mode("exact, MFMA route pipelined, TWO blocks  (halve it)", xp)
# VALU route: per 32x32 tile and block 128 x (dot4 + cvt + fmac) + 16 mul
xv = "".join(f"v_mul_f32_e32 v{32 + i}, v136, v{100 + i % 4}\n" for i in range(16))
for i in range(128):
    t = 120 + i % 16
    xv += f"v_dot4_i32_i8 v{t}, v{96 + i % 4}, v{100 + i % 4}, 0\n"
    xv += f"v_cvt_f32_i32_e32 v{t}, v{t}\n"
    xv += fmac(64 + i % 32, t, 32 + i % 16)
mode("exact, VALU route, ONE block: 16 mul + 128 x (dot4 + cvt + fmac)", xv)
xv2 = "".join(f"v_mul_f32_e32 v{32 + i}, v136, v{100 + i % 4}\n" for i in range(16))
for g in range(8):      # 16 independent dot4s, then their cvts, then the fmacs: no back-to-back dependent pairs
    xv2 += "".join(f"v_dot4_i32_i8 v{120 + i}, v{96 + i % 4}, v{100 + i % 4}, 0\n" for i in range(16))
    xv2 += "".join(f"v_cvt_f32_i32_e32 v{120 + i}, v{120 + i}\n" for i in range(16))
    xv2 += "".join(fmac(64 + (16 * g + i) % 32, 120 + i, 32 + i) for i in range(16))
mode("exact, VALU route, ONE block, grouped by 16 (no dependent neighbours)", xv2)

src = f'''// coexec4.hip -- GENERATED by gen_coexec4.py; do not edit.  Round 3: fp6 block-scaled MFMA in the fast mix; exact-mode mixes.
// build: hipcc --offload-arch=gfx950 -O3 coexec4.hip -o coexec4
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB {clob}
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int n) {{
    asm volatile("v_mov_b32 v96, 0x01010101\\n v_mov_b32 v97, 0x01010101\\n v_mov_b32 v98, 0x01010101\\n v_mov_b32 v99, 0x01010101\\n"
                 "v_mov_b32 v100, 0x01010101\\n v_mov_b32 v101, 0x01010101\\n v_mov_b32 v102, 0x01010101\\n v_mov_b32 v103, 0x01010101\\n"
                 "v_mov_b32 v136, 1.0\\n v_mov_b32 v137, 0\\n v_mov_b32 v138, 1.0\\n v_mov_b32 v139, 1.0\\n"
                 "v_mov_b32 v140, 0\\n v_mov_b32 v141, 0\\n v_mov_b32 v142, 0\\n v_mov_b32 v143, 0\\n"
                 "v_mov_b32 v144, 0\\n v_mov_b32 v145, 0\\n v_mov_b32 v146, 0\\n v_mov_b32 v147, 0\\n"
                 "v_mov_b32 v192, 0x7f7f7f7f\\n v_mov_b32 v193, 0x7f7f7f7f\\n" ::: CLOB);
'''
for r in list(range(104, 120)) + list(range(180, 192)):
    src += f'    asm volatile("v_mov_b32 v{r}, 0" ::: CLOB);\n'
src += "    for (int it = 0; it < n; ++it) {\n"
for i, (name, body) in enumerate(modes):
    lines = "".join('                "%s\\n"\n' % ln for ln in body.strip().split("\n"))
    src += f"        if (MODE == {i})\n            asm volatile(\n{lines}                ::: CLOB);\n"
src += '''    }
    float r;
    asm volatile("v_add_f32 %0, v0, v64" : "=v"(r)::CLOB);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int MODE>
static void run(const char *name, float *out) {
    const int n = 10000;
    printf("%-86s", name);
    for (int wps = 1; wps <= 2; ++wps) {   // 196 VGPRs: at most 2 waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %7.1f ns", wps, ms * 1e6 / n / wps);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\\n");
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
'''
for i, (name, _) in enumerate(modes):
    src += f'    run<{i}>("{name}", out);\n'
src += "    return 0;\n}\n"
open(sys.argv[1] if len(sys.argv) > 1 else "coexec4.hip", "w").write(src)
