#!/usr/bin/env python3
"""Generates coexec3.hip: the instruction mixes a 32x32x32-i8 / f16 rewrite of gemm_q4_mfma_kernel would run,
as fixed-register inline asm (no compiler scheduling).  One "block" = the work of 8 16x16x32 tiles (= coexec2.hip's block)
so the numbers compare directly with profiles/r01_ubench.txt.

register map:  v[0:15] D0   v[16:31] D1   v[32:63] P (two 32x32 tiles)   v[64:95] acc   v[96:99] A   v[100:103] B
               v[104:119] magic C   v[120:135] temps   v136 = -magic / scale a, v137 = scale b   v[140:147] f16 operands
"""
import sys

NV = 148
clob = ",".join('"v%d"' % i for i in range(NV))


def mf32(d, c="104:119"):
    return f"v_mfma_i32_32x32x32_i8 v[{d}:{d+15}], v[96:99], v[100:103], v[{c}]\n"


def mf16x16x64(d):
    return f"v_mfma_i32_16x16x64_i8 v[{d}:{d+3}], v[96:99], v[100:103], v[104:107]\n"


def mf16(d):
    return f"v_mfma_i32_16x16x32_i8 v[{d}:{d+3}], v[96:97], v[100:101], v[104:107]\n"


def pm32():
    return "v_mfma_f32_32x32x1_2b_f32 v[32:63], v136, v137, 0\n"


def pm16(d):
    return f"v_mfma_f32_16x16x1_4b_f32 v[{d}:{d+15}], v136, v137, 0\n"


def f16_16(d):
    return f"v_mfma_f32_16x16x32_f16 v[{d}:{d+3}], v[140:143], v[144:147], 0\n"


def f16_32(d, c):
    return f"v_mfma_f32_32x32x16_f16 v[{d}:{d+15}], v[140:143], v[144:147], {c}\n"


def add(r, src=None):
    return f"v_add_f32_e32 v{r}, v136, v{r if src is None else src}\n"


def fmac(acc, a, b):
    return f"v_fmac_f32_e32 v{acc}, v{a}, v{b}\n"


def unpack(n):  # n independent bit ops standing in for the nibble unpack
    return "".join(f"v_and_b32_e32 v{120 + (i % 16)}, v137, v{96 + (i % 4)}\n" for i in range(n))


modes = []


def mode(name, body):
    modes.append((name, body))


# --- i8 32x32x32 ---------------------------------------------------------------------------------
mode("2 mfma_i32_32x32x32_i8", mf32(0) + mf32(16))
for per in (2, 4, 8):   # VALU ops per 16x16x32 tile-equivalent
    n = per * 4
    body = ""
    for t in range(2):
        body += mf32(16 * t) + "".join(add(64 + (t * n + i) % 32) for i in range(n))
    mode(f"2 x (mfma32 + {n} add)   [{per}/tile]", body)
mode("1 mfma_f32_32x32x1_2b", pm32())
# realistic: P for both tiles, mfma(t+1) issued before the epilogue of tile t; epilogue = 16 sub + 16 fmac
real = pm32() + mf32(0) + mf32(16)
real += "".join(add(120 + i, i) for i in range(16)) + "".join(fmac(64 + i, 120 + i, 32 + i) for i in range(16))
real += "s_nop 7\ns_nop 7\n"
real += "".join(add(120 + i, 16 + i) for i in range(16)) + "".join(fmac(80 + i, 120 + i, 48 + i) for i in range(16))
mode("realistic i8-32: P2b + 2 x (mfma32 + 16 sub + 16 fmac)", real)
# same, software-pipelined over two blocks so that an MFMA is always in flight under an epilogue
real2 = pm32() + mf32(0)
real2 += mf32(16) + "".join(add(120 + i, i) for i in range(16)) + "".join(fmac(64 + i, 120 + i, 32 + i) for i in range(16))
real2 += mf32(0) + "".join(add(120 + i, 16 + i) for i in range(16)) + "".join(fmac(80 + i, 120 + i, 48 + i) for i in range(16))
real2 += pm32()
real2 += mf32(16) + "".join(add(120 + i, i) for i in range(16)) + "".join(fmac(64 + i, 120 + i, 32 + i) for i in range(16))
real2 += "".join(add(120 + i, 16 + i) for i in range(16)) + "".join(fmac(80 + i, 120 + i, 48 + i) for i in range(16))
mode("realistic i8-32 pipelined, TWO blocks (halve it)", real2)
# with the unpack (6 ops per A fragment, one fragment per block at a 32x64 wave tile)
mode("realistic i8-32 + 6 unpack ops", unpack(6) + real)
# VALU scale product instead of the P MFMA: 16 mul + 16 sub + 16 fmac per tile
valu = mf32(0) + mf32(16)
for t in range(2):
    valu += "".join(f"v_mul_f32_e32 v{32 + 16 * t + i}, v136, v{100 + i % 4}\n" for i in range(16))
    valu += "".join(add(120 + i, 16 * t + i) for i in range(16)) + "".join(fmac(64 + 16 * t + i, 120 + i, 32 + 16 * t + i) for i in range(16))
mode("VALU-P i8-32: 2 x (mfma32 + 16 mul + 16 sub + 16 fmac)", valu)

# --- packed f32 VALU: does v_pk_* retire two lanes' worth per issue slot? -------------------------------------------
def pkadd(d, a):
    return f"v_pk_add_f32 v[{d}:{d+1}], v[{a}:{a+1}], v[138:139]\n"


def pkfma(acc, a, b):
    return f"v_pk_fma_f32 v[{acc}:{acc+1}], v[{a}:{a+1}], v[{b}:{b+1}], v[{acc}:{acc+1}]\n"


mode("64 v_add_f32", "".join(add(64 + i % 32) for i in range(64)))
mode("32 v_pk_add_f32 (same 64 results)", "".join(pkadd(64 + 2 * (i % 16), 64 + 2 * (i % 16)) for i in range(32)))
mode("64 v_fmac_f32", "".join(fmac(64 + i % 32, 136, 137) for i in range(64)))
mode("32 v_pk_fma_f32 (same 64 results)", "".join(pkfma(64 + 2 * (i % 16), 32 + 2 * (i % 16), 0 + 2 * (i % 8)) for i in range(32)))
realpk = pm32() + mf32(0) + mf32(16)
realpk += "".join(pkadd(120 + 2 * i, 2 * i) for i in range(8)) + "".join(pkfma(64 + 2 * i, 120 + 2 * i, 32 + 2 * i) for i in range(8))
realpk += "s_nop 7\ns_nop 7\n"
realpk += "".join(pkadd(120 + 2 * i, 16 + 2 * i) for i in range(8)) + "".join(pkfma(80 + 2 * i, 120 + 2 * i, 48 + 2 * i) for i in range(8))
mode("realistic i8-32, PACKED: P2b + 2 x (mfma32 + 8 pk_add + 8 pk_fma)", realpk)
mode("realistic i8-32, PACKED + 6 unpack ops", unpack(6) + realpk)

# --- i8 16x16 forms (reference points) -----------------------------------------------------------------
mode("8 mfma_i32_16x16x32_i8", "".join(mf16(4 * i) for i in range(8)))
mode("8 mfma_i32_16x16x64_i8 (two blocks each: rate check only)", "".join(mf16x16x64(4 * i) for i in range(8)))

# --- f16 exact-integer forms ------------------------------------------------------------------------------
mode("8 mfma_f32_16x16x32_f16 (C=0)", "".join(f16_16(4 * i) for i in range(8)))
body = ""
for i in range(8):
    body += f16_16(4 * i) + "".join(fmac(64 + (4 * i + j) % 32, 136, 137) for j in range(4))
mode("8 x (f16 mfma + 4 fmac)", body)
real16 = pm16(32) + pm16(48) + f16_16(0)
for i in range(8):
    if i < 7:
        real16 += f16_16(4 * (i + 1))
    else:
        real16 += "s_nop 7\n"
    real16 += "".join(fmac(64 + 4 * i + j, 4 * i + j, 32 + 4 * i + j) for j in range(4))
mode("realistic f16-16: 2 P4b + 8 x (f16 mfma + 4 fmac)", real16)
mode("realistic f16-16 + 14 unpack ops (TM=2)", unpack(14) + real16)
mode("4 mfma_f32_32x32x16_f16 (two K=32 tiles)", f16_32(0, "0") + f16_32(0, "v[0:15]") + f16_32(16, "0") + f16_32(16, "v[16:31]"))
real32 = pm32() + f16_32(0, "0") + f16_32(0, "v[0:15]") + f16_32(16, "0") + f16_32(16, "v[16:31]")
real32 = pm32() + f16_32(0, "0") + f16_32(0, "v[0:15]") + f16_32(16, "0") + "".join(fmac(64 + i, i, 32 + i) for i in range(16)) \
    + f16_32(16, "v[16:31]") + "s_nop 7\ns_nop 7\n" + "".join(fmac(80 + i, 16 + i, 48 + i) for i in range(16))
mode("realistic f16-32: P2b + 2 x (2 mfma 32x32x16 + 16 fmac)", real32)

src = f'''// coexec3.hip -- GENERATED by gen_coexec3.py; do not edit.  Instruction mixes of a 32x32x32-i8 / f16 Q4 GEMM inner loop.
// build: hipcc --offload-arch=gfx950 -O3 coexec3.hip -o coexec3
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB {clob}
template <int MODE>
__global__ __launch_bounds__(768) void k(float *out, int n) {{
    asm volatile("v_mov_b32 v96, 0x01010101\\n v_mov_b32 v97, 0x01010101\\n v_mov_b32 v98, 0x01010101\\n v_mov_b32 v99, 0x01010101\\n"
                 "v_mov_b32 v100, 0x01010101\\n v_mov_b32 v101, 0x01010101\\n v_mov_b32 v102, 0x01010101\\n v_mov_b32 v103, 0x01010101\\n"
                 "v_mov_b32 v136, 1.0\\n v_mov_b32 v137, 0\\n v_mov_b32 v138, 1.0\\n v_mov_b32 v139, 1.0\\n"
                 "v_mov_b32 v140, 0\\n v_mov_b32 v141, 0\\n v_mov_b32 v142, 0\\n v_mov_b32 v143, 0\\n"
                 "v_mov_b32 v144, 0\\n v_mov_b32 v145, 0\\n v_mov_b32 v146, 0\\n v_mov_b32 v147, 0\\n" ::: CLOB);
'''
for r in range(104, 120):
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
    const int n = 20000;
    printf("%-62s", name);
    for (int wps = 1; wps <= 3; ++wps) {   // 148 VGPRs: at most 3 waves per SIMD
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
    printf("\\n");
}

int main() {
    float *out;
    hipMalloc(&out, 4096);
'''
for i, (name, _) in enumerate(modes):
    src += f'    run<{i}>("{name}", out);\n'
src += "    return 0;\n}\n"
open(sys.argv[1] if len(sys.argv) > 1 else "coexec3.hip", "w").write(src)
