// q4_layout.h -- device-resident data layouts of the MI355X Q4_0/Q4_1 x Q8_0 path.
//
// The reference keeps weights as arrays of 20-byte (Q4_0) / 24-byte (Q4_1) AoS blocks and the
// quantized activations as 40-byte AoS Q8_0 blocks (/root/reference/lib/ggml.c:590-626).  Neither
// is 16-byte aligned, so at upload (the Model::load hook, lib/llama.cpp:105) weights are repacked
// LOSSLESSLY into the SoA layouts below; fl_qtensor_download() inverts the repack bit-for-bit.
//
// ---------------------------------------------------------------------------------------------
// QW16 -- weights.  Rows are grouped by 16 (one MFMA M-tile); inside a group the 32-element
// quant blocks are block-major, so that (group, 4 consecutive blocks) is ONE contiguous KiB:
//
//   qs : uint32 [M16/16][KB][16 rows][4 dwords]     M16 = roundup(M,16), KB = K/32
//   d  : float  [M16/16][KB][16 rows]               Q4_0: d_w/16 (see below); Q4_1: d_w
//   m  : float  [M16/16][KB][16 rows]               Q4_1 only: m_w
//
//   * dword g (0..3) of a reference block holds elements 8g..8g+7 (byte j = elements 2j | 2j+1<<4,
//     lib/ggml.c:657).  It is stored at dword position  p = g ^ (((row>>3)&1)<<1)  so that the
//     per-lane ds_read_b32 of an MFMA A-fragment (lane = row + 16*g) is LDS-bank-conflict free.
//   * Q4_0 nibbles are stored XOR 8: a nibble is then the 4-bit two's-complement of (nib-8), and
//     (v<<4)&0xF0F0F0F0 / v&0xF0F0F0F0 are int8 vectors holding 16*(nib-8) -- 3 VALU ops per 8
//     weights, no subtract.  The factor 16 is taken back exactly by storing d_w/16 (a power-of-two
//     scaling, exact for every normal float; |d_w| < 2^-122 is rejected at upload).
//   * Q4_1 nibbles are stored as they are (0..15 are valid non-negative int8).
//   * rows M..M16-1 are zero (nibble value 0 after the transform, scale 0).
//
// QA16 -- quantized activations for the MFMA (N >= 9) path, columns grouped by 16:
//
//   q  : int8  [N16/16][KB][16 cols][32]
//   d  : float [N16/16][KB][16 cols]                d_x           (lib/ggml.c:1363)
//   s  : float [N16/16][KB][16 cols]                d_x * sum(q)  (lib/ggml.c:1433-1440)
//
//   * the 32 int8 of a block are 4 k-groups of 8; group g sits at 8-byte position
//     p = g ^ (((col>>3)&1)<<1) (bank-conflict-free ds_read_b64 of the MFMA B-fragment) and holds
//     elements 8g+{0,2,4,6,1,3,5,7} -- the order in which the nibble unpack above yields them.
//
// QA1 -- quantized activations for the wave-dot (N <= 8) path, one vector:
//
//   q  : int8  [KB][32]   (groups at their natural position, same in-group order as QA16)
//   d  : float [KB],  s : float [KB]
//
// The integer dot over a block is invariant under any permutation of k applied to both operands,
// so none of this changes a result bit.
//
// ---------------------------------------------------------------------------------------------
// H16 copies -- operands of the reference-order ("exact") prefill GEMM (gemm_q4_exact_h16.hip).  That kernel needs the EIGHT
// 4-element sums of every (output, block) separately, as floats (the AVX2 lanes of ggml_vec_dot_q4_{0,1}_q8_0,
// /root/reference/lib/ggml.c:2445-2487); v_mfma_f32_32x32x4_2b_f16 delivers two of them per instruction from f16 operands.
// Nibbles and int8 quants are exact in f16, so the copies hold the integers themselves, already in the order the MFMA
// fragments are read -- the inner loop has no unpack instruction left, and both operands reach LDS by DMA:
//
//   WH16 : f16 [ceil(M16/32) row tiles][KB][2 parts][64 lanes][8]     2 KiB per (32-row tile, block)
//   XH16 : f16 [ceil(N/32) column tiles][KB][2 parts][64 lanes][8]    the same, columns instead of rows
//
//   * lane = i + 32 h is the MFMA lane that reads the 16 bytes: row (column) i of the tile, element half h
//   * the 8 f16 of (part p, lane) are the MFMA steps s = 2p and 2p + 1, four elements each: elements 8s + 4h + {0,1,2,3}
//     of the block -- step s, half h is the reference's AVX2 lane j = 2s + h (elements 4j .. 4j + 3)
//   * WH16 values: Q4_0 16 (nib - 8) (the stored scale is d_w / 16, as for QW16), Q4_1 nib;  XH16 values: the int8 quants
//   * rows / columns past the tensor are zero; derived data: QW16 / QA16 stay the forms of record
//
// QWD -- the nibbles once more, for the reference-order decode kernel (gemv1_q4_exact_llc.hip), whose lane (row, k-group g) owns the
// two AVX2-lane chains 2g, 2g + 1 of its row over the whole of K:
//
//   qwd : uint32 [M16/16][NQ = ceil(KB/4)][16 rows][4 k-groups][4 blocks]      1 KiB per (row group, block quad), lane-linear
//
//   * the dword of (row, g, block) holds the 8 nibbles of k-group g with byte t = element t | element t+4 << 4 (QW16: 2t | 2t+1 << 4),
//     Q4_0 nibbles XOR 8 as in QW16: (v << 4) & 0xF0F0F0F0 and v & 0xF0F0F0F0 ARE the int8x4 operands of the two lane sums
//   * blocks past K in the last quad are zero; scales are read from the QW16 planes (d, m)
#pragma once
#include <stdint.h>

#define FL_QK 32
#define FL_TYPE_Q4_0 2   // enum ggml_type, /root/reference/include/ggml.h:200-212
#define FL_TYPE_Q4_1 3

struct fl_qtensor {
    int type;            // FL_TYPE_Q4_0 | FL_TYPE_Q4_1
    int M, K;            // logical rows / row length
    int M16, KB;         // roundup(M,16), K/32
    uint32_t *qs;        // device, QW16
    float *d;            // device
    float *m;            // device (Q4_1) or nullptr
    int owns;            // 1: qs/d/m were hipMalloc'ed by the library
    uint16_t *h16;       // device, WH16 copy (always library-owned) or nullptr: operand of the reference-order prefill GEMM
    uint32_t *qwd;       // device, QWD copy (always library-owned) or nullptr: nibbles as the reference-order decode kernel reads them
};

// Quantized-activation workspace (either QA16 or QA1 depending on the consumer).
struct fl_qact {
    int8_t *q;
    float *d;
    float *s;
    int N, N16, KB;
    uint16_t *h16;       // XH16 workspace (2 x the bytes of q, columns padded to 32) or nullptr
};

static inline int fl_roundup(int x, int a) { return (x + a - 1) / a * a; }
