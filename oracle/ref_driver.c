/*
 * ref_driver.c -- thin glue that runs the REAL reference op through its own public C API.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no reference code: it is
 * compiled against /root/reference/include/ggml.h where that header lies, and linked with the
 * reference's lib/ggml.c object, by oracle/Makefile, into oracle/_ref/libggml_ref.so.
 *
 * ref_mul_mat_q() builds the one-node graph  y = ggml_mul_mat(W_q4[K,M], x_f32[K,N])  and runs
 * ggml_graph_compute on it with n_threads workers, i.e. exactly the code path
 *   ggml_graph_compute (lib/ggml.c:10811) -> ggml_compute_forward_mul_mat (:8178)
 *   -> ggml_compute_forward_mul_mat_q_f32 (:7928) -> quantize_row_q8_0 / ggml_vec_dot_q4_x_q8_0
 * including the reference's pthread spin pool.  It is what bench.py times as
 * cpu_baseline.kind = "reference" and what the oracle restatement is pinned against.
 */
#include <stdlib.h>
#include <string.h>
#include "ggml.h"

/* type: 2 = GGML_TYPE_Q4_0, 3 = GGML_TYPE_Q4_1.  W: M rows of K/32 AoS blocks (host).
 * x: N rows of K floats.  y: N rows of M floats.  reps >= 1 repeats the graph compute (timing).
 * Returns 0 on success. */
int ref_mul_mat_q(int type, const void *W, const float *x, float *y,
                  int M, int K, int N, int n_threads, int reps) {
    if (type != GGML_TYPE_Q4_0 && type != GGML_TYPE_Q4_1) return -1;
    const size_t wbytes = (size_t)M * (size_t)(K / 32) * ggml_type_size((enum ggml_type)type);
    const size_t xbytes = (size_t)N * K * sizeof(float);
    const size_t ybytes = (size_t)N * M * sizeof(float);
    /* work buffer for the Q8_0 copy of x (lib/ggml.c:10949) + tensor headers + slack */
    const size_t need = wbytes + xbytes + ybytes + (size_t)N * (K / 32) * 40 * 2 + (64u << 20);
    struct ggml_init_params ip;
    memset(&ip, 0, sizeof ip);
    ip.mem_size = need;
    ip.mem_buffer = NULL;
    struct ggml_context *ctx = ggml_init(ip);
    if (!ctx) return -2;
    struct ggml_tensor *tw = ggml_new_tensor_2d(ctx, (enum ggml_type)type, K, M);
    struct ggml_tensor *tx = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, N);
    memcpy(tw->data, W, wbytes);
    memcpy(tx->data, x, xbytes);
    struct ggml_tensor *ty = ggml_mul_mat(ctx, tw, tx);
    struct ggml_cgraph gf = ggml_build_forward(ty);
    gf.n_threads = n_threads;
    for (int r = 0; r < (reps < 1 ? 1 : reps); ++r) ggml_graph_compute(ctx, &gf);
    memcpy(y, ty->data, ybytes);
    ggml_free(ctx);
    return 0;
}

/* Same graph, but W/x are copied in once and only ggml_graph_compute is inside the caller's
 * timed region: ref_mm_open -> ref_mm_run (timed, repeatable) -> ref_mm_close. */
struct ref_mm {
    struct ggml_context *ctx;
    struct ggml_cgraph gf;
    struct ggml_tensor *ty;
    size_t ybytes;
};

void *ref_mm_open(int type, const void *W, const float *x, int M, int K, int N, int n_threads) {
    if (type != GGML_TYPE_Q4_0 && type != GGML_TYPE_Q4_1) return NULL;
    const size_t wbytes = (size_t)M * (size_t)(K / 32) * ggml_type_size((enum ggml_type)type);
    const size_t xbytes = (size_t)N * K * sizeof(float);
    const size_t ybytes = (size_t)N * M * sizeof(float);
    struct ggml_init_params ip;
    memset(&ip, 0, sizeof ip);
    ip.mem_size = wbytes + xbytes + ybytes + (size_t)N * (K / 32) * 40 * 2 + (64u << 20);
    struct ref_mm *h = (struct ref_mm *)calloc(1, sizeof *h);
    if (!h) return NULL;
    h->ctx = ggml_init(ip);
    if (!h->ctx) { free(h); return NULL; }
    struct ggml_tensor *tw = ggml_new_tensor_2d(h->ctx, (enum ggml_type)type, K, M);
    struct ggml_tensor *tx = ggml_new_tensor_2d(h->ctx, GGML_TYPE_F32, K, N);
    memcpy(tw->data, W, wbytes);
    memcpy(tx->data, x, xbytes);
    h->ty = ggml_mul_mat(h->ctx, tw, tx);
    h->gf = ggml_build_forward(h->ty);
    h->gf.n_threads = n_threads;
    h->ybytes = ybytes;
    return h;
}

void ref_mm_run(void *vh) {
    struct ref_mm *h = (struct ref_mm *)vh;
    ggml_graph_compute(h->ctx, &h->gf);
}

void ref_mm_read(void *vh, float *y) {
    struct ref_mm *h = (struct ref_mm *)vh;
    memcpy(y, h->ty->data, h->ybytes);
}

void ref_mm_close(void *vh) {
    struct ref_mm *h = (struct ref_mm *)vh;
    if (!h) return;
    ggml_free(h->ctx);
    free(h);
}

/* LoRA merge as the reference does it (lib/llama.cpp:872-878): BA = ggml_mul_mat(loraA[r,K], loraB[r,M]) (f32 x f32 ->
 * ggml_compute_forward_mul_mat_f32 -> ggml_vec_dot_f32), then ggml_add_inplace(W_q4, BA) -> ggml_compute_forward_add_q_f32
 * (lib/ggml.c:6414): dequantize_row_q, ggml_vec_acc_f32, quantize_row_q (the SIMD quantizer).
 * a: K rows of r floats (NULL when ba is given), b: M rows of r floats, ba: optional M rows of K floats (cached adapter).
 * W (M rows of K/32 AoS blocks) is updated in place; ba_out (optional, M*K floats) receives the BA that was added. */
int ref_lora_add(int type, void *W, const float *a, const float *b, const float *ba, int r, int M, int K, float sign,
                 float *ba_out, int n_threads) {
    if (type != GGML_TYPE_Q4_0 && type != GGML_TYPE_Q4_1) return -1;
    const size_t wbytes = (size_t)M * (size_t)(K / 32) * ggml_type_size((enum ggml_type)type);
    const size_t need = wbytes + 3 * (size_t)M * K * 4 + (size_t)(M + K) * (r > 0 ? r : 1) * 4 + (size_t)n_threads * (K + 64) * 8 + (64u << 20);
    struct ggml_init_params ip;
    memset(&ip, 0, sizeof ip);
    ip.mem_size = need;
    struct ggml_context *ctx = ggml_init(ip);
    if (!ctx) return -2;
    struct ggml_tensor *tw = ggml_new_tensor_2d(ctx, (enum ggml_type)type, K, M);
    memcpy(tw->data, W, wbytes);
    struct ggml_tensor *tba;
    if (ba) {
        tba = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, M);
        memcpy(tba->data, ba, (size_t)M * K * 4);
    } else {
        struct ggml_tensor *ta = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, r, K);
        struct ggml_tensor *tb = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, r, M);
        memcpy(ta->data, a, (size_t)K * r * 4);
        memcpy(tb->data, b, (size_t)M * r * 4);
        tba = ggml_mul_mat(ctx, ta, tb);
    }
    struct ggml_tensor *delta = tba;
    if (sign != 1.0f) delta = ggml_scale(ctx, tba, ggml_new_f32(ctx, sign));     /* detach: lib/llama.cpp:933-934 */
    struct ggml_tensor *res = ggml_add_inplace(ctx, tw, delta);
    struct ggml_cgraph gf = ggml_build_forward(res);
    gf.n_threads = n_threads;
    ggml_graph_compute(ctx, &gf);
    memcpy(W, tw->data, wbytes);
    if (ba_out) memcpy(ba_out, tba->data, (size_t)M * K * 4);
    ggml_free(ctx);
    return 0;
}
