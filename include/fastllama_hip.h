/*
 * fastllama_hip.h -- kernel-level C-ABI of libfastllama_hip.so (MI355X / gfx950).
 *
 * This is the SECONDARY (operator) boundary of SURVEY.md section 8(b): a device-side replacement for the
 * reference's `quantize_fns_t` plug-in table and for `ggml_compute_forward_mul_mat_q_f32`.
 * The PRIMARY boundary (the 17 `llama_*` symbols of interfaces/c/fastllama.h) is declared in
 * include/fastllama.h and exported by the same shared object.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - every pointer named *_dev is a DEVICE pointer (hipMalloc / torch.cuda memory); *_host is host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream, which is also torch's default).
 *   - functions return FL_OK (0) or a negative FL_E* code; fl_last_error() holds the message.
 *     (The reference's row functions return void and GGML_ASSERT -> abort(), lib/ggml.c:138-144; an error
 *     code is the only deliberate deviation, see INTEGRATION.md.)
 *   - one caller thread per device at a time, exactly like the reference (no internal locking,
 *     SURVEY.md section 8(b) "Threading").
 *
 * All "replaces" citations are file:line into the reference tree (/root/reference).
 */
#ifndef FASTLLAMA_HIP_H
#define FASTLLAMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations ARE its export list */

#define FL_OK 0
#define FL_EINVAL (-1)  /* bad argument (shape, alignment, type)                        */
#define FL_EHIP (-2)    /* a HIP runtime call failed                                     */
#define FL_ENOMEM (-3)  /* device or host allocation failed                              */
#define FL_ENODEV (-4)  /* no gfx950 device / HIP runtime unavailable -- never a CPU fallback */

#define FL_TYPE_Q4_0 2 /* enum ggml_type GGML_TYPE_Q4_0, include/ggml.h:200-212 */
#define FL_TYPE_Q4_1 3 /* GGML_TYPE_Q4_1 */

/* ---------------------------------------------------------------- runtime ---------------------- */
/* replaces: nothing in the reference (CPU only); hook point is FastLlama::Params::build,
 * lib/bridge.cpp:110-150 ("bridge.cpp device init" in BASELINE.json). */
int fl_device_count(void);
int fl_init(int device);                 /* hipSetDevice + sanity check that the device is gfx950 */
const char *fl_last_error(void);
/* Warnings -- events that are not errors but that a caller should hear about (a derived operand copy that did not fit in device memory and the
 * slower kernel family that runs instead): one line each to `cb`; NULL (the default) prints them to stderr. */
void fl_set_warn_handler(void (*cb)(const char *line));
int fl_device_name(char *buf, size_t n);
const char *fl_version(void);

void *fl_malloc(size_t bytes);           /* device memory; NULL on failure */
int fl_free(void *p_dev);
int fl_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int fl_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int fl_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int fl_memset(void *dst_dev, int value, size_t bytes, void *stream);
int fl_stream_synchronize(void *stream);
void *fl_stream_create(void);
int fl_stream_destroy(void *stream);

/* HIP events on the stream the kernels are launched on (bench.py's live timing) */
void *fl_event_create(void);
int fl_event_destroy(void *ev);
int fl_event_record(void *ev, void *stream);
int fl_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises `stop` */

/* ---------------------------------------------------------------- weights ---------------------- */
/* A Q4_0 / Q4_1 matrix W[M rows][K], resident in HBM in the repacked QW16 layout
 * (fastllama_amd/csrc/q4_layout.h).  Input is the reference's own byte layout: M rows of K/32
 * consecutive block_q4_0 (20 B) / block_q4_1 (24 B), lib/ggml.c:590-603 -- i.e. a GGML/GGJT tensor
 * payload as is.  The repack is lossless: fl_qtensor_download returns the input bytes.
 * replaces: the host-resident `ggml_tensor` weights created in Model::load, lib/llama.cpp:223-258. */
typedef struct fl_qtensor fl_qtensor;
fl_qtensor *fl_qtensor_upload(int type, const void *blocks_host, int M, int K, void *stream);
fl_qtensor *fl_qtensor_from_device(int type, const void *blocks_dev, int M, int K, void *stream);
int fl_qtensor_download(const fl_qtensor *W, void *blocks_host, void *stream);
int fl_qtensor_info(const fl_qtensor *W, int *type, int *M, int *K);
size_t fl_qtensor_device_bytes(const fl_qtensor *W);
void fl_qtensor_free(fl_qtensor *W);
/* The reference-order prefill GEMM's f16 fragment copy of the weights (q4_layout.h "H16 copies", 64 bytes per row and block):
 * the integers of the nibbles, laid out as the v_mfma_f32_32x32x4_2b_f16 A fragments whose results ARE the AVX2 lane sums of
 * ggml_vec_dot_q4_{0,1}_q8_0 (lib/ggml.c:2445-2487).  A model builds the copies of its tensors on the first reference-order eval
 * with N >= 9 (fl_model_eval); a caller that rewrites a tensor's blocks in place rebuilds it.  Derived data -- download, decode and state files never see it. */
int fl_qtensor_build_h16(fl_qtensor *W, void *stream);
void fl_qtensor_drop_h16(fl_qtensor *W);
/* The reference-order DECODE kernel's copy of the nibbles (q4_layout.h "QWD", 16 bytes per row and block once more): a lane owns two of
 * a row's eight AVX2-lane chains and reads its dword of four blocks in one load.  Built by a model before its first reference-order
 * single-token eval; same rules as the H16 copy. */
int fl_qtensor_build_qwd(fl_qtensor *W, void *stream);
void fl_qtensor_drop_qwd(fl_qtensor *W);

/* ---------------------------------------------------------------- quantize_fns_t mirror -------- */
/* Same five entry points, same argument meaning as `quantize_fns_t` (include/ggml.h:850-862,
 * table lib/ggml.c:1731-1767), operating on DEVICE buffers that use the reference's AoS block
 * layouts, so outputs can be diffed byte-for-byte against the CPU functions.
 *
 *   dequantize_row_q(x, y, k)   replaces dequantize_row_q4_0 / _q4_1   lib/ggml.c:1443,1561
 *   quantize_row_q_dot(x, y, k) replaces quantize_row_q8_0             lib/ggml.c:1299
 *   vec_dot_q(n, s, x, y)       replaces ggml_vec_dot_q4_0_q8_0 / _q4_1_q8_0  lib/ggml.c:2368,2561
 *
 *   quantize_row_q(x, y, k)            replaces quantize_row_q4_0 / _q4_1 (the SIMD flavour the reference's x86 build
 *                                      runs: id = 7/amax for Q4_0, halves to even)            lib/ggml.c:666,958
 *   quantize_row_q_reference(x, y, k)  replaces quantize_row_q4_0_reference / _q4_1_reference (id = 1/d, roundf), the
 *                                      ones ggml_quantize_q4_0/1 build model files with      lib/ggml.c:630,917 */
int fl_quantize_row_q4_0(const float *x_dev, void *y_dev /* block_q4_0[k/32] */, int k, void *stream);
int fl_quantize_row_q4_1(const float *x_dev, void *y_dev /* block_q4_1[k/32] */, int k, void *stream);
int fl_quantize_row_q4_0_reference(const float *x_dev, void *y_dev, int k, void *stream);
int fl_quantize_row_q4_1_reference(const float *x_dev, void *y_dev, int k, void *stream);
int fl_quantize_row_q8_0(const float *x_dev, void *y_dev /* block_q8_0[k/32] */, int k, void *stream);
int fl_dequantize_row_q4_0(const void *x_dev /* block_q4_0[k/32] */, float *y_dev, int k, void *stream);
int fl_dequantize_row_q4_1(const void *x_dev /* block_q4_1[k/32] */, float *y_dev, int k, void *stream);
int fl_vec_dot_q4_0_q8_0(int n, float *s_dev, const void *x_dev, const void *y_dev, void *stream);
int fl_vec_dot_q4_1_q8_0(int n, float *s_dev, const void *x_dev, const void *y_dev, void *stream);

typedef void (*fl_dequantize_row_q_t)(const void *x_dev, float *y_dev, int k);
typedef void (*fl_quantize_row_q_t)(const float *x_dev, void *y_dev, int k);
typedef void (*fl_vec_dot_q_t)(const int n, float *s_dev, const void *x_dev, const void *y_dev);
typedef struct {
    fl_dequantize_row_q_t dequantize_row_q;
    fl_quantize_row_q_t quantize_row_q;
    fl_quantize_row_q_t quantize_row_q_reference;
    fl_quantize_row_q_t quantize_row_q_dot;
    fl_vec_dot_q_t vec_dot_q;
} fl_quantize_fns_t;
/* replaces ggml_internal_get_quantize_fn, lib/ggml.c:1769-1773.  Table entries run on the null
 * stream and synchronise it before returning (they mirror synchronous CPU functions). */
fl_quantize_fns_t fl_get_quantize_fn(size_t type);

/* ---------------------------------------------------------------- the op ----------------------- */
/* Quantized activations (the reference's `params->wdata` Q8_0 scratch, lib/ggml.c:8105-8119,
 * sized at :10949).  INIT and COMPUTE are separate entry points so that one quantized x feeds
 * wq/wk/wv (and w1/w3), which the reference re-quantizes per matmul. */
typedef struct fl_qact fl_qact;
fl_qact *fl_qact_create(int max_N, int K);
void fl_qact_free(fl_qact *a);
/* INIT phase: x_dev is N rows of K floats, row stride ldx elements (16-byte aligned rows). */
int fl_quantize_q8(fl_qact *a, const float *x_dev, int ldx, int N, int K, void *stream);
/* copy the workspace out as the reference's block_q8_0[N][K/32] (parity tests) */
int fl_qact_export(const fl_qact *a, void *blocks_dev, void *stream);
/* COMPUTE phase: y_dev[n*ldy + m] = vec_dot_q(K, W row m, q8 row n), n < N, m < M.
 * N <= 8 runs the wave-dot GEMV, N >= 9 the exact-integer MFMA GEMM. */
int fl_mul_mat_q(const fl_qtensor *W, const fl_qact *a, float *y_dev, int ldy, void *stream);

/* replaces ggml_compute_forward_mul_mat_q_f32 (lib/ggml.c:7928-8176) = INIT + COMPUTE, for
 * `dst = ggml_mul_mat(W, x)`: x_dev is N rows of K floats (ne10 = K, ne11 = N), y_dev is N rows of M
 * floats (ne0 = M, ne1 = N), both contiguous-row with the given strides.  Uses an internal
 * grow-on-demand workspace (do not call while a hipGraph capture is active). */
int fl_mul_mat_q_f32(const fl_qtensor *W, const float *x_dev, int ldx, float *y_dev, int ldy, int N, void *stream);


/* ---------------------------------------------------------------- collectives (RCCL over xGMI) -- */
/* New for SURVEY.md section 8(e): the reference is single-process.  One process per GPU; rank 0 creates
 * the id, the host program distributes its FL_COMM_ID_BYTES bytes to the other ranks. */
#define FL_COMM_ID_BYTES 128
typedef struct fl_comm fl_comm;
int fl_comm_unique_id(void *id_out /* FL_COMM_ID_BYTES */);
fl_comm *fl_comm_create(const void *id_bytes, int rank, int world); /* on the CURRENT device (fl_init) */
/* `world` shards driven by ONE process on the current device (one host thread per shard): out[r] is rank r's handle.
 * The all-reduce is a host rendezvous + one device kernel (rank-order sum).  For single-GPU validation of the
 * tensor-parallel path and for hosts that keep several shards in one process. */
#define FL_COMM_MAX_LOCAL 8
int fl_comm_create_local(int world, fl_comm **out /* [world] */);
int fl_comm_allreduce_sum_f32(fl_comm *c, float *buf_dev, size_t count, void *stream);
/* recv_dev[r * count + i] <- rank r's send_dev[i] (the logits slices of the row-split lm-head) */
int fl_comm_allgather_f32(fl_comm *c, const float *send_dev, size_t count, float *recv_dev, void *stream);
int fl_comm_is_local(const fl_comm *c);
/* Small messages (<= 64 KB: the partial sums and logits slices of a decode token) skip the ring collective: every rank owns an
 * exchange buffer its peers map (hipIpc, xGMI), and ONE kernel per rank publishes, waits for and adds -- in rank order -- the
 * G vectors.  fl_comm_create sets this up by itself (round 5: on by default; FL_P2P=0 in the environment: RCCL only): the handles travel
 * through RCCL, then a self-test moves patterned slices between all ranks -- the exchange has only ever run between processes on ONE GPU,
 * so any failure of the hand-over or of the self-test, on any rank, leaves a plain RCCL communicator on every rank.  The same buffers carry
 * the FOLD REGIONS of a row-split tensor-parallel model's decode: the four exchanges of a layer are the tails of the launches that produce
 * the data (csrc/tp_tail.h, fl_model_tp_folded) -- five launches per layer and no collective launch.  A host that
 * moves the handles itself: fl_comm_create_p2p on every rank, fl_comm_p2p_export -> gather the FL_COMM_P2P_HANDLE_BYTES of all
 * ranks in rank order -> fl_comm_p2p_import.  Such a communicator has no RCCL behind it: larger messages fail. */
#define FL_COMM_P2P_HANDLE_BYTES 128
fl_comm *fl_comm_create_p2p(int rank, int world);
int fl_comm_p2p_export(fl_comm *c, void *handles_out /* FL_COMM_P2P_HANDLE_BYTES */);
int fl_comm_p2p_import(fl_comm *c, const void *handles_all /* world * FL_COMM_P2P_HANDLE_BYTES, rank order */);
int fl_comm_has_p2p(const fl_comm *c);
/* COLLECTIVE (every rank, after the import): patterned slices through the exchange between all ranks, bounded waits -- the check fl_comm_create
 * itself runs before it keeps the exchange (a failure there leaves the communicator on RCCL alone). */
int fl_comm_p2p_selftest(fl_comm *c);
int fl_comm_p2p_timeouts(const fl_comm *c); /* exchanges that gave up waiting for a peer (~20 s each); 0 on a healthy group */
int fl_comm_p2p_check(fl_comm *c);          /* FL_EHIP if that count advanced since the last check (fl_model_eval calls it after every
                                              * synchronised tensor-parallel eval: such an eval's results are invalid); caller has synchronised */
int fl_comm_debug_graph_allreduce(fl_comm *c, float *buf_dev, size_t count, int replays, void *stream);
int fl_comm_rank(const fl_comm *c);
int fl_comm_size(const fl_comm *c);
int fl_comm_rccl_ranks(const fl_comm *c); /* ncclCommCount of the RCCL communicator; 0 = none behind this handle */
void fl_comm_destroy(fl_comm *c);

/* ---------------------------------------------------------------- the model -------------------- */
/* Device-resident replacement of fastllama::Model for the eval path.
 * replaces: Model::load tensor placement lib/llama.cpp:223-258, KVCacheBuffer lib/llama.cpp:24-51,
 *           Model::eval lib/llama.cpp:272-499 (and with it the ggml_graph_compute dispatch, lib/ggml.c:10811). */
typedef struct fl_model fl_model;
typedef struct {
    int n_vocab, n_embd, n_head, n_layer, n_ff; /* hparams, lib/llama.cpp:118-129 (n_ff already derived) */
    int n_ctx;                                  /* KV-cache positions                                    */
    int qtype;                                  /* FL_TYPE_Q4_0 | FL_TYPE_Q4_1 of all 2-D tensors         */
    int max_batch;                              /* largest N of one eval (n_batch)                        */
    int tp_rank, tp_size;                       /* tensor parallel shard of this process (0, 1 = none)    */
} fl_model_params;
fl_model *fl_model_create(const fl_model_params *p);
/* Feed one tensor with the name/shape/bytes it has in a GGML/GGMF/GGJT llama file (lib/llama.cpp:223-246):
 * type 0 = f32 (norm vectors), qtype for every 2-D weight; ne0 = row length K, ne1 = rows.  The FULL tensor is
 * passed on every rank; the tensor-parallel slice is taken here.  `data` may be a host OR a device pointer. */
int fl_model_set_tensor(fl_model *m, const char *name, int type, const void *data, int ne0, int ne1);
int fl_model_finalize(fl_model *m); /* checks completeness, allocates KV cache, tables, work buffers */
int fl_model_set_comm(fl_model *m, fl_comm *c);
/* Model::eval: N tokens at positions n_past..n_past+N-1.  logits_host receives n_vocab floats (last token) or
 * N*n_vocab (all_logits != 0, `should_put_all_logits`); embeddings_host (optional) n_embd floats of the last token. */
int fl_model_eval(fl_model *m, const int32_t *tokens_host, int N, int n_past, float *logits_host, int all_logits,
                  float *embeddings_host);
/* The consecutive evals of a long prompt (the session's ingest loop, lib/bridge.cpp:186-238, one llama_eval per n_batch
 * chunk): chunk c = the next chunk_len[c] tokens, evaluated at n_past + (tokens before it).  Same results as fl_model_eval
 * chunk by chunk, bit for bit; two chunks are in flight at once on two streams (chunk c+1 waits for chunk c layer by layer,
 * only where its attention reads the K/V cache) and only the last chunk runs the lm-head.  logits_host: n_vocab floats of the
 * last token, or NULL. */
int fl_model_ingest(fl_model *m, const int32_t *tokens_host, const int *chunk_len, int n_chunks, int n_past, float *logits_host);
/* enable: 1 = reset + start timing every quantized-matmul launch (HIP events on the eval stream), 0 = stop; the
 * accumulated totals so far are returned through the two pointers (either may be NULL). */
int fl_model_profile(fl_model *m, int enable, double *mm_ms_total, long *mm_launches);
/* decode (N = 1) evals replay a captured hipGraph by default; 0 switches to plain launches */
int fl_model_set_graph(fl_model *m, int mode); /* bit0: hipGraph replay for decode (default on); bit1: generic per-op decode kernels; bit2: three-kernel prefill attention;
                                                * bit3: decode attention never split; bit4: always split (default: two launches from position 256 on) */
/* Summation order of the matmuls.  1 ("exact"): the reference's own order -- 8 f32 lane accumulators per Q4 x Q8_0 dot in block
 * order and the AVX2 horizontal sum (lib/ggml.c:2445-2487, :2639-2689), ggml_vec_dot_f32's 4 x 8 lanes in the attention matmuls
 * (lib/ggml.c:2295-2330) -- logits bit-identical to the reference's x86 build.  0 ("fast"): exact integer block dots on the MFMA
 * units, block terms added in the kernels' own f32 order (1e-7 per matmul; ~1e-2 on the logits after 32 layers). */
int fl_model_set_exact(fl_model *m, int on);
int fl_model_get_exact(const fl_model *m);
int fl_default_exact(void); /* the mode new models start in: environment FL_EXACT=1|0, else the library default */
/* The reference-order kernels read derived copies of the weights (q4_layout.h): WH16 for evals with N >= 9 (4 x the nibble bytes: 13 GB at 7B,
 * 130 GB at 65B) and QWD for single-token evals (1 x).  By default they are built inside the first eval that needs them; fl_model_prepare
 * builds them NOW (flags bit 0: WH16, bit 1: QWD), so that no timed or latency-sensitive eval pays for it.  Returns FL_OK also when a copy does
 * not fit: the warning handler is told how many bytes were missing and the kernel family that reads the primary layout runs instead (same bits,
 * ~1.3-1.45 x the time).  fl_model_prepared: which copies are resident (same bits; a negative state = tried and dropped reads as 0).
 * flags bit 2 (value 4) -- LEAN memory mode: from now on the model holds exactly the copies named in this call (a decode-only session, flags 2 | 4,
 * never builds the 13 GB of WH16: its prefill runs the nibble-operand kernel, same bits), and when BOTH copies are resident the QW16 nibble planes of
 * the matmul tensors -- which nothing in the reference-order path reads then -- are freed (3.3 GB at 7B).  They come back from the QWD copy, bit for bit,
 * before anything that reads them: fl_model_set_exact(m, 0), a LoRA merge, fl_model_tensor_download, a later fl_model_prepare.
 * fl_model_memory: resident bytes by kind: [0] QW16 nibble planes, [1] scale planes, [2] WH16, [3] QWD, [4] everything else (K / V, work buffers). */
int fl_model_prepare(fl_model *m, int flags);
int fl_model_prepared(const fl_model *m);
int fl_model_memory(const fl_model *m, size_t *bytes5);
/* Kernel nodes of the decode hipGraph captured last = launches per decode token (0: none captured yet). */
int fl_model_graph_nodes(const fl_model *m);
/* 1: the decode exchanges of this row-split tensor-parallel model are the tails of the launches that produce the data (peer-mapped
 * fold regions of the communicator; five launches per layer and no collective launch), decided at the first single-token eval;
 * 0: the collective sequence (no peer-mapped exchange behind the communicator, FL_TP_FOLD=0, or not a row-split model). */
int fl_model_tp_folded(const fl_model *m);
/* LoRA on the resident Q4 weights -- replaces Model::attach_lora / detach_lora (lib/llama.cpp:697-944) and its
 * ggml_compute_forward_add_q_f32 (lib/ggml.c:6414-6520): W <- quantize_row_q(dequantize_row_q(W) + sign * BA), with the
 * reference's SIMD quantizer arithmetic.  base_name is the base tensor ("layers.3.attention.wq.weight"); pass either
 * ba [ne1][ne0] f32 (cached adapter) or a [ne0][r] + b [ne1][r] (BA[m][k] = ggml_vec_dot_f32(r, a_k, b_m)); host or
 * device pointers, FULL tensors also under tensor parallelism.  keep_backup: save the tensor's current contents on its
 * first touch so fl_model_lora_restore() can put the originals back (the reference's use_mmap detach, :714-726);
 * without it detach is a second apply with sign = -1 (:933-938). */
int fl_model_lora_shape(fl_model *m, const char *base_name, int *ne0, int *ne1);
int fl_model_lora_apply(fl_model *m, const char *base_name, const float *ba, const float *a, const float *b, int r, float sign,
                        int keep_backup);
int fl_model_lora_restore(fl_model *m);
/* this rank's rows of a base tensor as reference AoS blocks (block_q4_0 / block_q4_1), for tests and tooling */
int fl_model_tensor_download(fl_model *m, const char *base_name, void *aos_host);
/* test hooks (teacher-forced per-layer parity): layers [l0, l1) on a caller-provided layer input; then the Q8_0 operand the
 * last layer fed to wo (which = 0), w1|w3 (1) or w2 (2), as block_q8_0 rows */
int fl_model_debug_layers(fl_model *m, int l0, int l1, const float *x_host, int N, int n_past, float *x_out_host);
int fl_model_debug_export_q8(fl_model *m, int which, int N, void *blocks_host);
const float *fl_model_logits_dev(const fl_model *m); /* rows of fl_model_logits_ld() floats (n_vocab rounded up to 4) */
int fl_model_logits_ld(const fl_model *m);
int fl_model_logits_read(fl_model *m, int row0, int rows, float *logits_host);
/* -log softmax(logits[row0 + i])[next_tokens_host[i]] on the device (FastLlama::perplexity's row loop, lib/bridge.cpp:397-407) */
int fl_model_logits_nll(fl_model *m, int row0, int rows, const int32_t *next_tokens_host, double *nll_host);
void *fl_model_stream(const fl_model *m);
size_t fl_model_device_bytes(const fl_model *m);
int fl_model_kv_read(const fl_model *m, float *k_host, float *v_host);
int fl_model_kv_write(fl_model *m, const float *k_host, const float *v_host);
void fl_model_free(fl_model *m);

/* Operator-level entry points (fl_mul_mat_q, fl_mul_mat_q_f32): 1 = the reference's summation order, 0 = the fast kernels, -1 = the
 * library default (fl_default_exact(): reference order unless FL_FAST=1). */
int fl_set_op_mode(int mode);
/* fl_quantize_q8 with the workspace layout chosen by the caller (16 = QA16, the GEMM's; 1 = QA1, the GEMV's) instead of by N */
int fl_quantize_q8_layout(fl_qact *a, const float *x_dev, int ldx, int N, int K, int layout, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* FASTLLAMA_HIP_H */
