// capi.cpp -- kernel-level C-ABI (include/fastllama_hip.h) on top of the launchers in q4_kernels.h.
// Host-side only: argument checking, device memory ownership, error reporting.  There is no CPU
// compute path anywhere in this library: without a HIP device every entry point fails with FL_ENODEV.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/fastllama_hip.h"
#include "q4_kernels.h"
#include "eval_kernels.h"
#include "runtime.h"
#include "internal.h"

namespace fl {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// Warnings: things a caller should know about but that are not errors (a derived operand copy that did not fit and the slower kernel
// family that runs instead).  One line per event to the handler of fl_set_warn_handler -- stderr unless the embedder installs its own
// (llama_api.cpp routes them to the session's logger).
static void (*g_warn_cb)(const char *) = nullptr;
void warn(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g_warn_cb) g_warn_cb(buf);
    else fprintf(stderr, "fastllama_hip: warning: %s\n", buf);
}

int hip_fail(hipError_t e, const char *what) {
    return set_error(e == hipErrorOutOfMemory ? FL_ENOMEM : FL_EHIP, "%s: %s", what, hipGetErrorString(e));
}

static bool g_inited = false;

int ensure_device() {
    if (g_inited) return FL_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(FL_ENODEV, "no HIP device visible (%s); libfastllama_hip has no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    g_inited = true;
    return FL_OK;
}

}  // namespace fl

namespace fl { extern int g_gemm_force_cfg; extern int g_gemv_force_waves; extern int g_stream_min_groups; extern int g_stream_force_nw; extern int g_pv_waves; extern int g_stream_helpers; }
using namespace fl;

#define FL_HIP(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

static inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

// (struct fl_qact_impl, op_exact, ensure_h16, check_mm, mul_mat_q_which: internal.h -- shared with test_hooks.cpp)
namespace fl {
// Which arithmetic the operator-level entry points (fl_mul_mat_q, fl_mul_mat_q_f32) run: the reference's summation order (default, as
// for models: fl_default_exact()) or the fast kernels.  fl_set_op_mode(1 | 0) pins it for a process, -1 returns to the default.
int g_op_mode = -1;
bool op_exact() { return g_op_mode < 0 ? fl_default_exact() != 0 : g_op_mode != 0; }
// The derived weight copies are built on first use by the operator-level entry points (fl_mul_mat_q takes a const tensor: the copies are
// logically const).  The build allocates, launches on the caller's stream and synchronises it ONCE per tensor; the mutex makes a tensor
// shared between host threads safe.  Callers that cannot take the one-time synchronisation (stream capture) build the copies first:
// fl_qtensor_build_h16 / fl_qtensor_build_qwd, or fl_model_prepare for a model.
static std::mutex g_lazy_mu;
int ensure_h16(const fl_qtensor *W, fl_qact_impl *a, void *st) {
    if (!__atomic_load_n(&W->h16, __ATOMIC_ACQUIRE)) {     // (set = complete: fl_qtensor_build_h16 publishes the pointer behind the synchronisation)
        std::lock_guard<std::mutex> lk(g_lazy_mu);
        const int rc = W->h16 ? FL_OK : fl_qtensor_build_h16(const_cast<fl_qtensor *>(W), st);      // (a derived copy: logically const)
        if (rc != FL_OK) return rc;
    }
    if (!a->h16_valid) {
        FL_HIP(qa16_to_h16(*a, a->N, S(st)));
        a->h16_valid = 1;
    }
    return FL_OK;
}
}  // namespace fl


extern "C" {

const char *fl_version(void) { return "fastllama_hip 0.1 (gfx950)"; }
const char *fl_last_error(void) { return g_err; }
void fl_set_warn_handler(void (*cb)(const char *line)) { g_warn_cb = cb; }

int fl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fl_init(int device) {
    int rc = ensure_device();
    if (rc != FL_OK) return rc;
    FL_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    FL_HIP(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return set_error(FL_ENODEV, "device %d is %s; this library is built for gfx950 only", device, p.gcnArchName);
    return FL_OK;
}

int fl_device_name(char *buf, size_t n) {
    int rc = ensure_device();
    if (rc != FL_OK) return rc;
    int dev = 0;
    FL_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    FL_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, n, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return FL_OK;
}

void *fl_malloc(size_t bytes) {
    if (ensure_device() != FL_OK) return nullptr;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        hip_fail(e, "hipMalloc");
        return nullptr;
    }
    return p;
}
int fl_free(void *p) {
    if (!p) return FL_OK;
    FL_HIP(hipFree(p));
    return FL_OK;
}
int fl_memcpy_h2d(void *d, const void *s, size_t n, void *st) {
    FL_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, S(st)));
    FL_HIP(hipStreamSynchronize(S(st)));
    return FL_OK;
}
int fl_memcpy_d2h(void *d, const void *s, size_t n, void *st) {
    FL_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, S(st)));
    FL_HIP(hipStreamSynchronize(S(st)));
    return FL_OK;
}
int fl_memcpy_d2d(void *d, const void *s, size_t n, void *st) {
    FL_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, S(st)));
    return FL_OK;
}
int fl_memset(void *d, int v, size_t n, void *st) {
    FL_HIP(hipMemsetAsync(d, v, n, S(st)));
    return FL_OK;
}
int fl_stream_synchronize(void *st) {
    FL_HIP(hipStreamSynchronize(S(st)));
    return FL_OK;
}
void *fl_stream_create(void) {
    if (ensure_device() != FL_OK) return nullptr;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}
int fl_stream_destroy(void *st) {
    FL_HIP(hipStreamDestroy(S(st)));
    return FL_OK;
}
void *fl_event_create(void) {
    if (ensure_device() != FL_OK) return nullptr;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
int fl_event_destroy(void *ev) {
    FL_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return FL_OK;
}
int fl_event_record(void *ev, void *st) {
    FL_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), S(st)));
    return FL_OK;
}
int fl_event_elapsed_ms(void *a, void *b, float *ms) {
    FL_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(b)));
    FL_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(a), reinterpret_cast<hipEvent_t>(b)));
    return FL_OK;
}

/* ------------------------------------------------------------------ weights ------------------- */
static int check_wshape(int type, int M, int K) {
    if (type != FL_TYPE_Q4_0 && type != FL_TYPE_Q4_1) return set_error(FL_EINVAL, "type %d is not Q4_0/Q4_1", type);
    if (M <= 0 || K <= 0 || K % FL_QK != 0) return set_error(FL_EINVAL, "bad shape M=%d K=%d (K %% 32 != 0)", M, K);
    return FL_OK;
}

fl_qtensor *fl_qtensor_from_device(int type, const void *blocks_dev, int M, int K, void *stream) {
    if (ensure_device() != FL_OK || check_wshape(type, M, K) != FL_OK) return nullptr;
    fl_qtensor *W = new (std::nothrow) fl_qtensor();
    if (!W) return nullptr;
    W->type = type;
    W->M = M;
    W->K = K;
    W->M16 = fl_roundup(M, 16);
    W->KB = K / FL_QK;
    W->owns = 1;
    const size_t nblk = (size_t)W->M16 * W->KB;
    int *flag = nullptr;
    hipError_t e = hipMalloc((void **)&W->qs, nblk * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&W->d, nblk * 4);
    if (e == hipSuccess && type == FL_TYPE_Q4_1) e = hipMalloc((void **)&W->m, nblk * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&flag, 4);
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, 4, S(stream));
    if (e == hipSuccess) e = repack_to_qw16(type, blocks_dev, M, K, W->qs, W->d, W->m, flag, S(stream));
    int bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (flag) (void)hipFree(flag);
    if (e != hipSuccess) {
        hip_fail(e, "fl_qtensor_from_device");
        fl_qtensor_free(W);
        return nullptr;
    }
    if (bad) {
        set_error(FL_EINVAL, (bad & 2) ? "non-finite block scale (not a valid model)"
                                       : "Q4_0 block scale below 2^-122: d/16 would be subnormal (not a valid model)");
        fl_qtensor_free(W);
        return nullptr;
    }
    return W;
}

/* (re)build the f16 fragment copy the reference-order prefill GEMM reads (q4_layout.h "H16 copies"; 4 x the nibble bytes) */
// FL_TEST_FAIL_DERIVED=1|2|3 (tests only): the WH16 (bit 0) / QWD (bit 1) allocation fails as if device memory had run out -- the fallback and
// its warning cannot be exercised otherwise without filling 288 GB
static bool test_fail_derived(int bit) {
    const char *e = getenv("FL_TEST_FAIL_DERIVED");
    return e && (atoi(e) & bit) != 0;
}

int fl_qtensor_build_h16(fl_qtensor *W, void *stream) {
    if (!W) return set_error(FL_EINVAL, "null tensor");
    if (!W->h16 && test_fail_derived(1)) return set_error(FL_ENOMEM, "hipMalloc(WH16): out of memory (FL_TEST_FAIL_DERIVED)");
    // A first build goes into a LOCAL pointer and is published -- a release store -- only once the copy is complete and the stream synchronised:
    // another host thread that finds the pointer set (an acquire load, ensure_h16) may launch on it at once.  A rebuild (LoRA merge:
    // the tensor's owner changes the weights, nobody else is using the tensor) overwrites the published copy in place.
    uint16_t *dst = W->h16;
    if (!dst) {
        hipError_t ea = hipMalloc((void **)&dst, wh16_bytes(*W));
        if (ea != hipSuccess) return hip_fail(ea, "hipMalloc(WH16)");
    }
    hipError_t e = qw16_to_h16(*W, dst, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        if (!W->h16) (void)hipFree(dst);
        return hip_fail(e, "fl_qtensor_build_h16");
    }
    __atomic_store_n(&W->h16, dst, __ATOMIC_RELEASE);
    return FL_OK;
}

/* (re)build the nibble copy the reference-order decode kernel reads (q4_layout.h "QWD"; the size of the nibbles again) */
int fl_qtensor_build_qwd(fl_qtensor *W, void *stream) {
    if (!W) return set_error(FL_EINVAL, "null tensor");
    if (!W->qwd && test_fail_derived(2)) return set_error(FL_ENOMEM, "hipMalloc(QWD): out of memory (FL_TEST_FAIL_DERIVED)");
    uint32_t *dst = W->qwd;                                // (published only when complete: see fl_qtensor_build_h16)
    if (!dst) {
        hipError_t ea = hipMalloc((void **)&dst, qwd_bytes(*W));
        if (ea != hipSuccess) return hip_fail(ea, "hipMalloc(QWD)");
    }
    hipError_t e = qw16_to_qwd(*W, dst, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        if (!W->qwd) (void)hipFree(dst);
        return hip_fail(e, "fl_qtensor_build_qwd");
    }
    __atomic_store_n(&W->qwd, dst, __ATOMIC_RELEASE);
    return FL_OK;
}
void fl_qtensor_drop_qwd(fl_qtensor *W) {
    if (W && W->qwd) (void)hipFree(W->qwd);
    if (W) W->qwd = nullptr;
}

void fl_qtensor_drop_h16(fl_qtensor *W) {
    if (W && W->h16) (void)hipFree(W->h16);
    if (W) W->h16 = nullptr;
}

fl_qtensor *fl_qtensor_upload(int type, const void *blocks_host, int M, int K, void *stream) {
    if (ensure_device() != FL_OK || check_wshape(type, M, K) != FL_OK) return nullptr;
    const size_t bytes = (size_t)M * (K / FL_QK) * (type == FL_TYPE_Q4_0 ? 20 : 24);
    void *tmp = nullptr;
    hipError_t e = hipMalloc(&tmp, bytes);
    if (e != hipSuccess) {
        hip_fail(e, "hipMalloc(staging)");
        return nullptr;
    }
    e = hipMemcpyAsync(tmp, blocks_host, bytes, hipMemcpyHostToDevice, S(stream));
    fl_qtensor *W = nullptr;
    if (e == hipSuccess) W = fl_qtensor_from_device(type, tmp, M, K, stream);
    else hip_fail(e, "hipMemcpy(H2D)");
    (void)hipFree(tmp);
    return W;
}

int fl_qtensor_download(const fl_qtensor *W, void *blocks_host, void *stream) {
    if (!W || !blocks_host) return set_error(FL_EINVAL, "null argument");
    const size_t bytes = (size_t)W->M * W->KB * (W->type == FL_TYPE_Q4_0 ? 20 : 24);
    void *tmp = nullptr;
    FL_HIP(hipMalloc(&tmp, bytes));
    hipError_t e = unpack_from_qw16(W->type, W->qs, W->d, W->m, W->M, W->K, tmp, S(stream));
    if (e == hipSuccess) e = hipMemcpyAsync(blocks_host, tmp, bytes, hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "fl_qtensor_download");
    return FL_OK;
}

int fl_qtensor_info(const fl_qtensor *W, int *type, int *M, int *K) {
    if (!W) return set_error(FL_EINVAL, "null tensor");
    if (type) *type = W->type;
    if (M) *M = W->M;
    if (K) *K = W->K;
    return FL_OK;
}

size_t fl_qtensor_device_bytes(const fl_qtensor *W) {
    if (!W) return 0;
    const size_t nblk = (size_t)W->M16 * W->KB;
    return nblk * (16 + 4 + (W->type == FL_TYPE_Q4_1 ? 4 : 0)) + (W->h16 ? wh16_bytes(*W) : 0) + (W->qwd ? qwd_bytes(*W) : 0);
}

void fl_qtensor_free(fl_qtensor *W) {
    if (!W) return;
    if (W->owns) {
        if (W->qs) (void)hipFree(W->qs);
        if (W->d) (void)hipFree(W->d);
        if (W->m) (void)hipFree(W->m);
    }
    if (W->h16) (void)hipFree(W->h16);
    if (W->qwd) (void)hipFree(W->qwd);
    delete W;
}

/* ------------------------------------------------------------------ row functions ------------- */
static int check_row(const void *a, const void *b, int k) {
    int rc = ensure_device();
    if (rc != FL_OK) return rc;
    if (!a || !b) return set_error(FL_EINVAL, "null pointer");
    if (k < 0 || k % FL_QK != 0) return set_error(FL_EINVAL, "k=%d is not a multiple of 32", k);  // assert(k % QK == 0)
    return FL_OK;
}

int fl_quantize_row_q8_0(const float *x, void *y, int k, void *st) {
    int rc = check_row(x, y, k);
    if (rc != FL_OK) return rc;
    if (reinterpret_cast<uintptr_t>(x) & 15) return set_error(FL_EINVAL, "x must be 16-byte aligned");
    FL_HIP(quantize_q8_aos(x, k, 1, k, y, S(st)));
    return FL_OK;
}
int fl_dequantize_row_q4_0(const void *x, float *y, int k, void *st) {
    int rc = check_row(x, y, k);
    if (rc != FL_OK) return rc;
    FL_HIP(dequantize_aos(FL_TYPE_Q4_0, x, y, k, S(st)));
    return FL_OK;
}
int fl_dequantize_row_q4_1(const void *x, float *y, int k, void *st) {
    int rc = check_row(x, y, k);
    if (rc != FL_OK) return rc;
    FL_HIP(dequantize_aos(FL_TYPE_Q4_1, x, y, k, S(st)));
    return FL_OK;
}
static int quant_q4(int type, bool reference, const float *x, void *y, int k, void *st) {
    int rc = check_row(x, y, k);
    if (rc != FL_OK) return rc;
    FL_HIP(quantize_row_q4_aos(type, reference, x, y, k, S(st)));
    return FL_OK;
}
int fl_quantize_row_q4_0(const float *x, void *y, int k, void *st) { return quant_q4(FL_TYPE_Q4_0, false, x, y, k, st); }
int fl_quantize_row_q4_1(const float *x, void *y, int k, void *st) { return quant_q4(FL_TYPE_Q4_1, false, x, y, k, st); }
int fl_quantize_row_q4_0_reference(const float *x, void *y, int k, void *st) { return quant_q4(FL_TYPE_Q4_0, true, x, y, k, st); }
int fl_quantize_row_q4_1_reference(const float *x, void *y, int k, void *st) { return quant_q4(FL_TYPE_Q4_1, true, x, y, k, st); }
static int vec_dot(int type, int n, float *s, const void *x, const void *y, void *st) {
    int rc = check_row(x, y, n);
    if (rc != FL_OK) return rc;
    if (!s) return set_error(FL_EINVAL, "null output");
    if ((n / FL_QK) % 2 != 0) return set_error(FL_EINVAL, "n/32 must be even (assert(nb %% 2 == 0), ggml.c:2372)");
    FL_HIP(vec_dot_aos(type, n, s, x, y, S(st)));
    return FL_OK;
}
int fl_vec_dot_q4_0_q8_0(int n, float *s, const void *x, const void *y, void *st) {
    return vec_dot(FL_TYPE_Q4_0, n, s, x, y, st);
}
int fl_vec_dot_q4_1_q8_0(int n, float *s, const void *x, const void *y, void *st) {
    return vec_dot(FL_TYPE_Q4_1, n, s, x, y, st);
}

/* table entries: reference signatures (void return, no stream) -> null stream + sync; a failure aborts
 * like GGML_ASSERT does (lib/ggml.c:138-144). */
static void tbl_check(int rc, const char *fn) {
    if (rc != FL_OK) {
        fprintf(stderr, "%s: %s\n", fn, fl_last_error());
        abort();
    }
    (void)hipStreamSynchronize(nullptr);
}
static void tbl_deq_q4_0(const void *x, float *y, int k) { tbl_check(fl_dequantize_row_q4_0(x, y, k, nullptr), __func__); }
static void tbl_deq_q4_1(const void *x, float *y, int k) { tbl_check(fl_dequantize_row_q4_1(x, y, k, nullptr), __func__); }
static void tbl_q4_0(const float *x, void *y, int k) { tbl_check(fl_quantize_row_q4_0(x, y, k, nullptr), __func__); }
static void tbl_q4_1(const float *x, void *y, int k) { tbl_check(fl_quantize_row_q4_1(x, y, k, nullptr), __func__); }
static void tbl_q4_0_ref(const float *x, void *y, int k) { tbl_check(fl_quantize_row_q4_0_reference(x, y, k, nullptr), __func__); }
static void tbl_q4_1_ref(const float *x, void *y, int k) { tbl_check(fl_quantize_row_q4_1_reference(x, y, k, nullptr), __func__); }
static void tbl_q8_0(const float *x, void *y, int k) { tbl_check(fl_quantize_row_q8_0(x, y, k, nullptr), __func__); }
static void tbl_dot_q4_0(const int n, float *s, const void *x, const void *y) {
    tbl_check(fl_vec_dot_q4_0_q8_0(n, s, x, y, nullptr), __func__);
}
static void tbl_dot_q4_1(const int n, float *s, const void *x, const void *y) {
    tbl_check(fl_vec_dot_q4_1_q8_0(n, s, x, y, nullptr), __func__);
}

fl_quantize_fns_t fl_get_quantize_fn(size_t type) {
    fl_quantize_fns_t t;
    memset(&t, 0, sizeof t);
    if (type == FL_TYPE_Q4_0) {
        t.dequantize_row_q = tbl_deq_q4_0;
        t.quantize_row_q = tbl_q4_0;
        t.quantize_row_q_reference = tbl_q4_0_ref;
        t.quantize_row_q_dot = tbl_q8_0;
        t.vec_dot_q = tbl_dot_q4_0;
    } else if (type == FL_TYPE_Q4_1) {
        t.dequantize_row_q = tbl_deq_q4_1;
        t.quantize_row_q = tbl_q4_1;
        t.quantize_row_q_reference = tbl_q4_1_ref;
        t.quantize_row_q_dot = tbl_q8_0;
        t.vec_dot_q = tbl_dot_q4_1;
    }
    return t;
}

/* ------------------------------------------------------------------ the op -------------------- */
fl_qact *fl_qact_create(int max_N, int K) {
    if (ensure_device() != FL_OK) return nullptr;
    if (max_N <= 0 || K <= 0 || K % FL_QK != 0) {
        set_error(FL_EINVAL, "bad qact shape N=%d K=%d", max_N, K);
        return nullptr;
    }
    fl_qact_impl *a = new (std::nothrow) fl_qact_impl();
    if (!a) return nullptr;
    memset(a, 0, sizeof *a);
    a->cap_N16 = fl_roundup(max_N, 16);
    a->K = K;
    a->KB = K / FL_QK;
    a->q_bytes = qact_bytes_q(max_N, K);
    a->s_bytes = qact_bytes_scale(max_N, K);
    hipError_t e = hipMalloc((void **)&a->q, a->q_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&a->d, a->s_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&a->s, a->s_bytes);
    a->h16_bytes = xh16_bytes(a->cap_N16, K);
    if (e == hipSuccess) e = hipMalloc((void **)&a->h16, a->h16_bytes);
    if (e == hipSuccess) e = hipMemset(a->h16, 0, a->h16_bytes);
    if (e != hipSuccess) {
        hip_fail(e, "fl_qact_create");
        fl_qact_free(a);
        return nullptr;
    }
    return a;
}

void fl_qact_free(fl_qact *a) {
    if (!a) return;
    if (a->q) (void)hipFree(a->q);
    if (a->d) (void)hipFree(a->d);
    if (a->s) (void)hipFree(a->s);
    if (a->h16) (void)hipFree(a->h16);
    delete static_cast<fl_qact_impl *>(a);
}

int fl_quantize_q8_layout(fl_qact *a_, const float *x, int ldx, int N, int K, int layout, void *st) {
    fl_qact_impl *a = static_cast<fl_qact_impl *>(a_);
    if (!a || !x) return set_error(FL_EINVAL, "null argument");
    if (K <= 0 || K % FL_QK != 0) return set_error(FL_EINVAL, "K=%d is not a positive multiple of 32", K);
    if (N <= 0 || (size_t)fl_roundup(N, 16) * (size_t)K > a->q_bytes)
        return set_error(FL_EINVAL, "N=%d K=%d exceeds the workspace (%zu bytes)", N, K, a->q_bytes);
    // the XH16 copy holds WHOLE 32-column tiles: a workspace reused with a longer K and fewer columns can pass the q_bytes check and
    // still be too small for it (created N = 64, K = 4096; reused N = 16, K = 11008)
    if (layout == 16 && xh16_bytes(N, K) > a->h16_bytes)
        return set_error(FL_EINVAL, "N=%d K=%d exceeds the workspace's XH16 copy (%zu bytes)", N, K, a->h16_bytes);
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return set_error(FL_EINVAL, "x rows must be 16-byte aligned");
    if (layout != 1 && layout != 16) return set_error(FL_EINVAL, "layout must be 1 or 16");
    a->N = N;
    a->N16 = fl_roundup(N, 16);
    a->KB = K / FL_QK;
    a->layout = layout;
    a->h16_valid = 0;
    if (layout == 1) FL_HIP(quantize_q8_qa1(x, ldx, N, K, *a, S(st)));
    else FL_HIP(quantize_q8_qa16(x, ldx, N, K, *a, S(st)));
    return FL_OK;
}

int fl_quantize_q8(fl_qact *a, const float *x, int ldx, int N, int K, void *st) {
    return fl_quantize_q8_layout(a, x, ldx, N, K, N <= 8 ? 1 : 16, st);
}

int fl_set_op_mode(int mode) {
    if (mode < -1 || mode > 1) return set_error(FL_EINVAL, "fl_set_op_mode: mode %d", mode);
    g_op_mode = mode;
    return FL_OK;
}

int fl_qact_export(const fl_qact *a_, void *blocks_dev, void *st) {
    const fl_qact_impl *a = static_cast<const fl_qact_impl *>(a_);
    if (!a || !blocks_dev || a->N <= 0) return set_error(FL_EINVAL, "empty workspace");
    const int K = a->KB * FL_QK;
    if (a->layout == 1) FL_HIP(export_qa1_to_aos(*a, a->N, K, blocks_dev, S(st)));
    else FL_HIP(export_qa16_to_aos(*a, a->N, K, blocks_dev, S(st)));
    return FL_OK;
}

}  // extern "C"
namespace fl {
int check_mm(const fl_qtensor *W, const fl_qact_impl *a, const float *y, int ldy) {
    if (!W || !a || !y) return set_error(FL_EINVAL, "null argument");
    if (a->N <= 0) return set_error(FL_EINVAL, "activation workspace is empty (call fl_quantize_q8 first)");
    if (a->KB != W->KB) return set_error(FL_EINVAL, "K mismatch: W has %d, activations %d", W->K, a->KB * FL_QK);
    if (ldy < W->M) return set_error(FL_EINVAL, "ldy=%d < M=%d", ldy, W->M);
    return FL_OK;
}
}  // namespace fl
extern "C" {

int fl_mul_mat_q(const fl_qtensor *W, const fl_qact *a_, float *y, int ldy, void *st) {
    const fl_qact_impl *a = static_cast<const fl_qact_impl *>(a_);
    int rc = check_mm(W, a, y, ldy);
    if (rc != FL_OK) return rc;
    if (op_exact()) return mul_mat_q_which(W, a_, y, ldy, 3, st);
    if (a->layout == 1) {
        FL_HIP(gemv_q4(*W, *a, a->N, y, ldy, S(st)));
    } else {
        if ((ldy & 3) || (reinterpret_cast<uintptr_t>(y) & 15)) return set_error(FL_EINVAL, "y rows must be 16-byte aligned");
        FL_HIP(gemm_q4_mfma(*W, *a, a->N, y, ldy, S(st)));
    }
    return FL_OK;
}

}  // extern "C"
namespace fl {
// which: 3 = the reference-order kernel of record for the workspace's layout and N; the other values pin one kernel family (tests, A/B):
// 0 naive, 1 fast MFMA GEMM, 2 fast GEMV, 4 reference-order VALU tiles, 5 reference-order H16 tiles, 6 round 3's reference-order tiles
int mul_mat_q_which(const fl_qtensor *W, const fl_qact *a_, float *y, int ldy, int which, void *st) {
    fl_qact_impl *a = const_cast<fl_qact_impl *>(static_cast<const fl_qact_impl *>(a_));
    int rc = check_mm(W, a, y, ldy);
    if (rc != FL_OK) return rc;
    if (which == 3 && a->layout == 16 && a->N >= 9) which = 5;     // the reference-order kernel of record for prefill sizes
    if (which == 5 || which == 6) {        // reference-order tile kernels: 5 = the H16 form (round 4), 6 = round 3's nibble form
        if (a->layout != 16) return set_error(FL_EINVAL, "gemm needs the QA16 layout");
        if (which == 6) {
            FL_HIP(gemm_q4_exact_mfma(*W, *a, a->N, y, ldy, S(st)));
            return FL_OK;
        }
        if ((rc = ensure_h16(W, a, st)) != FL_OK) return rc;
        FL_HIP(gemm_q4_exact_h16(*W, *a, a->N, y, ldy, S(st)));
    } else if (which == 4) {               // reference-order tile kernel in its VALU (v_dot4) form: cross-check of the MFMA form
        if (a->layout != 16) return set_error(FL_EINVAL, "gemm needs the QA16 layout");
        FL_HIP(gemm_q4_exact_valu(*W, *a, a->N, y, ldy, S(st)));
    } else if (which == 3) {               // reference-order kernels, whichever the layout says
        if (a->layout == 1 && a->N == 1 && !__atomic_load_n(&W->qwd, __ATOMIC_ACQUIRE)) {
            std::lock_guard<std::mutex> lk(g_lazy_mu);
            if (!W->qwd && (rc = fl_qtensor_build_qwd(const_cast<fl_qtensor *>(W), st)) != FL_OK) return rc;
        }
        if (a->layout == 1) FL_HIP(gemv_q4_exact(*W, *a, a->N, y, ldy, S(st)));
        else FL_HIP(gemm_q4_exact(*W, *a, a->N, y, ldy, S(st)));
    } else if (which == 2) {
        if (a->layout != 1) return set_error(FL_EINVAL, "gemv needs the QA1 layout");
        FL_HIP(gemv_q4(*W, *a, a->N, y, ldy, S(st)));
    } else {
        if (a->layout != 16) return set_error(FL_EINVAL, "gemm needs the QA16 layout");
        if (which == 1) FL_HIP(gemm_q4_mfma(*W, *a, a->N, y, ldy, S(st)));
        else FL_HIP(gemm_q4_naive(*W, *a, a->N, y, ldy, S(st)));
    }
    return FL_OK;
}
}  // namespace fl
extern "C" {

/* grow-on-demand workspace of the one-call op */
static fl_qact *g_ws = nullptr;
static size_t g_ws_elems = 0;

int fl_mul_mat_q_f32(const fl_qtensor *W, const float *x, int ldx, float *y, int ldy, int N, void *st) {
    int rc = ensure_device();
    if (rc != FL_OK) return rc;
    if (!W || !x || !y) return set_error(FL_EINVAL, "null argument");
    if (N <= 0) return set_error(FL_EINVAL, "N=%d", N);
    if (ldx < W->K) return set_error(FL_EINVAL, "ldx=%d < K=%d", ldx, W->K);
    const size_t need = (size_t)fl_roundup(N, 32) * W->K;      // (whole 32-column tiles: the XH16 copy of the reference-order GEMM)
    if (!g_ws || need > g_ws_elems || xh16_bytes(N, W->K) > static_cast<fl_qact_impl *>(g_ws)->h16_bytes) {
        if (g_ws) {
            (void)hipDeviceSynchronize();
            fl_qact_free(g_ws);
        }
        g_ws = fl_qact_create(fl_roundup(N, 32), W->K);
        g_ws_elems = g_ws ? need : 0;
        if (!g_ws) return FL_ENOMEM;
    }
    rc = fl_quantize_q8(g_ws, x, ldx, N, W->K, st);
    if (rc != FL_OK) return rc;
    return fl_mul_mat_q(W, g_ws, y, ldy, st);
}

}  // extern "C"

/* the one entry point of the test-hook library into this one (internal.h) */
extern "C" const fl::InternalTable *fl_internal_table(void) {
    static const fl::InternalTable t = {
#define X(name) &fl::name,
        FL_INTERNAL_FUNCS(X)
#undef X
        &fl::g_gemm_force_cfg, &fl::g_gemv_force_waves, &fl::g_op_mode, &fl::g_stream_min_groups, &fl::g_stream_force_nw, &fl::g_pv_waves, &fl::g_stream_helpers};
    return &t;
}
