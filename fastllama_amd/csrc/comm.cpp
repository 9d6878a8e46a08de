// comm.cpp -- thin RCCL wrapper behind the C-ABI (include/fastllama_hip.h, "collectives").
//
// The reference has no distributed code at all; this is new work for
// section 8(e): per layer the wo and w2 partial sums [N, n_embd] f32 are summed over the tensor-parallel
// ranks.  One process per GPU; the 128-byte ncclUniqueId is created on rank 0 and handed to the other
// ranks by the host program (bench.py / tests broadcast it with torch.distributed or a file).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "comm.h"
#include "runtime.h"
#include "tp_tail.h"

using namespace fl;

// Shards of ONE process on ONE device (fl_comm_create_local): the collective is a rendezvous of the shards' host
// threads plus one device kernel that sums the partials in rank order.  It exists so that the tensor-parallel model
// path can be run -- and tested -- with G logical shards on a single GPU, and for hosts that
// drive several shards from one process.
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 1, arrived = 0, refs = 0;
    unsigned long gen = 0;
    float *bufs[FL_COMM_MAX_LOCAL] = {nullptr};
    float *recvs[FL_COMM_MAX_LOCAL] = {nullptr};   // all-gather destinations
    size_t count = 0;
    int status = FL_OK;        // of the round being collected
    int done_status = FL_OK;   // of the round that just completed
};

// One-shot exchange of small messages through peer-mapped buffers (hipIpc; p2p_exchange_kernel, eval_kernels.hip): what a decode
// token's 16-32 KB all-reduces and its logits all-gather take instead of a ring collective.  Set up next to the RCCL
// communicator when every step of the handle exchange succeeds; otherwise the communicator simply stays RCCL-only.
constexpr size_t P2P_CAP = 64 * 1024;          // floats per slot (256 KB): the exchange buffer of a rank is two slots + the fold region (tp_tail.h)
// flag page (4096 B, exported): words [0..1] the slots' flags, [16] this rank's epoch, [17] timeouts; the fold exchanges: [64 + kind * 8 + source rank]
// the flags peers write, [128 + kind] tickets, [144 + kind] epochs (local)
constexpr int FOLD_FLAG0 = 64, FOLD_TICKET0 = 128, FOLD_EPOCH0 = 144;
static_assert(FL_COMM_MAX_LOCAL <= 8 && TP_FOLD_KINDS <= 8, "flag page layout");
constexpr size_t P2P_MAX_COUNT = 16 * 1024;    // messages up to 64 KB go this way (one workgroup moves them)
// (FL_P2P_MAX_COUNT: up to a whole slot -- rehearsals of larger models on a communicator without RCCL behind it, scripts/dev/run_r5_o.sh)
static size_t p2p_max_count() {
    static const size_t v = [] { const char *e = getenv("FL_P2P_MAX_COUNT"); const long n = e ? atol(e) : 0; return n > 0 ? (size_t)n : P2P_MAX_COUNT; }();
    return v < P2P_CAP ? v : P2P_CAP;
}
struct P2PState {
    P2PPeers peers{};
    void *own_buf = nullptr, *own_flag = nullptr;
    void *mapped[2 * FL_COMM_MAX_LOCAL] = {nullptr};
    int n_mapped = 0;
    bool ready = false;
    unsigned timeouts_seen = 0;       // of peers.epoch[1] (the exchange kernel's give-up counter): fl_comm_p2p_check
};

struct fl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    LocalGroup *lg = nullptr;
    P2PState p2p;
};

unsigned long long fl::p2p_timeout_ticks() {
    static const unsigned long long v = [] {
        const char *e = getenv("FL_P2P_TIMEOUT_MS");
        const long ms = e ? atol(e) : 0;
        return (unsigned long long)(ms > 0 ? ms : 20000) * 100000ull;
    }();
    return v;
}

static int p2p_alloc(fl_comm *c) {
    P2PState &p = c->p2p;
    if (p.own_buf) return FL_OK;
    if (c->world < 2 || c->world > FL_COMM_MAX_LOCAL) return set_error(FL_EINVAL, "peer exchange needs 2..%d ranks", FL_COMM_MAX_LOCAL);
    // fine-grained device memory (what RCCL uses for its own peer buffers): coherent for peers while a kernel runs; the kernel
    // uses system-scope accesses on top of it.  A runtime that refuses the flag gets NO exchange (round 6): coarse-grained memory would
    // leave the decode launches' plain loads of peer-written slices to the self-test's luck -- the communicator stays on RCCL, and says so.
    auto alloc = [](void **ptr, size_t bytes) {
        hipError_t r = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained);
        if (r != hipSuccess) { (void)hipGetLastError(); warn("peer exchange: hipDeviceMallocFinegrained refused (%s): no peer-mapped exchange, collectives go through RCCL", hipGetErrorString(r)); }
        return r;
    };
    hipError_t e = alloc(&p.own_buf, 2 * P2P_CAP * sizeof(float) + TP_FOLD_BYTES);
    if (e == hipSuccess) e = alloc(&p.own_flag, 4096);               // [0..1] flags (exported), [16] this rank's epoch
    if (e == hipSuccess) e = hipMemset(p.own_flag, 0, 4096);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (p.own_buf) (void)hipFree(p.own_buf);
        if (p.own_flag) (void)hipFree(p.own_flag);
        p.own_buf = p.own_flag = nullptr;
        return set_error(FL_EHIP, "peer exchange buffers: %s", hipGetErrorString(e));
    }
    p.peers.world = c->world;
    p.peers.rank = c->rank;
    p.peers.cap = P2P_CAP;
    p.peers.timeout_ticks = p2p_timeout_ticks();
    p.peers.buf[c->rank] = static_cast<float *>(p.own_buf);
    p.peers.flag[c->rank] = static_cast<unsigned *>(p.own_flag);
    p.peers.epoch = static_cast<unsigned *>(p.own_flag) + 16;
    return FL_OK;
}
static void p2p_free(fl_comm *c) {
    P2PState &p = c->p2p;
    for (int i = 0; i < p.n_mapped; ++i) (void)hipIpcCloseMemHandle(p.mapped[i]);
    if (p.own_buf) (void)hipFree(p.own_buf);
    if (p.own_flag) (void)hipFree(p.own_flag);
    p = P2PState{};
}
static_assert(2 * sizeof(hipIpcMemHandle_t) == FL_COMM_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size");
static bool p2p_selftest(fl_comm *c);

static_assert(sizeof(ncclUniqueId) == FL_COMM_ID_BYTES, "ncclUniqueId size");

extern "C" {

/* The check fl_comm_create runs before it keeps the exchange, for hosts that moved the handles themselves (fl_comm_create_p2p): COLLECTIVE --
 * every rank calls it after fl_comm_p2p_import.  Three epochs of the decode pattern over the same slots (p2p_selftest below); the first epoch waits up
 * to 20 s for a peer, the next two 2 s.  On failure the exchange is FREED: the communicator can no longer fold a model's exchanges. */
int fl_comm_p2p_selftest(fl_comm *c) {
    if (!c || !c->p2p.ready) return set_error(FL_EINVAL, "fl_comm_p2p_selftest: no peer-mapped exchange behind this communicator");
    if (p2p_selftest(c)) return FL_OK;
    const int rank = c->rank;
    p2p_free(c);                 // a communicator known to be bad must not fold a model's exchanges: its waits would be skipped from now on (tp_tail.h)
    return set_error(FL_EHIP, "peer exchange self-test failed on rank %d (a slice did not arrive intact, or a peer did not in time): the exchange is gone", rank);
}

int fl_comm_unique_id(void *out) {
    if (!out) return set_error(FL_EINVAL, "null id buffer");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return set_error(FL_EHIP, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(out, &id, sizeof id);
    return FL_OK;
}

fl_comm *fl_comm_create(const void *id_bytes, int rank, int world) {
    if (ensure_device() != FL_OK) return nullptr;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) {
        set_error(FL_EINVAL, "fl_comm_create: bad arguments");
        return nullptr;
    }
    fl_comm *c = new (std::nothrow) fl_comm();
    if (!c) return nullptr;
    // peer-mapped exchange buffers for the small messages: handles travel through the communicator itself
    // ON by default since round 5 (FL_P2P=0: RCCL only).  The exchange has run between processes on ONE GPU only -- no multi-GPU node was
    // available to this project -- so a communicator keeps it only if (a) every step of the handle exchange worked on every rank and (b)
    // the self-test below moved patterned slices between all ranks within its bounded waits; anything else leaves the communicator on RCCL alone.
    const char *want_p2p_env = getenv("FL_P2P");
    const bool want_p2p = world >= 2 && world <= FL_COMM_MAX_LOCAL && !(want_p2p_env && want_p2p_env[0] == '0');
    // The handshake's staging buffer is allocated BEFORE the communicator exists: once ncclCommInitRank has returned, every rank
    // must execute both collectives below whatever happens locally (a local ncclCommAbort does not reliably unblock peers already
    // inside ncclAllGather), so nothing that can fail may sit between the init and them.  Device memory first, pinned host memory
    // (which RCCL reads and writes in place) if the device has none left; with neither, the rank gives up before joining anything.
    constexpr size_t STAGE_BYTES = FL_COMM_P2P_HANDLE_BYTES * (FL_COMM_MAX_LOCAL + 1);
    void *stage = nullptr;
    bool stage_host = false;
    if (want_p2p && hipMalloc(&stage, STAGE_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        stage = nullptr;
        if (hipHostMalloc(&stage, STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            set_error(FL_ENOMEM, "fl_comm_create: no memory for the peer-exchange handshake (nothing joined yet)");
            delete c;
            return nullptr;
        }
        stage_host = true;
    }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error(FL_EHIP, "ncclCommInitRank: %s", ncclGetErrorString(r));
        if (stage) (void)(stage_host ? hipHostFree(stage) : hipFree(stage));
        delete c;
        return nullptr;
    }
    c->rank = rank;
    c->world = world;
    if (want_p2p) {
        unsigned char mine[FL_COMM_P2P_HANDLE_BYTES], all[FL_COMM_P2P_HANDLE_BYTES * FL_COMM_MAX_LOCAL];
        // Everything from here on is collective-safe: a rank whose local setup failed still joins the all-gather with a zero handle
        // and the agreement with a 0.
        bool ok = fl_comm_p2p_export(c, mine) == FL_OK;
        if (ok) ok = hipMemcpy(static_cast<unsigned char *>(stage) + sizeof all, mine, sizeof mine, hipMemcpyHostToDevice) == hipSuccess;
        // (every rank takes part in the collective whatever happened locally: a rank that failed contributes a zero handle)
        if (!ok) (void)hipMemset(static_cast<unsigned char *>(stage) + sizeof all, 0, sizeof mine);
        {
            ncclResult_t g = ncclAllGather(static_cast<unsigned char *>(stage) + sizeof all, stage, sizeof mine, ncclUint8, c->comm, nullptr);
            if (g != ncclSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(all, stage, sizeof mine * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess)
                ok = false;
            bool every = ok;
            for (int r = 0; r < world && every; ++r) {
                bool nz = false;
                for (size_t i = 0; i < sizeof mine; ++i) nz = nz || all[(size_t)r * sizeof mine + i] != 0;
                every = nz;
            }
            if (!every || fl_comm_p2p_import(c, all) != FL_OK) p2p_free(c);
        }
        // every rank must take the same path for a given message: the exchange is used only if it came up on ALL ranks
        // (the agreement word lives in the staging buffer: no allocation that could fail between the two collectives)
        int *agree = static_cast<int *>(stage);
        auto all_agree = [&](bool mine) {
            int mine_ok = mine ? 1 : 0, all_ok = 0;
            const bool sent = hipMemcpy(agree, &mine_ok, sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
            if (!sent) (void)hipMemset(agree, 0, sizeof(int));
            return ncclAllReduce(agree, agree, 1, ncclInt32, ncclMin, c->comm, nullptr) == ncclSuccess && hipDeviceSynchronize() == hipSuccess &&
                   hipMemcpy(&all_ok, agree, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && all_ok == 1;
        };
        // ... and only if it then WORKS between these devices (every rank has the mappings: the self-test is collective), again on all ranks
        if (all_agree(c->p2p.ready)) {
            const bool works = getenv("FL_P2P_NO_SELFTEST") ? true : p2p_selftest(c);
            if (!all_agree(works)) p2p_free(c);
        } else {
            p2p_free(c);
        }
        (void)(stage_host ? hipHostFree(stage) : hipFree(stage));
        (void)hipGetLastError();
    }
    return c;
}

/* A communicator that has ONLY the peer-mapped exchange (no RCCL): for hosts that move the handles themselves
 * (fl_comm_p2p_export / fl_comm_p2p_import) -- and for testing the exchange with two processes on one GPU, where RCCL
 * refuses to start.  Messages beyond the exchange buffers' reach fail. */
fl_comm *fl_comm_create_p2p(int rank, int world) {
    if (ensure_device() != FL_OK) return nullptr;
    if (world < 2 || world > FL_COMM_MAX_LOCAL || rank < 0 || rank >= world) {
        set_error(FL_EINVAL, "fl_comm_create_p2p: bad arguments");
        return nullptr;
    }
    fl_comm *c = new (std::nothrow) fl_comm();
    if (!c) return nullptr;
    c->rank = rank;
    c->world = world;
    return c;
}

int fl_comm_p2p_export(fl_comm *c, void *handles_out) {
    if (!c || !handles_out || c->lg) return set_error(FL_EINVAL, "fl_comm_p2p_export: bad arguments");
    int rc = p2p_alloc(c);
    if (rc != FL_OK) return rc;
    hipIpcMemHandle_t h[2];
    hipError_t e = hipIpcGetMemHandle(&h[0], c->p2p.own_buf);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], c->p2p.own_flag);
    if (e != hipSuccess) return set_error(FL_EHIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handles_out, h, sizeof h);
    return FL_OK;
}

int fl_comm_p2p_import(fl_comm *c, const void *handles_all) {
    if (!c || !handles_all || c->lg || !c->p2p.own_buf) return set_error(FL_EINVAL, "fl_comm_p2p_import: export first");
    if (c->p2p.ready || c->p2p.n_mapped) return set_error(FL_EINVAL, "fl_comm_p2p_import: already imported");
    P2PState &p = c->p2p;
    const unsigned char *src = static_cast<const unsigned char *>(handles_all);
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h[2];
        memcpy(h, src + (size_t)r * sizeof h, sizeof h);
        for (int k = 0; k < 2; ++k) {
            void *ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, h[k], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return set_error(FL_EHIP, "hipIpcOpenMemHandle (rank %d): %s", r, hipGetErrorString(e));
            p.mapped[p.n_mapped++] = ptr;
            if (k == 0) p.peers.buf[r] = static_cast<float *>(ptr);
            else p.peers.flag[r] = static_cast<unsigned *>(ptr);
        }
    }
    p.ready = true;
    return FL_OK;
}

int fl_comm_has_p2p(const fl_comm *c) { return c && c->p2p.ready ? 1 : 0; }
}  // extern "C"

bool fl::comm_fold(const fl_comm *c, TpFold *out) {
    if (!c || !c->p2p.ready || !out) return false;
    const P2PPeers &pp = c->p2p.peers;
    *out = TpFold{};
    out->world = pp.world;
    out->rank = pp.rank;
    for (int r = 0; r < pp.world; ++r) {
        out->region[r] = reinterpret_cast<unsigned char *>(pp.buf[r] + 2 * pp.cap);
        out->flag[r] = pp.flag[r] + FOLD_FLAG0;
    }
    out->ticket = pp.flag[pp.rank] + FOLD_TICKET0;
    out->epoch = pp.flag[pp.rank] + FOLD_EPOCH0;
    out->timeouts = pp.epoch + 1;
    return true;
}

// One exchange kind's record for a slice [off, off + bytes) of the fold region
static TpTail fold_tail(const TpFold &f, int kind, unsigned off, unsigned bytes, unsigned long long timeout_ticks) {
    TpTail t{};
    t.world = f.world; t.rank = f.rank;
    t.n_ranges = 1; t.off[0] = off; t.bytes[0] = bytes;
    for (int r = 0; r < f.world; ++r) { t.region[r] = f.region[r]; t.flag[r] = f.flag[r] + kind * FL_COMM_MAX_LOCAL; }
    t.ticket = f.ticket + kind; t.epoch = f.epoch + kind; t.timeouts = f.timeouts;
    t.timeout_ticks = timeout_ticks;
    return t;
}

// Does the exchange work between THESE devices, the way the decode path uses it?  Three epochs over the SAME slots, each a multi-workgroup producer
// launch with fused tails (tp_put from many workgroups, the ticket, the publish / wait) followed by a consumer launch that reads every rank's slice with
// plain loads -- after the producer launch of the next epoch has first pulled the old lines into its caches (tp_selftest_epoch, eval_kernels.hip).
// Collective: every rank of a communicator whose handles were imported calls it.  A failure leaves the communicator on RCCL alone (fl_comm_create) or
// without the exchange (fl_comm_p2p_selftest).  What it cannot show on the one-GPU test box: that system-scope stores cross xGMI the way they cross
// between processes on one device -- it is the check that runs on the real node before any model folds its exchanges.
static bool p2p_selftest(fl_comm *c) {
    TpFold f;
    if (!comm_fold(c, &f)) return false;
    constexpr unsigned SLW = 1024;                                   // words per rank's slice (4 KB: several cache lines per workgroup)
    static_assert((size_t)FL_COMM_MAX_LOCAL * SLW * 4 <= TP_FOLD_BYTES, "the self-test's area");
    struct Dev { TpTail t[2]; unsigned errors, sink[8]; } *dv = nullptr;
    if (hipMalloc((void **)&dv, sizeof(Dev)) != hipSuccess) { (void)hipGetLastError(); return false; }
    // (the first epoch waits up to 20 s: a peer's first launch from this library may still be loading its code object; then 2 s per wait)
    Dev h{};
    h.t[0] = fold_tail(f, TP_FOLD_KINDS - 1, (unsigned)f.rank * SLW * 4, SLW * 4, 2000000000ull);      // the exchange
    h.t[1] = fold_tail(f, TP_FOLD_KINDS - 2, 0, 0, 2000000000ull);                                       // the barrier behind the consumer
    h.t[1].n_ranges = 0;
    bool ok = true;
    unsigned before = 0, after = 0;
    ok = ok && hipMemcpy(&before, f.timeouts, 4, hipMemcpyDeviceToHost) == hipSuccess;
    // (the area is NOT cleared first: a peer that is ahead may already be writing its first slice into it; whatever the slots hold, only this epoch's
    //  pattern passes)
    for (unsigned epoch = 1; epoch <= 3 && ok; ++epoch) {
        if (epoch == 2) h.t[0].timeout_ticks = h.t[1].timeout_ticks = 200000000ull;
        ok = hipMemcpy(dv, &h, sizeof h, hipMemcpyHostToDevice) == hipSuccess &&
             tp_selftest_epoch(&dv->t[0], &dv->t[1], 0, SLW, epoch, &dv->errors, dv->sink, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
             hipMemcpy(&h.errors, &dv->errors, 4, hipMemcpyDeviceToHost) == hipSuccess && h.errors == 0;
    }
    if (hipMemcpy(&after, f.timeouts, 4, hipMemcpyDeviceToHost) != hipSuccess || after != before) ok = false;
    if (after != before) c->p2p.timeouts_seen = after;          // (the self-test's give-ups are not an eval's)
    (void)hipFree(dv);
    (void)hipGetLastError();
    return ok;
}

extern "C" {
/* exchanges of this rank that gave up waiting for a peer (p2p_exchange_kernel's bounded spin); synchronises the device */
int fl_comm_p2p_timeouts(const fl_comm *c) {
    if (!c || !c->p2p.ready) return 0;
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&n, c->p2p.peers.epoch + 1, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}

/* After a synchronised tensor-parallel eval: did the exchange kernel of this rank give up waiting for a peer since the last check?
 * (Its bounded spin then went on with whatever the slots held -- the results of that eval are invalid, and the epochs of the ranks
 * may be out of step.)  The caller has synchronised the stream the exchanges ran on.  FL_OK, or FL_EHIP with the count. */
int fl_comm_p2p_check(fl_comm *c) {
    if (!c || !c->p2p.ready) return FL_OK;
    unsigned n = 0;
    if (hipMemcpy(&n, c->p2p.peers.epoch + 1, sizeof n, hipMemcpyDeviceToHost) != hipSuccess)
        return set_error(FL_EHIP, "peer exchange: cannot read the timeout counter");
    if (n == c->p2p.timeouts_seen) return FL_OK;
    const unsigned fresh = n - c->p2p.timeouts_seen;
    c->p2p.timeouts_seen = n;
    return set_error(FL_EHIP, "peer exchange: %u exchange(s) of rank %d gave up waiting for a peer -- the tensor-parallel results of this eval "
                              "are invalid and the communicator should be recreated", fresh, c->rank);
}

int fl_comm_create_local(int world, fl_comm **out) {
    if (ensure_device() != FL_OK) return FL_ENODEV;
    if (!out || world < 1 || world > FL_COMM_MAX_LOCAL) return set_error(FL_EINVAL, "fl_comm_create_local: world must be 1..%d", FL_COMM_MAX_LOCAL);
    LocalGroup *g = new (std::nothrow) LocalGroup();
    if (!g) return set_error(FL_ENOMEM, "out of host memory");
    g->world = world;
    g->refs = world;
    for (int r = 0; r < world; ++r) {
        fl_comm *c = new (std::nothrow) fl_comm();
        if (!c) {
            for (int q = 0; q < r; ++q) { delete out[q]; out[q] = nullptr; }
            delete g;
            return set_error(FL_ENOMEM, "out of host memory");
        }
        c->rank = r;
        c->world = world;
        c->lg = g;
        out[r] = c;
    }
    return FL_OK;
}

static int local_allreduce(fl_comm *c, float *buf, size_t count, hipStream_t st) {
    LocalGroup *g = c->lg;
    hipError_t e = hipStreamSynchronize(st);                 // this shard's partial sum is complete
    if (e != hipSuccess) return set_error(FL_EHIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
    std::unique_lock<std::mutex> lk(g->mu);
    const unsigned long my_gen = g->gen;
    g->bufs[c->rank] = buf;
    if (g->arrived == 0) { g->count = count; g->status = FL_OK; }     // a new round starts clean: an earlier failure is not sticky
    else if (g->count != count) g->status = FL_EINVAL;
    if (++g->arrived == g->world) {
        if (g->status == FL_OK) {
            e = sum_buffers_inplace(g->bufs, g->world, count, st);   // every buffer <- sum over ranks, rank order
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) g->status = FL_EHIP;
        }
        g->done_status = g->status;      // the verdict of THIS round, read by every member after the wake-up
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->gen != my_gen; });
    }
    const int rc = g->done_status;
    return rc == FL_OK ? FL_OK : set_error(rc, "local all-reduce failed (mismatched counts or a device error)");
}

int fl_comm_allreduce_sum_f32(fl_comm *c, float *buf_dev, size_t count, void *stream) {
    if (!c || !buf_dev) return set_error(FL_EINVAL, "fl_comm_allreduce: null argument");
    if (c->lg) return local_allreduce(c, buf_dev, count, reinterpret_cast<hipStream_t>(stream));
    if (c->p2p.ready && count <= p2p_max_count()) {
        hipError_t e = p2p_exchange(c->p2p.peers, buf_dev, count, nullptr, reinterpret_cast<hipStream_t>(stream));
        return e == hipSuccess ? FL_OK : set_error(FL_EHIP, "peer all-reduce: %s", hipGetErrorString(e));
    }
    if (!c->comm) return set_error(FL_EINVAL, "all-reduce of %zu floats needs an RCCL communicator", count);
    ncclResult_t r = ncclAllReduce(buf_dev, buf_dev, count, ncclFloat32, ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return set_error(FL_EHIP, "ncclAllReduce: %s", ncclGetErrorString(r));
    return FL_OK;
}

// recv[r * count + i] <- rank r's send[i]: the logits slices of the row-split lm-head
static int local_allgather(fl_comm *c, const float *send, size_t count, float *recv, hipStream_t st) {
    LocalGroup *g = c->lg;
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_error(FL_EHIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
    std::unique_lock<std::mutex> lk(g->mu);
    const unsigned long my_gen = g->gen;
    g->bufs[c->rank] = const_cast<float *>(send);
    g->recvs[c->rank] = recv;
    if (g->arrived == 0) { g->count = count; g->status = FL_OK; }
    else if (g->count != count) g->status = FL_EINVAL;
    if (++g->arrived == g->world) {
        if (g->status == FL_OK) {
            for (int d = 0; d < g->world && e == hipSuccess; ++d)
                for (int r = 0; r < g->world && e == hipSuccess; ++r)
                    e = hipMemcpyAsync(g->recvs[d] + (size_t)r * count, g->bufs[r], count * 4, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) g->status = FL_EHIP;
        }
        g->done_status = g->status;
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->gen != my_gen; });
    }
    const int rc = g->done_status;
    return rc == FL_OK ? FL_OK : set_error(rc, "local all-gather failed (mismatched counts or a device error)");
}

int fl_comm_allgather_f32(fl_comm *c, const float *send_dev, size_t count, float *recv_dev, void *stream) {
    if (!c || !send_dev || !recv_dev) return set_error(FL_EINVAL, "fl_comm_allgather: null argument");
    if (c->lg) return local_allgather(c, send_dev, count, recv_dev, reinterpret_cast<hipStream_t>(stream));
    if (c->p2p.ready && count <= p2p_max_count()) {
        hipError_t e = p2p_exchange(c->p2p.peers, const_cast<float *>(send_dev), count, recv_dev, reinterpret_cast<hipStream_t>(stream));
        return e == hipSuccess ? FL_OK : set_error(FL_EHIP, "peer all-gather: %s", hipGetErrorString(e));
    }
    if (!c->comm) return set_error(FL_EINVAL, "all-gather of %zu floats needs an RCCL communicator", count);
    ncclResult_t r = ncclAllGather(send_dev, recv_dev, count, ncclFloat32, c->comm, reinterpret_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return set_error(FL_EHIP, "ncclAllGather: %s", ncclGetErrorString(r));
    return FL_OK;
}

int fl_comm_is_local(const fl_comm *c) { return c && c->lg ? 1 : 0; }

/* test hook: one all-reduce captured into a hipGraph and replayed `replays` times (RCCL collectives inside the decode graph) */
int fl_comm_debug_graph_allreduce(fl_comm *c, float *buf_dev, size_t count, int replays, void *stream) {
    if (!c || c->lg || !buf_dev || !stream) return set_error(FL_EINVAL, "fl_comm_debug_graph_allreduce: needs an RCCL or peer-exchange communicator and a stream");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return set_error(FL_EHIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    const int rc = fl_comm_allreduce_sum_f32(c, buf_dev, count, stream);
    e = hipStreamEndCapture(st, &g);
    if (rc != FL_OK || e != hipSuccess) { if (g) (void)hipGraphDestroy(g); return set_error(FL_EHIP, "capture of ncclAllReduce failed"); }
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    for (int i = 0; i < replays && e == hipSuccess; ++i) e = hipGraphLaunch(ex, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (ex) (void)hipGraphExecDestroy(ex);
    return e == hipSuccess ? FL_OK : set_error(FL_EHIP, "graph replay of ncclAllReduce: %s", hipGetErrorString(e));
}

int fl_comm_rank(const fl_comm *c) { return c ? c->rank : -1; }
int fl_comm_size(const fl_comm *c) { return c ? c->world : 0; }
/* ranks the RCCL communicator itself reports (ncclCommCount); 0 for a communicator without RCCL behind it (local shards, p2p only) */
int fl_comm_rccl_ranks(const fl_comm *c) {
    if (!c || !c->comm) return 0;
    int n = 0;
    return ncclCommCount(c->comm, &n) == ncclSuccess ? n : -1;
}

void fl_comm_destroy(fl_comm *c) {
    if (!c) return;
    p2p_free(c);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->lg) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(c->lg->mu);
            last = --c->lg->refs == 0;
        }
        if (last) delete c->lg;
    }
    delete c;
}

}  // extern "C"
