// comm.cpp -- thin RCCL wrapper behind the C-ABI (include/fastllama_hip.h, "collectives").
//
// The reference has no distributed code at all; this is new work for
// section 8(e): per layer the wo and w2 partial sums [N, n_embd] f32 are summed over the tensor-parallel
// ranks.  One process per GPU; the 128-byte ncclUniqueId is created on rank 0 and handed to the other
// ranks by the host program (bench.py / tests broadcast it with torch.distributed or a file).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>

#include "comm.h"
#include "runtime.h"

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

struct fl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    LocalGroup *lg = nullptr;
};

static_assert(sizeof(ncclUniqueId) == FL_COMM_ID_BYTES, "ncclUniqueId size");

extern "C" {

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
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error(FL_EHIP, "ncclCommInitRank: %s", ncclGetErrorString(r));
        delete c;
        return nullptr;
    }
    c->rank = rank;
    c->world = world;
    return c;
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
    ncclResult_t r = ncclAllGather(send_dev, recv_dev, count, ncclFloat32, c->comm, reinterpret_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return set_error(FL_EHIP, "ncclAllGather: %s", ncclGetErrorString(r));
    return FL_OK;
}

int fl_comm_is_local(const fl_comm *c) { return c && c->lg ? 1 : 0; }

/* test hook: one all-reduce captured into a hipGraph and replayed `replays` times (RCCL collectives inside the decode graph) */
int fl_comm_debug_graph_allreduce(fl_comm *c, float *buf_dev, size_t count, int replays, void *stream) {
    if (!c || c->lg || !buf_dev || !stream) return set_error(FL_EINVAL, "fl_comm_debug_graph_allreduce: needs an RCCL communicator and a stream");
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

void fl_comm_destroy(fl_comm *c) {
    if (!c) return;
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
