// comm.cpp -- thin RCCL wrapper behind the C-ABI (include/fastllama_hip.h, "collectives").
//
// The reference has no distributed code at all; this is new work for
// section 8(e): per layer the wo and w2 partial sums [N, n_embd] f32 are summed over the tensor-parallel
// ranks.  One process per GPU; the 128-byte ncclUniqueId is created on rank 0 and handed to the other
// ranks by the host program (bench.py / tests broadcast it with torch.distributed or a file).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <new>

#include "comm.h"
#include "runtime.h"

using namespace fl;

struct fl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
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

int fl_comm_allreduce_sum_f32(fl_comm *c, float *buf_dev, size_t count, void *stream) {
    if (!c || !buf_dev) return set_error(FL_EINVAL, "fl_comm_allreduce: null argument");
    ncclResult_t r = ncclAllReduce(buf_dev, buf_dev, count, ncclFloat32, ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return set_error(FL_EHIP, "ncclAllReduce: %s", ncclGetErrorString(r));
    return FL_OK;
}

int fl_comm_rank(const fl_comm *c) { return c ? c->rank : -1; }
int fl_comm_size(const fl_comm *c) { return c ? c->world : 0; }

void fl_comm_destroy(fl_comm *c) {
    if (!c) return;
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
}

}  // extern "C"
