// tp_tail.h -- the tensor-parallel exchange as the TAIL of the kernel that produced the data (round 5).
//
// Row-split tensor parallelism (the reference's own split across threads, lib/ggml.c:8127-8135: rank = thread with its own HBM) moves four
// small vectors per layer and decode token between the ranks: the Q8_0 attention output, the wo rows, the silu features, the w2 rows.  As
// collectives they were pack -> all-gather -> unpack (-> add): 9 kernels + 4 collectives per layer.  Here every rank owns a FOLD REGION
// (comm.cpp; peer-mapped like the small-message exchange buffers) with the SAME layout on every rank -- [x row][x2 row][silu features]
// [Q8_0 planes of the attention output] -- and the decode kernels of a tensor-parallel model read and write their operands THERE:
//   * a producer writes its slice (its rows / features / blocks) into its own region, in place, AND into every peer's region at the same
//     offset (system-scope stores: tp_put in the GEMV epilogues, tp_push of a workgroup's own Q8_0 blocks in the attention);
//   * every workgroup takes a ticket; the LAST one of the launch publishes this rank's epoch in every peer's flag word and waits until every
//     peer's epoch has arrived here (bounded spin).  (A producer without tp_put / tp_push: the last workgroup copies the slice -- the ranges
//     of the record -- to the peers first; that is also what the tail does as a launch of its own, tp_tail_kernel.)
//   * the launch ends => every slice of the vector is in this rank's region, and the next launch -- an ordinary kernel -- reads it.
// One workgroup per rank waits, so ranks that share a GPU (the two-process rehearsal on the one-GPU test box) cannot starve each other.
// No pack / unpack / add kernels and no collective launch: the layer is its five decode launches (model.cpp, run_eval_kernels).
//
// What keeps a slot from being overwritten while it is still read: a vector's only full-width reader is the launch that produces the NEXT
// exchange (wo reads the attention planes and produces x2; w1|w3 reads x2 and produces the silu features; w2 reads those and produces x;
// wq|wk|wv and the lm-head read x, and the attention -- the next producer -- follows them on the stream), and a peer can pass exchange k + 1
// only after this rank published it, i.e. after all of this rank's workgroups of that reader have finished.  Its write into the same slot
// comes three exchanges later.
//
// Visibility inside the launch: a peer's flag must not overtake the data.  Every store to a peer is a system-scope atomic store (written
// through), each workgroup waits for its own (s_waitcnt) before its ticket, the tickets are agent-scope atomics, and the flag stores are
// system-scope RELEASE stores by the workgroup that saw every ticket.  Stores into the own region that the LAST workgroup has to read back
// (no tp_put: the ranges) must have reached memory before the ticket too -- other workgroups run on other XCDs, whose L2s are not coherent
// with each other: agent-scope atomic stores, or plain stores and one agent-scope release per workgroup (FENCE = true; a release writes back
// the XCD's whole L2 -- per workgroup of a GEMV that was 58 us per launch, profiles/r04_decode_exact.md -- so it is for launches of a few
// dozen workgroups only; unused since the attention pushes its own blocks).
// Never run between GPUs: fl_comm_create keeps the exchange only if its collective self-test passes.
#pragma once
#include "../../include/fastllama_hip.h"
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fl {

constexpr int TP_TAIL_RANGES = 3;
struct TpTail {
    int world, rank;
    int n_ranges;
    unsigned off[TP_TAIL_RANGES], bytes[TP_TAIL_RANGES];      // this rank's slice: byte ranges of the fold region (multiples of 4; the same offsets on every rank)
    unsigned char *region[FL_COMM_MAX_LOCAL];                 // fold region of every rank (own + mapped peers')
    unsigned *flag[FL_COMM_MAX_LOCAL];                        // rank r's flag words of THIS exchange, one per source rank
    unsigned *ticket;                                         // local: workgroups of the launch that have finished (0 between launches)
    unsigned *epoch;                                          // local: exchanges of this kind completed
    unsigned *timeouts;                                       // local: waits given up (fl_comm_p2p_check)
    unsigned long long timeout_ticks;                         // of the 100 MHz wall clock
};

// the tail a launcher may fuse into the kernel it is about to launch: set by the model before the call, cleared by the launcher that took it
// (a launcher that cannot -- round 3's fallback kernels -- leaves it, and the model launches tp_tail_kernel behind the producer)
extern thread_local const TpTail *tp_pending_tail;
inline const TpTail *tp_take_tail() { const TpTail *t = tp_pending_tail; tp_pending_tail = nullptr; return t; }
hipError_t tp_tail_launch(const TpTail *tt_dev, hipStream_t st);       // eval_kernels.hip: the tail as a launch of its own (one workgroup)
// one epoch of the communicator's self-test (eval_kernels.hip): a multi-workgroup producer with fused tails, then a consumer launch reading with plain loads
hipError_t tp_selftest_epoch(const TpTail *exchange_dev, const TpTail *barrier_dev, unsigned area_off, unsigned slice_words, unsigned epoch,
                             unsigned *errors_dev, unsigned *sink_dev, hipStream_t st);

#ifdef __HIPCC__
// A producer's store of one value of its slice, with a tail: into this rank's region (written through: the workgroup that publishes runs on another
// XCD) and straight into every peer's (the same offset), so that the last workgroup has nothing left to copy.  p points into this rank's region.
__device__ __forceinline__ void tp_put(const TpTail *__restrict__ tt, float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int world = tt->world, rank = tt->rank;
    const size_t off = reinterpret_cast<unsigned char *>(p) - tt->region[rank];
    for (int r = 0; r < world; ++r)
        if (r != rank) __hip_atomic_store(reinterpret_cast<float *>(tt->region[r] + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A producer workgroup's own piece of the slice (plain stores, then a barrier) -> every peer's region: all threads of the workgroup
__device__ __forceinline__ void tp_push(const TpTail *__restrict__ tt, const void *own, unsigned bytes) {
    const int world = tt->world, rank = tt->rank;
    const size_t off = reinterpret_cast<const unsigned char *>(own) - tt->region[rank];
    for (unsigned w = threadIdx.x; w < (bytes >> 2); w += blockDim.x) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(own)[w];
        for (int r = 0; r < world; ++r)
            if (r != rank) __hip_atomic_store(reinterpret_cast<uint32_t *>(tt->region[r] + off) + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Called by EVERY workgroup of the launch, by all of its threads, after its last global store.  STANDALONE: the launch is the tail itself
// (one workgroup behind a producer that could not carry it: the kernel boundary made the producer's stores visible).
// PUSHED: the workgroups sent their values to the peers themselves (tp_put), the last one only publishes.
template <bool FENCE, bool STANDALONE = false, bool PUSHED = false>
__device__ __forceinline__ void tp_tail(const TpTail *__restrict__ tt) {
    __shared__ unsigned tp_last_s;
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    if constexpr (!STANDALONE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
        __syncthreads();
        if (tid == 0) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned total = gridDim.x * gridDim.y * gridDim.z;
            tp_last_s = __hip_atomic_fetch_add(tt->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
        }
        __syncthreads();
        if (!tp_last_s) return;
    }
    const int world = tt->world, rank = tt->rank;
    const unsigned e = __hip_atomic_load(tt->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    // an earlier exchange over this communicator already gave up on a peer: the ranks are out of step and every result since is void -- waiting the
    // full bound again at each of the token's remaining exchanges would only delay the error the host is about to get (fl_comm_p2p_check)
    bool gave_up = __hip_atomic_load(tt->timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (int i = 0; i < (PUSHED ? 0 : tt->n_ranges); ++i) {
        const unsigned words = tt->bytes[i] >> 2;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tt->region[rank] + tt->off[i]);
        for (unsigned w = tid; w < words; w += nt) {
            const uint32_t v = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int r = 0; r < world; ++r)
                if (r != rank)
                    __hip_atomic_store(reinterpret_cast<uint32_t *>(tt->region[r] + tt->off[i]) + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < (unsigned)world && (int)tid != rank) {
        __hip_atomic_store(tt->flag[tid] + rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned *f = tt->flag[rank] + tid;
        // bounded: a peer that died must not hang this GPU's queue for ever (the host sees the count and fails the eval: fl_comm_p2p_check)
        const unsigned long long t0 = wall_clock64();
        // (relaxed polls and ONE acquire behind them: an acquire load invalidates the caches every time round the loop)
        while (!gave_up && (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {      // (epochs only grow)
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > tt->timeout_ticks) gave_up = true;
        }
        if (gave_up) atomicAdd(tt->timeouts, 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(tt->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tt->epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#endif

}  // namespace fl
