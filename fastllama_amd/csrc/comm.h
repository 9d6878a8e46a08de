// comm.h -- RCCL communicator for the tensor-parallel eval (one process per GPU, xGMI inside the node).
#pragma once
#include "../../include/fastllama_hip.h"
#include <hip/hip_runtime.h>

namespace fl {
// every bufs[r][i] <- bufs[0][i] + bufs[1][i] + ... (rank order) for i < count   (eval_kernels.hip)
hipError_t sum_buffers_inplace(float *const *bufs, int world, size_t count, hipStream_t st);
// one-shot exchange of a small message through peer-mapped buffers (eval_kernels.hip): all-reduce in rank order into `data`, or
// (gather_out != NULL) all-gather of `count` floats per rank into gather_out[rank][count]
struct P2PPeers {
    float *buf[FL_COMM_MAX_LOCAL];        // [2 slots][cap] floats of every rank (own + mapped peers')
    unsigned *flag[FL_COMM_MAX_LOCAL];    // [2] epoch flags of every rank
    unsigned *epoch;                      // this rank's epoch counter (device memory)
    size_t cap;
    int world, rank;
    unsigned long long timeout_ticks;     // bounded waits (100 MHz wall clock): p2p_timeout_ticks()
};
// how long an exchange waits for a peer before it gives up (counted: fl_comm_p2p_check fails the eval): 20 s, FL_P2P_TIMEOUT_MS overrides (tests)
unsigned long long p2p_timeout_ticks();
hipError_t p2p_exchange(const P2PPeers &peers, float *data, size_t count, float *gather_out, hipStream_t st);

// The fold regions of a communicator with the peer-mapped exchange behind it (tp_tail.h): per rank TP_FOLD_BYTES behind the two
// small-message slots, and in the flag page the words of the exchange kinds -- what a model needs to build its TpTail records.
constexpr size_t TP_FOLD_BYTES = 256 * 1024;
constexpr int TP_FOLD_KINDS = 8;                 // 0..3: the four exchanges of a decode layer; 7: the communicator's self-test
struct TpFold {
    int world, rank;
    unsigned char *region[FL_COMM_MAX_LOCAL];
    unsigned *flag[FL_COMM_MAX_LOCAL];           // rank r's words: [TP_FOLD_KINDS][FL_COMM_MAX_LOCAL source ranks]
    unsigned *ticket, *epoch;                    // local, [TP_FOLD_KINDS]
    unsigned *timeouts;                          // local (the counter fl_comm_p2p_check reads)
};
bool comm_fold(const fl_comm *c, TpFold *out);   // false: no peer-mapped exchange behind this communicator
}  // namespace fl
