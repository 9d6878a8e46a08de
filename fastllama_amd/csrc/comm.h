// comm.h -- RCCL communicator for the tensor-parallel eval (one process per GPU, xGMI inside the node).
#pragma once
#include "../../include/fastllama_hip.h"
#include <hip/hip_runtime.h>

namespace fl {
// every bufs[r][i] <- bufs[0][i] + bufs[1][i] + ... (rank order) for i < count   (eval_kernels.hip)
hipError_t sum_buffers_inplace(float *const *bufs, int world, size_t count, hipStream_t st);
}  // namespace fl
