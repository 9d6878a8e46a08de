// comm.h -- RCCL communicator for the tensor-parallel eval (one process per GPU, xGMI inside the node).
#pragma once
#include "../../include/fastllama_hip.h"
