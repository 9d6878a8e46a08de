"""One rank of the peer-mapped small-message exchange (fl_comm_create_p2p / p2p_export / p2p_import): the handles travel
through files, so two processes can run it on ONE GPU (where RCCL refuses to start) as well as on two.

    for r in 0 1; do python -m harness.p2p_worker $r 2 /tmp/p2p_dir [device] & done; wait
"""
import ctypes as C
import os
import sys
import time

import numpy as np


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    device = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from fastllama_amd import hip
    torch.cuda.set_device(device)
    L = hip.load()
    hip.require_device(device)
    comm = L.fl_comm_create_p2p(rank, world)
    if not comm:
        raise SystemExit("fl_comm_create_p2p: " + L.fl_last_error().decode())
    comm = C.c_void_p(comm)
    mine = (C.c_ubyte * 128)()
    hip.check(L.fl_comm_p2p_export(comm, mine), "p2p_export")
    with open(os.path.join(d, f"h{rank}.tmp"), "wb") as f:
        f.write(bytes(mine))
    os.replace(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}.bin"))
    allh = b""
    t0 = time.time()
    for r in range(world):
        p = os.path.join(d, f"h{r}.bin")
        while not os.path.exists(p):
            if time.time() - t0 > 120:
                raise SystemExit("no handle from rank %d" % r)
            time.sleep(0.02)
        allh += open(p, "rb").read()
    buf = (C.c_ubyte * len(allh))(*allh)
    hip.check(L.fl_comm_p2p_import(comm, buf), "p2p_import")
    assert L.fl_comm_has_p2p(comm) == 1
    st = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(st):
        # eager all-reduces of different sizes and contents (the slots alternate, epochs advance)
        for k, n in enumerate((4096, 1, 8192, 16384, 333, 4096)):
            x = torch.from_numpy(np.random.default_rng(100 * k + rank).standard_normal(n).astype(np.float32)).cuda()
            hip.check(L.fl_comm_allreduce_sum_f32(comm, x.data_ptr(), n, st.cuda_stream), "allreduce")
            st.synchronize()
            out[f"ar{k}"] = x.cpu().numpy()
        # all-gather
        x = torch.from_numpy(np.random.default_rng(7 + rank).standard_normal(1000).astype(np.float32)).cuda()
        g = torch.empty(world * 1000, device="cuda")
        hip.check(L.fl_comm_allgather_f32(comm, x.data_ptr(), 1000, g.data_ptr(), st.cuda_stream), "allgather")
        st.synchronize()
        out["ag"] = g.cpu().numpy()
        # one all-reduce captured in a hipGraph, replayed 3 times: x <- G^2 * sum after the third
        x = torch.from_numpy(np.random.default_rng(55 + rank).standard_normal(4096).astype(np.float32)).cuda()
        hip.check(L.fl_comm_debug_graph_allreduce(comm, x.data_ptr(), 4096, 3, st.cuda_stream), "graph allreduce")
        out["graph"] = x.cpu().numpy()
        # latency: 300 back-to-back all-reduces of a 7B decode message (4096 floats)
        x = torch.zeros(4096, device="cuda")
        for _ in range(20):
            hip.check(L.fl_comm_allreduce_sum_f32(comm, x.data_ptr(), 4096, st.cuda_stream))
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            hip.check(L.fl_comm_allreduce_sum_f32(comm, x.data_ptr(), 4096, st.cuda_stream))
        st.synchronize()
        out["us_per_allreduce"] = np.array((time.perf_counter() - t0) / 300 * 1e6)
    np.savez(os.path.join(d, f"out{rank}.npz"), **out)
    L.fl_comm_destroy(comm)


if __name__ == "__main__":
    main()
