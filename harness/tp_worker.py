"""One rank of a tensor-parallel eval over RCCL (one process per GPU).  Used by tests/test_model_gpu.py (2 ranks, skipped on
a 1-GPU box) and runnable by hand on a multi-GPU node:

    for r in 0 1; do python -m harness.tp_worker $r 2 /tmp/tp_id.bin /tmp/tp_out_$r.npz & done; wait

With a fifth argument (a directory) every rank runs on device 0 and the communicator is the peer-mapped exchange alone
(fl_comm_create_p2p; handles through files in that directory): two processes, ONE GPU.

Rank 0 writes the 128-byte RCCL id to <idfile>; the others wait for it.  Every rank evaluates the same synthetic SMALL model
(harness/ggjt.py) on its own GPU: a prefill, a decode step through the captured hipGraph (RCCL collectives inside), and a
plain-launch decode step; rank 0 additionally evaluates the unsharded model.  Results go to <outfile>."""
import ctypes as C
import os
import sys
import time

import numpy as np


def p2p_comm(L, hip, rank, world, p2p_dir):
    """no RCCL (it refuses two ranks on one GPU): the peer-mapped exchange alone carries the collectives of a model whose messages are
    small enough for it; the hipIpc handles travel through files in p2p_dir"""
    comm = L.fl_comm_create_p2p(rank, world)
    if not comm:
        raise SystemExit("fl_comm_create_p2p: " + L.fl_last_error().decode())
    comm = C.c_void_p(comm)
    mine = (C.c_ubyte * 128)()
    hip.check(L.fl_comm_p2p_export(comm, mine), "p2p_export")
    with open(os.path.join(p2p_dir, f"h{rank}.tmp"), "wb") as f:
        f.write(bytes(mine))
    os.replace(os.path.join(p2p_dir, f"h{rank}.tmp"), os.path.join(p2p_dir, f"h{rank}.bin"))
    allh, t0 = b"", time.time()
    for r in range(world):
        pth = os.path.join(p2p_dir, f"h{r}.bin")
        while not os.path.exists(pth):
            if time.time() - t0 > 120:
                raise SystemExit("no handle from rank %d" % r)
            time.sleep(0.02)
        allh += open(pth, "rb").read()
    hip.check(L.fl_comm_p2p_import(comm, (C.c_ubyte * len(allh))(*allh)), "p2p_import")
    hip.check(L.fl_comm_p2p_selftest(comm), "p2p_selftest")          # (collective: what fl_comm_create does before it keeps the exchange)
    return comm


def main():
    rank, world, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from fastllama_amd import hip, ops
    from harness import ggjt
    from harness.flmodel import FlModel
    p2p_dir = sys.argv[5] if len(sys.argv) > 5 else None         # peer-exchange communicator, every rank on device 0 (see below)
    dev = 0 if p2p_dir else rank
    torch.cuda.set_device(dev)
    L = hip.load()
    hip.require_device(dev)
    raw = (C.c_ubyte * 128)()
    if p2p_dir:
        comm = p2p_comm(L, hip, rank, world, p2p_dir)
    elif rank == 0:
        hip.check(L.fl_comm_unique_id(raw), "fl_comm_unique_id")
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(raw))
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise SystemExit("no RCCL id from rank 0")
            time.sleep(0.05)
        raw = (C.c_ubyte * 128)(*open(idfile, "rb").read())
    if not p2p_dir:
        comm = L.fl_comm_create(raw, rank, world)
        if not comm:
            raise SystemExit("fl_comm_create: " + L.fl_last_error().decode())
        comm = C.c_void_p(comm)
    cfg, qtype = ggjt.SMALL, ggjt.Q4_0
    # the weights are quantized by the LIBRARY's quantize_row_q_reference -- bit for bit the rule of the reference's ggml_quantize_q4_x
    # (tests/test_kernels_gpu.py) and of the checker the parent test quantizes its copy with: a rank needs no checker library to start
    def quantize(qt, w):
        t = torch.from_numpy(np.ascontiguousarray(w)).cuda(dev)
        return ops.quantize_row_q(qt, t.view(-1), reference=True).view(w.shape[0], -1).cpu().numpy()
    tensors = ggjt.synth_tensors(cfg, qtype, quantize, seed=4321)
    toks = ggjt.text_tokens("The quick brown fox jumps over the lazy dog")
    m = FlModel(cfg, qtype, tensors, n_ctx=128, max_batch=64, tp_rank=rank, tp_size=world, device=dev)
    m.set_comm(comm)
    if os.environ.get("FL_TP_WORKER_MODE") == "peer_never_arrives":
        # both ranks decode two tokens; then rank 1 stops taking part: rank 0's next token must come back as an ERROR within the exchange's
        # bounded waits (FL_P2P_TIMEOUT_MS), not hang (tests/test_model_gpu.py)
        m.eval([toks[0]], n_past=0)
        m.eval([toks[1]], n_past=1)
        folded = L.fl_model_tp_folded(m.h)
        marker = os.path.join(p2p_dir, "rank0_done")
        if rank == 0:
            time.sleep(1.0)
            t0, err = time.time(), ""
            try:
                m.eval([toks[2]], n_past=2)
            except hip.FastLlamaHipError as e:
                err = str(e)
            np.savez(outfile, error=err, seconds=time.time() - t0, folded=folded)
            open(marker, "w").write("x")
        else:
            t0 = time.time()
            while not os.path.exists(marker) and time.time() - t0 < 120:
                time.sleep(0.05)
            np.savez(outfile, folded=folded)
        os._exit(0)                                              # (the communicator is out of step: no orderly teardown to test here)
    pre = m.eval(toks, all_logits=True)
    dec_graph = m.eval([toks[3]], n_past=len(toks))
    dec_graph2 = m.eval([toks[4]], n_past=len(toks) + 1)            # a replay of the captured graph
    hip.check(L.fl_model_set_graph(m.h, 0))
    dec_plain = m.eval([toks[4]], n_past=len(toks) + 1)
    out = dict(pre=pre, dec_graph=dec_graph, dec_graph2=dec_graph2, dec_plain=dec_plain)
    # a run of decode tokens: replays of the graph, then the two-launch attention of long contexts (its own graph), then plain launches again;
    # row-split models over the peer exchange run them with the exchanges folded into the producing launches (fl_model_tp_folded)
    def run(mm):
        seq, p = [], len(toks) + 2
        hip.check(L.fl_model_set_graph(mm.h, 1))
        for i in range(6):
            seq.append(mm.eval([toks[5 + i]], n_past=p)); p += 1
        nodes = L.fl_model_graph_nodes(mm.h)
        hip.check(L.fl_model_set_graph(mm.h, 1 | 16))          # every position takes the split attention
        for i in range(4):
            seq.append(mm.eval([toks[11 + i]], n_past=p)); p += 1
        nodes_split = L.fl_model_graph_nodes(mm.h)
        hip.check(L.fl_model_set_graph(mm.h, 16))
        for i in range(2):
            seq.append(mm.eval([toks[15 + i]], n_past=p)); p += 1
        hip.check(L.fl_model_set_graph(mm.h, 1))
        return np.concatenate(seq), nodes, nodes_split
    out["seq"], out["graph_nodes"], out["graph_nodes_split"] = run(m)
    out["folded"] = L.fl_model_tp_folded(m.h)
    out["n_layer"] = cfg["n_layer"]
    t0 = time.perf_counter()
    for i in range(32):
        m.eval_nocopy(np.asarray([toks[3]], np.int32), len(toks) + 14 + i)
    torch.cuda.synchronize()
    out["us_per_token"] = (time.perf_counter() - t0) / 32 * 1e6
    if rank == 0:
        full = FlModel(cfg, qtype, tensors, n_ctx=128, max_batch=64, device=0)
        out["full_pre"] = full.eval(toks, all_logits=True)
        out["full_dec"] = full.eval([toks[3]], n_past=len(toks))
        full.eval([toks[4]], n_past=len(toks) + 1)
        out["full_seq"], out["full_graph_nodes"], _ = run(full)
        full.free()
    np.savez(outfile, **out)
    m.free()
    L.fl_comm_destroy(comm)


if __name__ == "__main__":
    main()
