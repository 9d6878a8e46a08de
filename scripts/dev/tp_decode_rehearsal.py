"""Row-split tensor-parallel DECODE rehearsed with G processes on ONE GPU (all this project can reach): the peer-mapped exchange as the
communicator (RCCL refuses several ranks per device), a full-size model sharded G ways, a run of decode tokens through the replayed hipGraph.
Reports, per rank: are the exchanges the tails of the producing launches (fl_model_tp_folded), kernel nodes per token and layer, ms per token --
the ranks share one GPU's bandwidth and CUs, so the time is NOT a scaling number; what it shows is the folded sequence against the collective
one (FL_TP_FOLD=0) on the same box, and that the logits equal the unsharded model's bit for bit (rank 0 checks, models up to 13B).

  python scripts/dev/tp_decode_rehearsal.py <model> <G> <dir> [steps]      (starts the G ranks itself)"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def worker(name, rank, world, d, steps):
    import torch
    from fastllama_amd import hip
    from harness import synth
    from harness.flmodel import FlModel
    from harness.tp_worker import p2p_comm
    torch.cuda.set_device(0)
    L = hip.load()
    hip.require_device(0)
    comm = p2p_comm(L, hip, rank, world, d)
    qtype = int(os.environ.get("FL_QTYPE", "2"))
    cfg = dict(synth.MODELS[name])
    if os.environ.get("FL_LAYERS"):
        cfg["n_layer"] = int(os.environ["FL_LAYERS"])
    n_ctx = int(os.environ.get("FL_NCTX", "256"))
    m = FlModel(cfg, qtype, synth.synth_model_tensors(cfg, qtype), n_ctx=n_ctx, max_batch=8, tp_rank=rank, tp_size=world, device=0)
    m.set_comm(comm)
    toks = np.random.default_rng(0).integers(3, 259, 256).astype(np.int32)
    p = 0
    for i in range(3):                                   # a short history, token by token (prefill messages do not fit the exchange)
        m.eval([toks[p]], n_past=p); p += 1
    seq = [m.eval([toks[p + i]], n_past=p + i) for i in range(4)]
    p += 4
    nodes, folded = L.fl_model_graph_nodes(m.h), L.fl_model_tp_folded(m.h)
    t1 = toks[:1].copy()
    for i in range(4):
        m.eval_nocopy(t1, p + i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            m.eval_nocopy(t1, p + 4 + i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    nl = cfg["n_layer"]
    line = f"rank {rank}/{world} {name} Q4_{qtype - 2} x {nl} layers: folded={folded} graph nodes {nodes} = {(nodes - 4) / nl:.2f} per layer + 4; {best * 1e3:.3f} ms per token ({1 / best:.1f} tok/s, {world} ranks on ONE GPU)"
    m.free()
    if rank == 0 and os.environ.get("FL_CHECK", "1") == "1":
        full = FlModel(cfg, qtype, synth.synth_model_tensors(cfg, qtype), n_ctx=n_ctx, max_batch=8, device=0)
        q = 0
        for i in range(3):
            full.eval([toks[q]], n_past=q); q += 1
        want = [full.eval([toks[q + i]], n_past=q + i) for i in range(4)]
        bad = sum(int((a.view(np.uint32) != b.view(np.uint32)).sum()) for a, b in zip(seq, want))
        line += f"; logits of 4 decode tokens vs the unsharded model: {bad} values differ"
        t0 = time.perf_counter()
        for i in range(steps):
            full.eval_nocopy(t1, q + 8 + i)
        torch.cuda.synchronize()
        line += f"; unsharded, same box (the other rank idle): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per token"
        full.free()
    print(line, flush=True)
    L.fl_comm_destroy(comm)


if __name__ == "__main__":
    if len(sys.argv) > 5 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]))
    else:
        name, G, d = sys.argv[1], int(sys.argv[2]), sys.argv[3]
        steps = sys.argv[4] if len(sys.argv) > 4 else "64"
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            if f.startswith("h") and f.endswith(".bin"):
                os.remove(os.path.join(d, f))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", name, str(r), str(G), d, steps], cwd=ROOT, env=env) for r in range(G)]
        rc = 0
        for pr in procs:
            try:
                rc |= pr.wait(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise SystemExit("rehearsal workers did not finish")
        raise SystemExit(rc)
