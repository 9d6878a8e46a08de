"""GPU: bench.py's contract -- one JSON line with the required keys -- at N = 1 and through the world > 1 code path
(two ranks launched like the driver does; on a 1-GPU box both replicas sit on device 0 and the timing reduction runs over
gloo, see the FL_BENCH_* overrides in bench.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_rank_tiny():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--n-batch", "64", "--steps", "2",
                        "--warmup", "1", "--decode-steps", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "workload" in d["config"]


def test_bench_two_ranks_code_path():
    env = dict(os.environ, FL_BENCH_DEVICE="0", FL_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model",
                        "tiny", "--n-batch", "64", "--steps", "2", "--warmup", "1", "--decode-steps", "4"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _line(r.stdout)                                      # rank 0 only prints
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "cpu_baseline" not in d                          # reported at N = 1 only


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the driver's form): bench.py spawns torch.distributed.run itself
    and rank 0's line says n_gpus = 2.  On this 1-GPU box both ranks sit on device 0 and reduce their timings over gloo
    (FL_BENCH_* overrides), so the leg is the replica one: tp_ok is False and no RCCL communicator is counted."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--n-batch", "64", "--steps",
                        "2", "--warmup", "1", "--decode-steps", "4"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=_clean_env(FL_BENCH_DEVICE="0", FL_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["tp_ok"] is False and d["n_ranks_rccl"] == 0


def test_bench_self_launched_tensor_parallel_leg():
    """The same bare command with the tensor-parallel leg rehearsed as two processes on ONE GPU (peer exchange instead of RCCL):
    the headline is the TP eval (tp_ok, strong scaling) and the line says that no RCCL communicator carried it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--n-batch", "32", "--steps",
                        "2", "--warmup", "1", "--decode-steps", "4"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=_clean_env(FL_BENCH_DEVICE="0", FL_BENCH_BACKEND="gloo", FL_BENCH_P2P_ONLY="1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["tp_ok"] is True and d["scaling"] == "strong" and d["n_ranks_rccl"] == 0 and "tp_error" not in d


def test_bench_tensor_parallel_leg_as_eight_processes_on_one_gpu():
    """`python bench.py --gpus 8` the way the driver will run it on an 8-GPU node, rehearsed as EIGHT processes on the one GPU there is (peer
    exchange instead of RCCL, a model that splits eight ways by rows): rank bookkeeping, the handle exchange, the self-test, the row split and
    the folded decode all run with world = 8, so that the first real node does not trip on a rank-count bug."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--model", "tiny8", "--n-batch", "32", "--steps",
                        "2", "--warmup", "1", "--decode-steps", "4", "--no-fast", "--tp-timeout", "600"], capture_output=True, text=True, timeout=1500,
                       cwd=ROOT, env=_clean_env(FL_BENCH_DEVICE="0", FL_BENCH_BACKEND="gloo", FL_BENCH_P2P_ONLY="1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _line(r.stdout)
    assert d["n_gpus"] == 8 and d["tp_ok"] is True and d["scaling"] == "strong" and d["n_ranks_rccl"] == 0 and "tp_error" not in d
    assert d["tp_decode"]["exchanges_folded_into_producers"] is True


def test_bench_reports_both_modes_and_config3():
    """N = 1: the headline is the default (reference-order) mode, the fast mode's timings sit beside it; tp_ok / n_ranks_rccl are null."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--n-batch", "64", "--steps", "2",
                        "--warmup", "1", "--decode-steps", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["mode"] == "exact" and d["fast_mode"]["prefill_tokens_per_s"] > 0 and d["fast_mode"]["decode_tokens_per_s"] > 0
    assert "exact_h16" in d["roofline"]["kernel"] and "exact" in d["roofline_decode"]["kernel"] and "traffic_source" in d["roofline"]
    assert d["fast_mode"]["roofline"]["frac"] > 0
    assert d["tp_ok"] is None and d["n_ranks_rccl"] is None
