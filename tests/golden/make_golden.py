"""Generate tests/golden/q4_golden.npz by running the REFERENCE ITSELF (oracle/_ref, compiled from
/root/reference by oracle/Makefile).  Run in the build container:  python tests/golden/make_golden.py

The fixture travels to the GPU box (which has no /root/reference) and pins both the C restatement
(oracle/q4_oracle.c) and the HIP path to outputs of the real reference functions:
  quantize_row_q8_0, dequantize_row_q4_{0,1}, ggml_vec_dot_q4_{0,1}_q8_0 (via ggml_internal_get_quantize_fn)
  ggml_quantize_q4_{0,1} and a full ggml_graph_compute of ggml_mul_mat (via oracle/ref_driver.c).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def edge_rows(rng, n, K):
    x = rng.normal(0, 1, (n, K)).astype(np.float32)
    x[1, :32] = 0.0                     # an all-zero block: amax = 0 -> id = 0
    x[2, 5] = 3.0e30                    # one huge element dominates its block
    x[3] *= 1e-20                       # tiny values
    x[4, 32:64] = -np.abs(x[4, 32:64])  # all-negative block
    x[5, 64:96] = np.float32(0.5)       # constant block: x*id lands exactly on 127
    x[6, 96:128] = np.arange(32, dtype=np.float32) - 15.5  # ties at .5 after scaling? exercise rounding
    return x


def main():
    R = oracle.Ref()
    rng = np.random.default_rng(20260925)
    out = {}
    for tag, (M, K, N) in {"s": (48, 256, 8), "m": (64, 4096, 3)}.items():
        w = rng.normal(0, 0.02, (M, K)).astype(np.float32)
        w[0, :32] = 0.0                 # zero block in the weights (d = 0)
        x = edge_rows(rng, 8, K) if tag == "s" else rng.normal(0, 1, (N, K)).astype(np.float32)
        if tag == "s":
            out[f"{tag}_w"] = w             # the medium case keeps only the quantized weights (size)
        out[f"{tag}_x"] = x
        out[f"{tag}_q8"] = np.stack([R.quantize_row_q8_0(r) for r in x])
        for qt, nm in ((oracle.Q4_0, "q40"), (oracle.Q4_1, "q41")):
            wq = R.quantize_q4(qt, w)
            out[f"{tag}_{nm}"] = wq
            if tag == "s":
                out[f"{tag}_{nm}_deq"] = np.stack([R.dequantize_row(qt, r, K) for r in wq])
            out[f"{tag}_{nm}_y"] = R.mul_mat_q(qt, wq, x, n_threads=3)
            vd = np.empty((x.shape[0], M), dtype=np.float32)
            for n in range(x.shape[0]):
                for m in range(M):
                    vd[n, m] = R.vec_dot(qt, K, wq[m], out[f"{tag}_q8"][n])
            out[f"{tag}_{nm}_vd"] = vd
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "q4_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
