"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1 --no-cpu-baseline` into
profiles/<round>_pmc_traffic.json: HBM-side bytes per launch for the kernels whose name pins their shape.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python3 tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w profiles/r5_pmc_traffic.json

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
-> x 1024; on gfx950 FETCH_SIZE counts wide (16 B/lane) coalesced reads at half their bytes -> x 2; WRITE_SIZE is exact for
16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

M, D = 512 * 316, 768
QKV = 2 * M * 3 * D          # bytes of the packed q|k|v activations (bf16)
KERNELS = {      # substring of the rocprofv3 kernel name -> (label, algorithmic bytes per launch)
    "gemm_nt_pp_kernel<6, 12, 2, 0, true>": (f"gemm_nt_pp_kernel<6, 12, 2, 0, true> M={M} N={4 * D} K={D}", 2 * (M * D + 4 * D * D) + 3 * M * 4 * D),
    "gemm_nt_pp_kernel<7, 12, 2, 0, true>": (f"gemm_nt_pp_kernel<7, 12, 2, 0, true> M={M} N={4 * D} K={D}", 2 * (M * D + 4 * D * D) + 3 * M * 4 * D),
    # fp16 stream rows, bf16 gradient stream: dy (2) + x (2) + dres (2) in, dx (2) out
    "ln_bwd_kernelILi3ELb0ELb1EDF16_": (f"ln_bwd_kernel<3, bf16 dy, bf16 dres, fp16 x> M={M} D={D}", 8 * M * D),
    # fp16 stream in and out: x (2) + add (2) in, x_out (2) + h (2) out
    "ln_fwd_kernelILi3EDF16_DF16_": (f"ln_fwd_kernel<3, fp16, fp16> M={M} D={D}", 8 * M * D),
    "mha_fwd_kernel<20": ("mha_fwd_kernel<20> b=512 S=316 H=12", QKV + 2 * M * D + 4 * 512 * 12 * 316),
    # q, k, v, dO, O in; dq, dk, dv out (+ lse in, delta out and back in)
    # the last block's one-query attention: K, V of every token in (+ probs out / in); backward: K (twice), V in, dK, dV out
    # round 4: the same attention against the LayerNorm output h1 itself: h1 in (+ probs out); backward: h1 in, dh1 out (+ probs in)
    "rows_ctx_fwd_kernel": ("rows_ctx_fwd_kernel b=512 S=316 H=12", 2 * M * D + 4 * 512 * 12 * 316),
    "rows_ctx_bwd_kernel": ("rows_ctx_bwd_kernel b=512 S=316 H=12", 2 * 2 * M * D + 4 * 512 * 12 * 316),
    "mha_bwd1s_kernel<20, false>": ("mha_bwd1s_kernel<20, false> b=512 S=316 H=12", QKV + 2 * 2 * M * D + QKV + 3 * 4 * 512 * 12 * 316),
    "mha_rows_fwd_kernel": ("mha_rows_fwd_kernel b=512 S=316 H=12", 2 * 2 * M * D + 4 * 512 * 12 * 316),
    "mha_rows_bwd_kernel": ("mha_rows_bwd_kernel b=512 S=316 H=12", 4 * 2 * M * D + 4 * 512 * 12 * 316),
}


def per_kernel(folder, counter):
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {folder}"
    vals = defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"]
        for sub in KERNELS:
            if sub in name:
                vals[sub].append(float(row["Counter_Value"]))
    # the frozen image tower (M = 25 600) launches the same kernels on a smaller problem: keep the audio-tower launches,
    # i.e. the cluster of the largest values (within 20 % of the maximum)
    out, cnt = {}, {}
    for sub, v in vals.items():
        big = [x for x in v if x >= 0.8 * max(v)]
        out[sub], cnt[sub] = sum(big) / len(big), len(big)
    return out, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, nf = per_kernel(fdir, "FETCH_SIZE")
    write, nw = per_kernel(wdir, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, `python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-full-last-block-check` (round 5, MI355X, the shipped build: tools/round_batch.sh pmc; tools/pmc_traffic.py holds the commands); counters are KiB -> x 1024; "
                     "FETCH_SIZE x 2 per MI355X_MICROARCH.md (wide coalesced reads report 1/2), WRITE_SIZE as is; mean per launch",
           "kernels": {}}
    for sub, (label, alg) in KERNELS.items():
        if sub in fetch and sub in write:
            res["kernels"][label] = {"fetch_bytes": int(fetch[sub] * 1024 * 2), "write_bytes": int(write[sub] * 1024),
                                     "algorithmic_bytes": int(alg), "launches": [nf[sub], nw[sub]]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
