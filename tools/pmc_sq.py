"""Summarise one rocprofv3 SQ-counter pass of `bench.py --steps 1 --warmup 1 --no-cpu-baseline` into profiles/<round>_pmc_sq.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
        SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_sq -o s -- python3 bench.py \
        --steps 1 --warmup 1 --no-cpu-baseline
    python3 tools/pmc_sq.py /tmp/pmc_sq profiles/r2_pmc_sq.json

Per kernel (launches of the largest problem only: grid and duration within 20 % of the maximum): mfma_busy_frac =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) (GRBM_GUI_ACTIVE sums the 8 XCDs); wait / active fractions are of
SQ_WAVE_CYCLES (quad-cycles, all three); durations from the kernel trace of the same run."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KEEP = ("gemm_nt_pp_kernel", "gemm_tn_pp_kernel", "gemm_tn8_pp_kernel", "mha_fwd_kernel<20", "mha_bwd1s_kernel", "mha_bwd_dq_kernel<20", "mha_bwd_dkv_kernel<20", "ln_fwd_kernel",
        "ln_bwd_kernel", "nce_tile_kernel")


def main():
    folder, out = sys.argv[1:3]
    cfile = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)[0]
    tfile = glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    if tfile:
        for r in csv.DictReader(open(tfile[0])):
            dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch -> counter -> value
    grid = {}
    for r in csv.DictReader(open(cfile)):
        name = r["Kernel_Name"]
        if not any(k in name for k in KEEP):
            continue
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name) or re.search(r"\d+(\w+_kernelI[\w]*?E)Ev", name)
        short = m.group(1) if m else name[:60]
        per[short][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        grid[r["Dispatch_Id"]] = int(r.get("Grid_Size", 0) or 0)
    res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                     "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline (MI355X; tools/pmc_sq.py); per-launch averages over the largest-problem launches",
           "kernels": []}
    for short, disp in sorted(per.items()):
        ids = list(disp)
        gmax = max(grid[i] for i in ids)
        ids = [i for i in ids if grid[i] >= 0.8 * gmax]
        if dur:
            dmax = max(dur.get(i, 0.0) for i in ids)
            ids = [i for i in ids if dur.get(i, 0.0) >= 0.8 * dmax] or ids
        def avg(c):
            v = [disp[i].get(c) for i in ids if c in disp[i]]
            return sum(v) / len(v) if v else 0.0
        wave, gui = avg("SQ_WAVE_CYCLES"), avg("GRBM_GUI_ACTIVE")
        us = sum(dur.get(i, 0.0) for i in ids) / len(ids) if dur else None
        res["kernels"].append({
            "kernel": short, "grid_threads": gmax, "launches": len(ids), "avg_us": round(us, 1) if us else None,
            "mfma_busy_frac": round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024), 3) if gui else None,
            "wait_any_frac": round(avg("SQ_WAIT_ANY") / wave, 2) if wave else None,
            "wait_inst_any_frac": round(avg("SQ_WAIT_INST_ANY") / wave, 2) if wave else None,
            "active_inst_frac": round(avg("SQ_ACTIVE_INST_ANY") / wave, 2) if wave else None,
            "wait_inst_lds_frac": round(avg("SQ_WAIT_INST_LDS") / wave, 3) if wave else None,
            "lds_bank_conflict_cycles_per_cu": round(avg("SQ_LDS_BANK_CONFLICT") / 256),
            "gui_active_per_us": round(gui / 8 / us / 1e3, 2) if us else None})
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k in res["kernels"]:
        print(k)


if __name__ == "__main__":
    main()
