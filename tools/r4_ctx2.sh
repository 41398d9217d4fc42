#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -k "last_block" 2>&1 | tail -5 | tee gpurun_out/r4f_model_tests.log
rm -rf gpurun_out/r4f_prof
VIPANT_TOWER_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_prof -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-last-block-check > gpurun_out/r4f_bench_serial.json 2> gpurun_out/r4f_bench_serial.err
python3 tools/kstats_shapes.py gpurun_out/r4f_prof 10 > gpurun_out/r4f_kernel_shapes_serial.txt 2>&1
find gpurun_out/r4f_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4f_kernel_stats_serial.csv \;
find gpurun_out/r4f_prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/r4f_trace.csv \;
rm -rf gpurun_out/r4f_prof
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4f_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the rows_ctx kernels and print 12 kernels around the forward one and 14 around the backward one (last occurrence)
names=[r['Kernel_Name'] for r in rows]
def show(i0,i1):
    for r in rows[i0:i1]:
        print(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>4s}  {r['Kernel_Name'][:110]}")
fi=[i for i,n in enumerate(names) if 'rows_ctx_fwd' in n and 'Li12' in n]
bi=[i for i,n in enumerate(names) if 'rows_ctx_bwd' in n and 'Li12' in n]
print(len(fi),len(bi))
i=fi[-2]; show(i-6,i+8); print('----'); i=bi[-2]; show(i-6,i+12)
PY
rm -f gpurun_out/r4f_trace.csv
