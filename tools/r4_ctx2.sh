#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/r4f_prof
VIPANT_TOWER_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4f_prof -o s -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-full-last-block-check > gpurun_out/r4f_bench_serial.json 2> gpurun_out/r4f_bench_serial.err
find gpurun_out/r4f_prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/r4f_trace.csv \;
rm -rf gpurun_out/r4f_prof
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4f_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
def show(i0,i1):
    t0=int(rows[i0]['Start_Timestamp'])
    for r in rows[i0:i1]:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r['Grid_Size_X']:>8s}  {r['Kernel_Name'][:100]}")
fi=[i for i,n in enumerate(names) if 'rows_ctx_fwd' in n and '12' in n]
bi=[i for i,n in enumerate(names) if 'rows_ctx_bwd' in n and '12' in n]
print(len(fi),len(bi))
# audio forward is the one with the big grid neighbours: pick the occurrence whose duration is largest
fa=max(fi[-4:], key=lambda i:int(rows[i]['End_Timestamp'])-int(rows[i]['Start_Timestamp']))
i=bi[-1]
fa=max(j for j in fi if j < i)
show(fa-8,i+18)
PY
rm -f gpurun_out/r4f_trace.csv
