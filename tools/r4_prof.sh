#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/r4d_prof
VIPANT_TOWER_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4d_prof -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-last-block-check > gpurun_out/r4d_bench_serial.json 2> gpurun_out/r4d_bench_serial.err
python3 tools/kstats_shapes.py gpurun_out/r4d_prof 10 > gpurun_out/r4d_kernel_shapes_serial.txt 2>&1
python3 tools/kstats.py gpurun_out/r4d_prof 10 40 > gpurun_out/r4d_kstats.txt 2>&1
find gpurun_out/r4d_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4d_kernel_stats_serial.csv \;
rm -rf gpurun_out/r4d_prof
cat gpurun_out/r4d_bench_serial.json | head -c 600; echo; cat gpurun_out/r4d_kernel_shapes_serial.txt | head -80
