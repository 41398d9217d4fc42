#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "rows_ctx or head_expand" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -k "last_block or golden or trajectory" 2>&1 | tail -5 | tee gpurun_out/r4f_model_tests.log
for v in 1 0; do
rm -rf gpurun_out/r4f_prof
VIPANT_LAST_BLOCK_CTX=$v VIPANT_TOWER_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_prof -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-last-block-check > gpurun_out/r4f_bench_serial_$v.json 2> gpurun_out/r4f_bench_serial.err
python3 tools/kstats_shapes.py gpurun_out/r4f_prof 10 > gpurun_out/r4f_kernel_shapes_serial_$v.txt 2>&1
rm -rf gpurun_out/r4f_prof
done
for v in 1 0 1 0; do
  VIPANT_LAST_BLOCK_CTX=$v python bench.py --steps 20 --warmup 3 2> gpurun_out/r4f_bench_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ctx=$v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4f_ab.txt
done
