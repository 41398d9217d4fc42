#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4final_gpu_tests.log
cp gpurun_out/parity_observed.jsonl gpurun_out/r4final_parity_observed.jsonl 2>/dev/null
python bench.py --steps 20 --warmup 3 > gpurun_out/r4final_bench.json 2> gpurun_out/r4final_bench.err
cat gpurun_out/r4final_gpu_tests.log; cat gpurun_out/r4final_bench.json
