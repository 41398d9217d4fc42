#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "lars or trainer or optimizer" 2>&1 | tail -4
rm -rf gpurun_out/r4k_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4k_prof -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-last-block-check > gpurun_out/r4k_bench.json 2> gpurun_out/r4k_bench.err
python3 tools/kstats_shapes.py gpurun_out/r4k_prof 10 2>&1 | grep -E "lars|cast_multi|sum of"
rm -rf gpurun_out/r4k_prof
