#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nt or colsum" 2>&1 | tail -4
bash tools/r4_ctx2.sh 2>&1 | grep -E "skinny|colsum" | cut -c1-140
for v in 0 2097152 0 2097152; do
  VIPANT_GEMM_VARIANT=$v python bench.py --steps 20 --warmup 3 2> gpurun_out/r4i_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('variant=$v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4i_ab.txt
done
