#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/r4_ctx2.sh 2>&1 | grep -E "rows_ctx|ln_fwd_kernelILi3EDF16_DF16|ln_bwd_kernelILi3ELb0ELb0EDF16" | cut -c1-140
for v in 1 0 1 0 1 0; do
  VIPANT_LAST_BLOCK_CTX=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-last-block-check 2> gpurun_out/r4f_bench_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ctx=$v', d['ms_per_step'], d['value'], d['step_mfma_frac'])" | tee -a gpurun_out/r4f_ab.txt
done
