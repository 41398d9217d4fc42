#!/bin/bash
# BASELINE.json configs[4]'s tower on one GPU (audio ViT-L / 24 blocks, 1024 clips): e4m3 against bf16, alternating on one box.
# usage: bash tools/cfg5_ab.sh <tag> [rounds] [plan]     plan = recompute | keep (default: no recomputation, 235 GB) | both
tag=${1:-r5}; rounds=${2:-2}; plan=${3:-keep}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
run() {   # name, flags...
  local name=$1; shift
  timeout 600 python bench.py --script at --width 1024 --layers 24 --batch 1024 --steps 3 --warmup 2 --no-cpu-baseline "$@" \
      > gpurun_out/${tag}_cfg5_${name}.json 2> gpurun_out/${tag}_cfg5_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_cfg5_${name}.json").read().strip().splitlines()[-1])
    print("${name}: %.1f ms/step, loss %.4f, peak %.1f GB" % (d["ms_per_step"], d["loss"], d["peak_mem_gb"]))
except Exception as e:
    print("${name}: failed", e)
PY
}
for r in $(seq 1 $rounds); do
  if [ $plan = keep ] || [ $plan = both ]; then
    run fp8_keep_$r --fp8
    VIPANT_ATTN_EMIT=0 run fp8_keep_noemit_$r --fp8
    run bf16_keep_$r
  fi
  if [ $plan = recompute ] || [ $plan = both ]; then
    run fp8_$r --fp8 --recompute-mlp
    run bf16_$r --recompute-mlp
  fi
done
