#!/bin/bash
# BASELINE.json configs[4]'s tower on one GPU (audio ViT-L / 24 blocks, 1024 clips, recompute): e4m3 against bf16, alternating on one box.
# usage: bash tools/cfg5_ab.sh <tag> [rounds]
tag=${1:-r5}; rounds=${2:-2}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for mode in fp8 bf16; do
    flag=""; [ $mode = fp8 ] && flag="--fp8"
    timeout 600 python bench.py --script at --width 1024 --layers 24 --batch 1024 --recompute-mlp --steps 3 --warmup 2 --no-cpu-baseline $flag \
        > gpurun_out/${tag}_cfg5_${mode}_$r.json 2> gpurun_out/${tag}_cfg5_${mode}_$r.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_cfg5_${mode}_$r.json").read().strip().splitlines()[-1])
print("$mode round $r: %.1f ms/step, loss %.4f, peak %.1f GB" % (d["ms_per_step"], d["loss"], d["peak_mem_gb"]))
PY
  done
done
