#!/bin/bash
# e4m3 weight gradients (round 6): parity of the e4m3 towers with VIPANT_FP8_TN on / off, and the configs[4] bench both ways on one box
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r6}
rm -f gpurun_out/parity_observed.jsonl
python -m pytest tests/test_fp8_gpu.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_replicas_gpu.py -m gpu -x -q -k "e4m3" 2>&1 | tail -3
cp gpurun_out/parity_observed.jsonl gpurun_out/${tag}_parity_fp8tn_on.jsonl; rm -f gpurun_out/parity_observed.jsonl
VIPANT_FP8_TN=0 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "end_to_end_golden_e4m3" 2>&1 | tail -2
cp gpurun_out/parity_observed.jsonl gpurun_out/${tag}_parity_fp8tn_off.jsonl
for r in 1 2; do
  for tn in 1 0; do
    VIPANT_FP8_TN=$tn python bench.py --script at --width 1024 --layers 24 --batch 1024 --fp8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FP8_TN=$tn round $r', d['ms_per_step'], 'ms/step', d['peak_mem_gb'], 'GB loss', d['loss'])"
  done
done | tee gpurun_out/${tag}_cfg5_fp8tn_ab.txt
