#!/bin/bash
# configs[4] tower (audio ViT-L / 24 blocks, 1024 clips, e4m3): rocprofv3 kernel trace of 3 steps (1 warm-up + 2 timed) -> <tag>_cfg5_fp8_kernel_stats.csv
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r6}
export TMPDIR=/tmp
rm -rf /tmp/prof5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o run -- python3 bench.py --script at --width 1024 --layers 24 --batch 1024 --fp8 \
    --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_cfg5_prof.json 2> gpurun_out/${tag}_cfg5_prof.err
f=$(find /tmp/prof5 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${tag}_cfg5_fp8_kernel_stats.csv
head -32 gpurun_out/${tag}_cfg5_fp8_kernel_stats.csv | cut -c1-170
tail -c 300 gpurun_out/${tag}_cfg5_prof.json
