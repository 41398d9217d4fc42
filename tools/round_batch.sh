#!/bin/bash
# One gpurun call = one batch: `gpurun --timeout S -- 'bash tools/round_batch.sh <batch> [tag]'`.  Everything a batch writes goes
# to gpurun_out/<tag>_* (scratch; what is to be judged is copied into profiles/ afterwards).  Replaces the per-call r4_*.sh scripts.
#   tests      the whole GPU suite
#   bench      default bench.py line (20 steps) -> <tag>_bench.json
#   shadow     tools/comm_shadow.py -> <tag>_comm_shadow.json
#   prof       rocprofv3 kernel trace of 13 serial steps (2 warm-up + 8 timed + the 3 of bench.py's serial post-pass) + per-shape table -> <tag>_kernel_stats_serial.csv, <tag>_kernel_shapes_serial.txt
#   pmc        HBM traffic of the dominant kernels (separate --pmc passes) -> <tag>_pmc_traffic.json
#   ab_prev    this tree against the previous round's (a `git worktree` checked out as _r4 with its own built library), default bench
#              lines alternating on this box -> <tag>_ab_prev.txt
#   alone      per-launch tables: ticket vs static walk, attention kernels with / without e4m3 emission -> <tag>_walk_ab.txt, <tag>_attention_alone.txt
#   soak       200 timed steps of the default bench (ticket counters over ~20 k persistent launches) -> <tag>_bench_soak.json
set -u
batch=${1:-tests}; tag=${2:-r6}
mkdir -p gpurun_out
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
case "$batch" in
  tests)  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ;;
  bench)  timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 3000 gpurun_out/${tag}_bench.json ;;
  shadow) timeout 1500 python tools/comm_shadow.py --out gpurun_out/${tag}_comm_shadow.json 2>&1 | tail -60 ;;
  prof)
    rm -rf /tmp/prof && VIPANT_TOWER_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- \
        python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-last-block-check --no-encoder-alone > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_prof.err
    f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${tag}_kernel_stats_serial.csv
    python tools/kstats_shapes.py /tmp/prof 13 > gpurun_out/${tag}_kernel_shapes_serial.txt
    head -40 gpurun_out/${tag}_kernel_shapes_serial.txt ;;
  pmc)
    rm -rf /tmp/pmc_f /tmp/pmc_w
    timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-last-block-check --no-encoder-alone > /dev/null 2> gpurun_out/${tag}_pmc.err
    timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-last-block-check --no-encoder-alone > /dev/null 2>> gpurun_out/${tag}_pmc.err
    python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/${tag}_pmc_traffic.json | tail -40 ;;
  ab_prev)
    : > gpurun_out/${tag}_ab_prev.txt
    for r in 1 2; do
      for tree in _r4 .; do
        ( cd $tree && timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-last-block-check --no-encoder-alone 2> /dev/null | tail -1 | \
          python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tree', 'round $r', d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_ms'], 'ms dominant kernel')" ) >> gpurun_out/${tag}_ab_prev.txt
      done
    done
    cat gpurun_out/${tag}_ab_prev.txt ;;
  alone)
    timeout 600 python tools/walk_ab.py 21 > gpurun_out/${tag}_walk_ab.txt 2>&1; tail -8 gpurun_out/${tag}_walk_ab.txt
    timeout 600 python tools/attn_bench.py 40 > gpurun_out/${tag}_attention_alone.txt 2>&1; tail -9 gpurun_out/${tag}_attention_alone.txt ;;
  pmcsq)     # SQ counters + GRBM_GUI_ACTIVE (effective clock per kernel) of the headline step and of the configs[4] tower
    rm -rf /tmp/pmc_sq /tmp/pmc_sq5
    C="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
    VIPANT_TOWER_OVERLAP=0 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-last-block-check --no-encoder-alone > /dev/null 2> gpurun_out/${tag}_pmcsq.err
    python tools/pmc_sq.py /tmp/pmc_sq gpurun_out/${tag}_pmc_sq.json | tail -3
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_sq5 -o s -- python3 bench.py --script at --width 1024 --layers 24 --batch 1024 --fp8 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> gpurun_out/${tag}_pmcsq.err
    python tools/pmc_sq.py /tmp/pmc_sq5 gpurun_out/${tag}_pmc_sq_cfg5.json | tail -3 ;;
  soak)   timeout 900 python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-full-last-block-check --no-encoder-alone > gpurun_out/${tag}_bench_soak.json 2> gpurun_out/${tag}_bench_soak.err; tail -c 400 gpurun_out/${tag}_bench_soak.json ;;
  *) echo "unknown batch $batch"; exit 2 ;;
esac
