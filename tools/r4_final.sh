#!/bin/bash
# round-4 measurement batch: tests, bench lines (headline, AT, cfg5), serial rocprof profile, library A/B, SQ counters, attention alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/parity_observed.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/r4z_gpu_tests.log
cp $O/parity_observed.jsonl $O/r4z_parity_observed.jsonl 2>/dev/null
python bench.py --steps 20 --warmup 3 > $O/r4z_bench.json 2> $O/r4z_bench.err
python bench.py --script at --steps 8 --warmup 2 > $O/r4z_bench_at.json 2>> $O/r4z_bench.err
python bench.py --script at --steps 8 --warmup 2 --full-last-block > $O/r4z_bench_at_full.json 2>> $O/r4z_bench.err
python bench.py --script at --width 1024 --layers 24 --batch 1024 --recompute-mlp --steps 3 --warmup 1 > $O/r4z_cfg5_bf16.json 2>> $O/r4z_bench.err
python bench.py --script at --width 1024 --layers 24 --batch 1024 --recompute-mlp --fp8 --steps 3 --warmup 1 > $O/r4z_cfg5_fp8.json 2>> $O/r4z_bench.err
python bench.py --script at --width 1024 --layers 24 --batch 1024 --recompute-mlp --fp8 --full-last-block --steps 3 --warmup 1 > $O/r4z_cfg5_fp8_full.json 2>> $O/r4z_bench.err
python bench.py --script at --width 1024 --layers 24 --batch 1024 --recompute-mlp --full-last-block --steps 3 --warmup 1 > $O/r4z_cfg5_bf16_full.json 2>> $O/r4z_bench.err
python tools/rows_bench.py > $O/r4z_rows_bench.txt 2>&1
python tools/lib_ab.py 9 > $O/r4z_lib_ab.txt 2>&1
for v in "16 3" "32 3" "16 1" "16 4"; do set -- $v; VIPANT_ATTN_FWD=$1 VIPANT_ATTN_BWD=$2 python tools/mha_check.py "fwd$1/bwd$2" 2>&1 | grep "audio\|ViT-L" ; done > $O/r4z_attention_alone.txt
rm -rf $O/r4z_prof
VIPANT_TOWER_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4z_prof -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-last-block-check > $O/r4z_bench_serial.json 2>> $O/r4z_bench.err
python3 tools/kstats_shapes.py $O/r4z_prof 10 > $O/r4z_kernel_shapes_serial.txt 2>&1
find $O/r4z_prof -name "*kernel_stats.csv" -exec cp {} $O/r4z_kernel_stats_serial.csv \;
rm -rf $O/r4z_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4z_prof2 -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-last-block-check > $O/r4z_bench_prof.json 2>> $O/r4z_bench.err
python3 tools/kstats_shapes.py $O/r4z_prof2 10 > $O/r4z_kernel_shapes.txt 2>&1
find $O/r4z_prof2 -name "*kernel_stats.csv" -exec cp {} $O/r4z_kernel_stats.csv \;
rm -rf $O/r4z_prof2
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-last-block-check > $O/r4z_pmc_run.log 2>&1
python3 tools/pmc_sq.py /tmp/pmc_sq $O/r4z_pmc_sq.json >> $O/r4z_pmc_run.log 2>&1
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-last-block-check >> $O/r4z_pmc_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-last-block-check >> $O/r4z_pmc_run.log 2>&1
python3 tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/r4z_pmc_traffic.json >> $O/r4z_pmc_run.log 2>&1
cat $O/r4z_gpu_tests.log; cat $O/r4z_bench.json | head -c 400; echo; for f in at at_full cfg5_bf16 cfg5_fp8 cfg5_fp8_full cfg5_bf16_full; do python3 -c "import json,sys; d=json.load(open('$O/r4z_bench_'+'$f'+'.json')) if '$f'.startswith('at') else json.load(open('$O/r4z_'+'$f'+'.json')); print('$f', d['ms_per_step'], d['value'], d.get('peak_mem_gb'), d['loss'])"; done; cat $O/r4z_lib_ab.txt $O/r4z_attention_alone.txt
