#!/bin/bash
# round-4 first GPU call: parity corners, the vendor library's kernel names on the K = 768 shapes, baseline bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
python -m pytest tests/test_model_gpu.py -q -x -k "trajectory_golden" 2>&1 | tail -5 > gpurun_out/r4a_traj.log
cp gpurun_out/parity_observed.jsonl gpurun_out/r4a_traj_observed.jsonl
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -3 > gpurun_out/r4a_gemm_tests.log
python tools/lib_gemm_ref.py 7 > gpurun_out/r4a_lib_gemm.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r4a_libtrace -o lib -- python3 tools/lib_gemm_ref.py 3 > gpurun_out/r4a_libtrace.log 2>&1
python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
ls -R gpurun_out/r4a_libtrace | head -30
find gpurun_out/r4a_libtrace -name "*kernel_stats*" | head
cat gpurun_out/r4a_traj.log gpurun_out/r4a_gemm_tests.log gpurun_out/r4a_lib_gemm.txt
cat gpurun_out/r4a_bench.json
