#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VIPANT_ATTN_FWD=16
timeout 300 python tools/mha_check.py stream 2>&1 | grep -v Warn | grep "S=3\|audio\|ViT-L"
VIPANT_ATTN_BWD=1 timeout 300 python tools/mha_check.py resident 2>&1 | grep "audio\|ViT-L"
timeout 300 python tools/mha_check.py stream 2>&1 | grep "audio\|ViT-L"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "mha" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "block_golden or end_to_end_golden" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ms/step', d['ms_per_step'], 'full last block', d['full_last_block']['ms_per_step'], 'frac', d['step_mfma_frac'])"
VIPANT_ATTN_BWD=1 python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench (resident bwd) ms/step', d['ms_per_step'], 'full last block', d['full_last_block']['ms_per_step'])"
