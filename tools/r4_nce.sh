#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 1 0 1 0; do
  VIPANT_NCE_ROWS=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-last-block-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('rows=$v', d['ms_per_step'], d['value'])"
done
for b in 512 64; do for v in 1 0; do VIPANT_NCE_ROWS=$v python tools/nce_run.py $b 2>&1 | grep InfoNCE | sed "s/^/rows=$v /"; done; done
