#!/bin/bash
# One logged gpurun call: tools/gpu.sh <timeout-seconds> '<command>'.  Appends the call and the budget line (GPU-minutes left) to
# profiles/r6_gpurun_log.txt so that the round's use of the 270 GPU-minutes can be checked afterwards.
t=$1; shift
/usr/local/graft/bin/gpurun --timeout "$t" -- "$*"; rc=$?
left=$(/usr/local/graft/bin/gpurun --status | python3 -c "import json,sys; d=json.load(sys.stdin); print(d.get('gpu_minutes_left'))")
echo "$(date -u +%FT%TZ) rc=$rc timeout=${t}s gpu_minutes_left=$left cmd: $*" >> "$(dirname "$0")/../profiles/r6_gpurun_log.txt"
exit $rc
