#!/bin/bash
# comm-shadow table of BASELINE configs[4]'s tower (audio ViT-L, e4m3, 1024 clips): the step with a stand-in all-reduce of a block's 50 MB
# bucket on a side stream, with one long workgroup per CU in the weight-gradient launches (default) and with twice / four times as many
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r6}
for k in 1 2 4; do
  echo "VIPANT_TN_SPLIT=$k"
  VIPANT_TN_SPLIT=$k timeout 900 python tools/comm_shadow.py --cfg5 --batch 1024 --steps 4 --rounds 2 --out gpurun_out/${tag}_comm_shadow_cfg5_split$k.json 2>&1 | grep "^(" 
done
