#!/bin/bash
# timing experiments on gemm_tn8_pp_kernel (VIPANT_TN8_DBG bits: see csrc/gemm_tn.hip)
for d in 0 1 2 4 8 16 6 12 14 30; do
  echo "DBG=$d"; VIPANT_TN8_DBG=$d python tools/tn8_bench.py 5 2>&1 | grep "M=323584 P=1024 Q=4096\|M=323584 P=1024 Q=1024" | sed 's/quant_mx32.*//'
done
