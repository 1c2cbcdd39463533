#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# phase ablation of the fused bottleneck (MRCNN_BNECK_DBG: 1 / 2 / 4 = phase A / B / C cut to one step; results invalid, timing only)
for d in 0 1 2 4 3 5 6 7; do
  echo "== MRCNN_BNECK_DBG=$d"
  MRCNN_BNECK_DBG=$d timeout 120 python tools/bneck_ab.py ${1:-8} 20 2>&1 | grep -E "C[234] "
done
