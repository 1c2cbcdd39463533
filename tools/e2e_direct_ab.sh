#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# end-to-end A/B of the fp32-tensor epilogue of the 128-row kernels: block-staged (MRCNN_DIRECT=1) vs wave-private tiles (=2)
for r in 1 2 3; do for d in 1 2; do
  MRCNN_DIRECT=$d python bench.py --no-cpu-baseline --no-other-modes --steps 15 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('MRCNN_DIRECT=$d', j['value'], j['ms_per_step'])"
done; done
