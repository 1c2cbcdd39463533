#!/bin/bash
# single-image latency A/Bs (interleaved, one process each): level-parallel region off / on, in the headline mode and in fp16
for dt in f32x3 f16; do BATCH=1 timeout 300 python tools/e2e_ab.py $dt level_parallel 0 2 4 30 2>&1 | tail -1; done
BATCH=2 timeout 300 python tools/e2e_ab.py f32x3 level_parallel 0 2 3 20 2>&1 | tail -1
BATCH=8 timeout 300 python tools/e2e_ab.py f32x3 level_parallel 0 8 3 10 2>&1 | tail -1
