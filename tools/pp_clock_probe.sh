#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# Sustained clock and MFMA-busy of the ping-pong kernel and its ablations on the largest layer (RPN 3x3 256->512 at 256², batch 8):
#   pp_clock_probe.sh <f16|f32s|f32x3> ["<dbg bits> ..."]   dbg: 0 shipped, 4 no DMA in the main loop, 8 no fragment reads, 256 every activation DMA from the same 128 bytes (L1-hot)
# clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
for dbg in ${2:-0 4 8 12 256}; do
  rm -rf /tmp/p1; MRCNN_PP=1 MRCNN_PP_SPLIT=1 MRCNN_PP_MIN_TILES=1 MRCNN_PP_MIN_FILL=0 MRCNN_PP_DBG=$dbg timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/conv_one.py 8 256 256 256 512 3 1 3 $1 > /tmp/p1.log 2>&1
  DBG=$dbg python - <<'PY'
import csv,glob,collections,os
f=glob.glob('/tmp/p1/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'k_conv' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
kt=glob.glob('/tmp/p1/**/*kernel_trace.csv',recursive=True)
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt[0])) if 'k_conv' in r['Kernel_Name']]
g=sum(acc['GRBM_GUI_ACTIVE'])/len(acc['GRBM_GUI_ACTIVE']); m=sum(acc['SQ_VALU_MFMA_BUSY_CYCLES'])/len(acc['SQ_VALU_MFMA_BUSY_CYCLES']); us=sum(d)/len(d)
print(f"dbg={os.environ['DBG']}: {us:8.1f} us  clock {g/8/us/1e3:5.2f} GHz  mfma util {m/(1024*g/8):5.3f}", flush=True)
PY
done
