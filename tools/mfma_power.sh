#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# The board's sustained matrix rate with no data movement (tools/probes/mfma_probe.hip) with rocm-smi power / clock samples beside it.
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/probes/mfma_probe $R/tools/probes/mfma_probe.hip
for cfg in "8 0" "8 2" "4 0" "8 1"; do
  set -- $cfg
  $R/tools/probes/mfma_probe 6 $1 $2 > /tmp/mf.log 2>&1 &
  pid=$!
  sleep 2.5
  pw=(); ck=()
  while kill -0 $pid 2>/dev/null; do
    line=$(rocm-smi --showpower --showclocks 2>/dev/null)
    p=$(echo "$line" | grep -i -E "Power \(W\)" | head -1 | grep -o -E "[0-9]+\.[0-9]+" | tail -1)
    c=$(echo "$line" | grep -i "sclk" | head -1 | grep -o -E "\([0-9]+Mhz\)" | grep -o -E "[0-9]+")
    [ -n "$p" ] && pw+=($p); [ -n "$c" ] && ck+=($c)
    sleep 0.3
  done
  wait $pid
  cat /tmp/mf.log
  echo "   power W samples: ${pw[*]}"
  echo "   sclk MHz samples: ${ck[*]}"
done
