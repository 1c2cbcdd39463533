#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# Board power and shader clock while ONE convolution shape runs back to back for a few seconds, per compute mode:
# is the sustained clock under matrix load (1.45-1.7 GHz against 2.4 GHz nominal) a power limit?
#   power_probe.sh [modes...]      (default: f32 f32x3 f32s f16), RPN 3x3 256->512 at 256^2, batch 8
R=$(cd "$(dirname "$0")/.." && pwd)
echo "# power cap / limits:"; rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|power" | head -4
for m in ${@:-f32 f32x3 f32s f16}; do
  it=2500; [ $m = f32 ] && it=700; [ $m = f16 ] && it=6000
  MRCNN_PP=${MRCNN_PP:-1} timeout 120 python $R/tools/conv_one.py 8 256 256 256 512 3 1 $it $m > /tmp/pw_$m.log 2>&1 &
  pid=$!
  sleep 4      # library load + warm-up
  pw=(); ck=()
  while kill -0 $pid 2>/dev/null; do
    line=$(rocm-smi --showpower --showclocks 2>/dev/null)
    p=$(echo "$line" | grep -i -E "Power \(W\)|Socket Power" | head -1 | grep -o -E "[0-9]+\.[0-9]+" | tail -1)
    c=$(echo "$line" | grep -i "sclk" | head -1 | grep -o -E "\([0-9]+Mhz\)" | grep -o -E "[0-9]+")
    [ -n "$p" ] && pw+=($p); [ -n "$c" ] && ck+=($c)
    sleep 0.3
  done
  wait $pid
  echo "$m: $(tail -1 /tmp/pw_$m.log)"
  echo "   power W samples: ${pw[*]}"
  echo "   sclk MHz samples: ${ck[*]}"
done
