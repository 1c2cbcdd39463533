#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# PMC digest of the conv kernels on one shape, 128-row vs ping-pong:
#   pmc_conv_probe.sh <dtype> "<b h w cin cout k stride>" "<counter set 1>" "<counter set 2>" ...
# (each set is its own rocprofv3 --pmc pass, kernel-trace only; both kernels: MRCNN_PP=0 / 1 with the size gate off)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
dt=$1; shift
shape=$1; shift
for pp in 0 1; do
for set in "$@"; do
  rm -rf /tmp/p1; MRCNN_PP=$pp MRCNN_PP_MIN_TILES=1 MRCNN_PP_MIN_FILL=0 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/conv_one.py $shape 3 $dt > /tmp/p1.log 2>&1
  PP=$pp python - <<'PY'
import csv,glob,collections,os
f=glob.glob('/tmp/p1/**/*counter_collection.csv',recursive=True)
if not f: print("no output (timeout or unsupported counter set)"); print(open('/tmp/p1.log').read()[-600:])
else:
    acc=collections.defaultdict(list)
    name=None
    for r in csv.DictReader(open(f[0])):
        if 'k_conv' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); name=r['Kernel_Name'][:60]
    out={k: round(sum(v)/len(v)) for k,v in acc.items()}
    kt=glob.glob('/tmp/p1/**/*kernel_trace.csv',recursive=True)
    if kt:
        d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt[0])) if 'k_conv' in r['Kernel_Name']]
        if d: out['avg_us']=round(sum(d)/len(d),1)
    print('pp=%s'%os.environ['PP'], name, out, flush=True)
PY
done
done
