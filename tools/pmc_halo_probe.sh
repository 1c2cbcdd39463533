#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# PMC digest of one 3x3 shape in a split mode, 128-row kernel (MRCNN_HALO=0) vs halo kernel (MRCNN_HALO=1):
#   pmc_halo_probe.sh <dtype> "<b h w cin cout k stride>" "<counter set 1>" "<counter set 2>" ...
# (each set is its own rocprofv3 --pmc pass, kernel-trace only)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
dt=$1; shift
shape=$1; shift
for halo in 0 1; do
for set in "$@"; do
  rm -rf /tmp/p1; MRCNN_HALO=$halo MRCNN_PP=0 timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/conv_one.py $shape 3 $dt > /tmp/p1.log 2>&1
  HALO=$halo python - <<'PY'
import csv,glob,collections,os
f=glob.glob('/tmp/p1/**/*counter_collection.csv',recursive=True)
if not f: print("no output (timeout or unsupported counter set)"); print(open('/tmp/p1.log').read()[-600:])
else:
    acc=collections.defaultdict(list)
    name=None
    for r in csv.DictReader(open(f[0])):
        if 'k_conv' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); name=r['Kernel_Name'][:48]
    out={k: round(sum(v)/len(v)) for k,v in acc.items()}
    kt=glob.glob('/tmp/p1/**/*kernel_trace.csv',recursive=True)
    if kt:
        d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt[0])) if 'k_conv' in r['Kernel_Name']]
        if d: out['avg_us']=round(sum(d)/len(d),1)
    if 'GRBM_GUI_ACTIVE' in out and 'avg_us' in out:
        clk=out['GRBM_GUI_ACTIVE']/8/out['avg_us']/1e3
        out['clock_GHz']=round(clk,3)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in out: out['mfma_util']=round(out['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*out['GRBM_GUI_ACTIVE']/8),3)
    print('halo=%s'%os.environ['HALO'], name, out, flush=True)
PY
done
done
