#!/bin/bash
# PMC digest of the dominant conv kernel on one shape: pmc_conv_probe.sh <dtype> "<counter set 1>" "<counter set 2>" ...
# (each set is its own rocprofv3 --pmc pass; RPN 3x3 256->512 @256², batch 8)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
dt=$1; shift
for set in "$@"; do
  rm -rf /tmp/p1; timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/conv_one.py 8 256 256 256 512 3 1 3 $dt > /tmp/p1.log 2>&1
  python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/p1/**/*counter_collection.csv',recursive=True)
if not f: print("no output (timeout or unsupported counter set)")
else:
    acc=collections.defaultdict(list)
    dur=[]
    for r in csv.DictReader(open(f[0])):
        if 'k_conv_mfma_glds' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            if 'End_Timestamp' in r: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    out={k: round(sum(v)/len(v)) for k,v in acc.items()}
    if dur: out['avg_us']=round(sum(dur)/len(dur),1)
    print(out)
PY
done
