#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# LDS bank conflicts (and VALU / LDS instruction counts) of the halo kernel per layer shape, round-3 tile geometries
# (MRCNN_HALO_GEO=0) vs round 4 (=1):   pmc_halo_geo_probe.sh <dtype> "<b h w cin cout k stride>" ...
# Each shape is its own rocprofv3 --pmc pass (kernel-trace only), per MI355X_MICROARCH.md.
export TMPDIR=/tmp; R=$(pwd); cd /tmp
dt=$1; shift
for shape in "$@"; do
for geo in 0 1; do
  rm -rf /tmp/p1; MRCNN_HALO_GEO=$geo timeout 180 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/conv_one.py $shape 3 $dt > /tmp/p1.log 2>&1
  GEO=$geo SHAPE="$shape" python - <<'PY'
import csv,glob,collections,os
f=glob.glob('/tmp/p1/**/*counter_collection.csv',recursive=True)
if not f: print("no output"); print(open('/tmp/p1.log').read()[-600:])
else:
    acc=collections.defaultdict(list); name=None
    for r in csv.DictReader(open(f[0])):
        if 'k_conv_halo' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); name=r['Kernel_Name'][:60]
    out={k: round(sum(v)/len(v)) for k,v in acc.items()}
    if out.get('SQ_LDS_IDX_ACTIVE'): out['conflict_ratio']=round(out['SQ_LDS_BANK_CONFLICT']/out['SQ_LDS_IDX_ACTIVE'],4)
    print('[%s] geo=%s'%(os.environ['SHAPE'],os.environ['GEO']), name, out, flush=True)
PY
done
done
