#!/bin/bash
export MRCNN_TEST_KNOBS=1      # arm the test / measurement knobs (csrc/common.h)
# Regenerates everything under profiles/ for round $1 (default r02) on a GPU box:
#   gpurun --timeout 3300 -- 'bash tools/refresh_profiles.sh r04'
# Outputs land in gpurun_out/profiles_<round>/ (merged back by gpurun); copy them into profiles/ and commit.
# Counter passes follow MI355X_MICROARCH.md: --pmc in its own run with --kernel-trace only, FETCH_SIZE and WRITE_SIZE separately.
set -u
RND=${1:-r06}
MODES=${2:-"f32x3 f16 f32s f32"}
R=$(pwd)
OUT=$R/gpurun_out/profiles_$RND
mkdir -p $OUT
export TMPDIR=/tmp
for dt in $MODES; do
  [ $dt = f32x3 ] || python bench.py --steps 10 --warmup 3 --dtype $dt --no-cpu-baseline --no-other-modes > $OUT/${RND}_bench_n1_$dt.json 2> $OUT/bench_$dt.err
  python tools/conv_layer_table.py $dt 3 8 > $OUT/${RND}_conv_shapes_$dt.txt 2>/dev/null
done
python tools/conv_layer_table.py f32x3 5 1 > $OUT/${RND}_conv_shapes_f32x3_batch1.txt 2>/dev/null
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-modes --no-kernel-events --no-live-probe"
for dt in $MODES; do
  rm -rf /tmp/prof_$dt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dt -o p -- python $R/bench.py --steps 5 --warmup 2 --dtype $dt --no-cpu-baseline --no-other-modes --no-live-probe > $OUT/stats_$dt.json 2> $OUT/stats_$dt.err
  db=$(find /tmp/prof_$dt -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $db $OUT/${RND}_kernel_stats_$dt.csv
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1)); rm -rf /tmp/pmc_${dt}_$i
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${dt}_$i -o p -- $BENCH --dtype $dt > /dev/null 2> $OUT/pmc_${dt}_$i.err
  done
  python $R/tools/pmc_digest.py $OUT/${RND}_pmc_kernels_$dt.json $OUT/${RND}_kernel_stats_$dt.csv /tmp/pmc_${dt}_1 /tmp/pmc_${dt}_2 /tmp/pmc_${dt}_3 /tmp/pmc_${dt}_4 /tmp/pmc_${dt}_5 > $OUT/${RND}_pmc_kernels_$dt.txt
  python - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench                                   # the SAME class table bench.py names its dominant kernel by
d = json.load(open("$OUT/${RND}_pmc_kernels_$dt.json"))["kernels"]
conv = {k: v for k, v in d.items() if (k.startswith("k_conv") or k.startswith("k_bneck")) and "hbm_bytes_per_launch" in v}
cls = {}
for name, tags in bench.CLASS_KERNELS.items():
    if name == "128x256tail":
        continue
    sel = {k: v for k, v in conv.items() if any(t in k for t in tags)}
    if sel:
        cls[name] = sel
dom = max(cls, key=lambda c: sum(v.get("percent_of_gpu_time", 0) for v in cls[c].values()))
sel = cls[dom]
n = sum(v["launches_sampled"] for v in sel.values())
json.dump({"tile_class": dom, "kernels": sorted(sel), "launches_sampled": n,
           "percent_of_gpu_time": round(sum(v.get("percent_of_gpu_time", 0) for v in sel.values()), 2),
           "hbm_bytes_per_launch_corrected": round(sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in sel.values()) / n),
           "note": "dominant conv tile CLASS of bench.py --dtype $dt (the class bench.py's roofline names: largest summed share of GPU time), launch-weighted over its "
                   "instantiations; 2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes (tools/refresh_profiles.sh)"},
          open("$OUT/${RND}_pmc_traffic_$dt.json", "w"), indent=1)
PY
done
cd $R
# MODES_ONLY=1: refresh the per-mode files of $MODES only (a kernel of one mode changed after the round's full refresh)
if [ "${MODES_ONLY:-0}" = 1 ]; then
  for dt in $MODES; do [ $dt = f32x3 ] || cut -c1-200 $OUT/${RND}_bench_n1_$dt.json; cat $OUT/${RND}_pmc_kernels_$dt.txt; done
  exit 0
fi
timeout 300 python tools/conv_ab.py 3 10 1 f16 > $OUT/${RND}_conv_ab_f16.txt 2>/dev/null
# round 5 (fp16 mode): the fused identity bottleneck against the three launches, its phase ablation, the 3x3 halo-tile kernel against the ping-pong kernel
timeout 300 python tools/bneck_ab.py 8 20 2>/dev/null | grep -v amdgpu > $OUT/${RND}_bneck_ab_f16.txt
{ BNECK_ONLY=C4 timeout 300 bash tools/bneck_phases.sh 8; BNECK_ONLY=C2 timeout 300 bash tools/bneck_phases.sh 8; } 2>/dev/null | grep -v amdgpu > $OUT/${RND}_bneck_phases_f16.txt
timeout 300 python tools/c3h_ab.py 3 10 2>/dev/null | grep -v amdgpu > $OUT/${RND}_c3h_ab_f16.txt
# round 6 (fp16 mode): C4's 22 identity blocks as ONE launch with per-tile neighbour counters against one launch per block (kernel level and whole model)
{ timeout 300 python tools/bneck_stage_ab.py 8 22 20 3 2>/dev/null | grep -v amdgpu; timeout 300 python tools/e2e_ab.py f16 conv_bneck_stage 0 1 4 10 2>/dev/null | tail -1; } > $OUT/${RND}_bneck_stage_ab_f16.txt
{ for kv in "conv_bneck 0 1" "conv_c3h 0 1"; do timeout 300 python tools/e2e_ab.py f16 $kv 3 10 2>/dev/null | tail -1; done; BATCH=1 timeout 300 python tools/e2e_ab.py f16 conv_bneck 0 1 3 20 2>/dev/null | tail -1; } > $OUT/${RND}_e2e_ab_f16.txt
for dt in f32x3 f32s; do timeout 300 python tools/halo_ab.py 3 10 $dt 2>/dev/null | grep -v amdgpu > $OUT/${RND}_halo_ab_$dt.txt; done
timeout 200 python tools/halo_ablate.py f32x3 2>/dev/null | grep -v amdgpu > $OUT/${RND}_halo_ablate_f32x3.txt
# round 4: the halo kernel's tile geometries (MRCNN_HALO_GEO=0 = round 3's one-row tiles, five staging pieces, pitch W + 2) — timing and LDS conflicts per layer
for g in 0 1; do MRCNN_HALO_GEO=$g timeout 300 python tools/halo_ab.py 3 10 f32x3 2>/dev/null | grep -v amdgpu > $OUT/${RND}_halo_ab_geo${g}_f32x3.txt; done
timeout 600 bash tools/pmc_halo_geo_probe.sh f32x3 "8 256 256 256 512 3 1" "8 256 256 256 256 3 1" "800 14 14 256 256 3 1" "8 64 64 256 256 3 1" "8 32 32 512 512 3 1" "8 32 32 256 256 3 1" > $OUT/${RND}_pmc_halo_geo_f32x3.txt 2>&1
python tools/lds_bank_model.py > $OUT/${RND}_lds_bank_model.txt 2>&1
# the split modes' calibration report on the headline model (per tensor group: max |a|, exponent, counters)
timeout 300 python tools/split_report.py > $OUT/${RND}_split_report_f32x3.txt 2>/dev/null
bash tools/pmc_halo_probe.sh f32x3 "8 256 256 256 512 3 1" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" > $OUT/${RND}_pmc_probe_rpn3x3_f32x3.txt 2>&1
[ -x tools/probes/vmem_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/vmem_probe tools/probes/vmem_probe.hip 2>/dev/null
[ -x tools/probes/vmem_probe ] && timeout 120 ./tools/probes/vmem_probe 2048 > $OUT/${RND}_vmem_probe.txt 2>&1
python tools/split_scale_curve.py 2>/dev/null | grep -v amdgpu > $OUT/${RND}_split_scale_curve.txt
# stage-attributed trace: roctx ranges with the reference's signpost names (marker trace + kernel trace, no counters)
( cd /tmp && rm -rf /tmp/mk && timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/mk -o m -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-modes --no-kernel-events --no-live-probe > /dev/null 2> $OUT/marker.err; f=$(find /tmp/mk -name "*marker*stats*.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${RND}_marker_stats_f32x3.csv; f=$(find /tmp/mk -name "*marker_api_trace.csv" | head -1); [ -n "$f" ] && head -80 $f > $OUT/${RND}_marker_trace_head_f32x3.csv )
timeout 200 bash tools/power_probe.sh > $OUT/${RND}_power_probe.txt 2>&1
timeout 200 bash tools/mfma_power.sh > $OUT/${RND}_mfma_power.txt 2>&1
timeout 400 python tools/fp64_trunk_parity.py --out $OUT/${RND}_fp64_trunk_parity.json > $OUT/fp64.log 2>&1
# round 4 (late): canonical K chunks and the halo kernel's latency form — knob A/Bs on single layers; where a single image's time goes
{ for kv in "conv_ksplit 0 1" "conv_kchunk 0 1" "halo_lat 0 1" "halo_lat 0 2"; do timeout 200 python tools/knob_ab.py $kv f32x3 3 20 all 2>/dev/null | grep -v amdgpu; done; } > $OUT/${RND}_knob_ab_f32x3.txt
( cd /tmp && export TMPDIR=/tmp
  for b in 1 8; do rm -rf /tmp/tr$b; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$b -o t -- python $R/bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --e2e-images 0 --no-other-modes --no-kernel-events --no-live-probe > /dev/null 2>&1; done
  { echo "# batch 1"; python $R/tools/trace_gaps.py /tmp/tr1; echo "# batch 8"; python $R/tools/trace_gaps.py /tmp/tr8; } > $OUT/${RND}_trace_gaps_f32x3.txt
  python $R/tools/trace_gaps.py /tmp/tr1 --list > $OUT/${RND}_trace_b1_kernels_f32x3.txt )
# per-GPU slices of the other BASELINE configs (configs[2]: ResNet50; configs[3]: fp16; configs[4]: 1536², 2 classes, pre_nms 12000) and the batch sweep
python bench.py --steps 10 --warmup 3 --arch resnet50 --no-cpu-baseline --no-other-modes > $OUT/${RND}_bench_n1_resnet50.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --arch resnet50 --dtype f16 --no-cpu-baseline --no-other-modes > $OUT/${RND}_bench_n1_resnet50_f16.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --size 1536 --num-classes 2 --pre-nms 12000 --no-cpu-baseline --no-other-modes > $OUT/${RND}_bench_n1_config5_1536.json 2>/dev/null
for b in 1 2 4 16 32; do python bench.py --steps 10 --warmup 3 --batch $b --no-cpu-baseline --no-other-modes --no-kernel-events > $OUT/${RND}_bench_n1_batch$b.json 2>/dev/null; done
# 64-image end-to-end agreement with the CPU oracle (about 6 minutes of host time)
python bench.py --steps 5 --warmup 2 --e2e-images 64 --no-kernel-events 2>/dev/null | python -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read())['parity_e2e'], indent=1))" > $OUT/${RND}_parity_e2e_64.json
timeout 900 python tools/soak_determinism.py 200 > $OUT/${RND}_soak_determinism.txt 2>&1
# the default bench line last: it reports the counters / probe figures of THIS refresh (bench.py reads them from profiles/)
cp $OUT/${RND}_pmc_traffic_*.json $OUT/${RND}_mfma_power.txt $R/profiles/ 2>/dev/null
python bench.py --steps 10 --warmup 3 > $OUT/${RND}_bench_n1.json 2> $OUT/bench_default.err
head -5 $OUT/${RND}_kernel_stats_f32x3.csv | cut -c1-160
cut -c1-300 $OUT/${RND}_bench_n1.json
cat $OUT/${RND}_pmc_kernels_f32x3.txt
