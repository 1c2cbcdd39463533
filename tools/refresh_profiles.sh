#!/bin/bash
# Regenerates everything under profiles/ for round $1 (default r01) on a GPU box:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
# Outputs land in gpurun_out/profiles_<round>/ (merged back by gpurun); copy them into profiles/ and commit.
set -u
RND=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/profiles_$RND
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 > $OUT/${RND}_bench_n1.json 2> $OUT/bench_f32.err
python bench.py --steps 10 --warmup 3 --dtype f16 --no-cpu-baseline --no-other-modes > $OUT/${RND}_bench_n1_f16.json 2> $OUT/bench_f16.err
python bench.py --steps 10 --warmup 3 --dtype f32s --no-cpu-baseline --no-other-modes > $OUT/${RND}_bench_n1_f32s.json 2> $OUT/bench_f32s.err
cd /tmp
for dt in f32 f16 f32s; do
  rm -rf /tmp/prof_$dt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$dt -o p -- python $R/bench.py --steps 5 --warmup 2 --dtype $dt --no-cpu-baseline --no-other-modes > $OUT/stats_$dt.json 2> $OUT/stats_$dt.err
  db=$(find /tmp/prof_$dt -name "*.db" | head -1)
  sfx=""; [ $dt != f32 ] && sfx="_$dt"
  python $R/tools/rocpd_summary.py $db $OUT/${RND}_kernel_stats$sfx.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes --no-kernel-events > /dev/null 2> $OUT/pmc_$c.err
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/${RND}_pmc_traffic.json
head -4 $OUT/${RND}_kernel_stats.csv | cut -c1-160
cut -c1-400 $OUT/${RND}_bench_n1.json
