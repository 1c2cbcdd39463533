cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -5 > gpurun_out/r4s/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4s/smoke.log 2>&1
MODES_ONLY=1 bash tools/refresh_profiles.sh r04 f16 > gpurun_out/r4s/refresh_f16.log 2>&1
python bench.py --steps 10 --warmup 3 --arch resnet50 --dtype f16 --no-cpu-baseline --no-other-modes > gpurun_out/profiles_r04/r04_bench_n1_resnet50_f16.json 2>/dev/null
cat gpurun_out/r4s/tests.log; tail -1 gpurun_out/r4s/smoke.log; cut -c1-200 gpurun_out/profiles_r04/r04_bench_n1_f16.json
