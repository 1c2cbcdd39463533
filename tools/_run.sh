cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout 2400 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | grep -v amdgpu | tail -14 > gpurun_out/r4m/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4m/smoke.log 2>&1
cat gpurun_out/r4m/tests.log; tail -2 gpurun_out/r4m/smoke.log
