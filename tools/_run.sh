cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu > gpurun_out/r4t/tests_full.log
grep -E "passed|failed|error|Error" gpurun_out/r4t/tests_full.log | tail -5
