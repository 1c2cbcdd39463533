cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
timeout 3000 python bench.py --steps 5 --warmup 2 --e2e-images 256 --no-kernel-events --no-other-modes 2>/dev/null | python -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read())['parity_e2e'], indent=1))" > gpurun_out/r5p/r04_parity_e2e_256.json
head -30 gpurun_out/r5p/r04_parity_e2e_256.json
