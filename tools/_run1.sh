timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/test_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|Error|assert" gpurun_out/test_gpu.log | tail -8
timeout 300 python bench.py --steps 30 --warmup 10 > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err; tail -2 gpurun_out/bench_c.err; cut -c1-330 gpurun_out/bench_c.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_v15 -- python bench.py --steps 5 --warmup 2 > gpurun_out/prof_v15.log 2>&1
