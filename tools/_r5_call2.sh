#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_dist.py tests/test_gpu_harness.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
ROUNDS=3 bash tools/ab_bench.sh $O/ab "open|" "closed|--no-pipeline-tail"
bash tools/trace_variant.sh $O open
