#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_train_step.py tests/test_gpu_modules.py tests/test_gpu_fused_step.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
ROUNDS=3 bash tools/ab_bench.sh $O/ab "fused|" "twopass|--call cpc_set_nce_fused=0" "fused64|--call cpc_set_conv_small_tile=64" "f64ns|--call cpc_set_conv_small_tile=64 --call cpc_set_dgrad_nsplit=256"
bash tools/trace_variant.sh $O fused64 --call cpc_set_conv_small_tile=64
