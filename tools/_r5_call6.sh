#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
ROUNDS=3 bash tools/ab_bench.sh $O/ab "fused|" "twopass|--call cpc_set_nce_fused=0" "fused64|--call cpc_set_conv_small_tile=64"
bash tools/trace_variant.sh $O fused
