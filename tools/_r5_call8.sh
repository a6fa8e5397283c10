#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_full_configs.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
ROUNDS=3 bash tools/ab_bench.sh $O/ab "nsfwd|" "nofwdns|--call cpc_set_fwd_nsplit=0,-1"
bash tools/trace_variant.sh $O nsfwd
