#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
ROUNDS=3 bash tools/ab_bench.sh $O/ab "open|" "small64|--call cpc_set_conv_small_tile=64"
bash tools/trace_variant.sh $O small64 --call cpc_set_conv_small_tile=64
