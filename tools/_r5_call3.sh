#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
ROUNDS=3 bash tools/ab_bench.sh $O/ab "closed|--no-pipeline-tail" "open|" "open_prio|ENV:CPC_SIDE_PRIORITY=2" "early|--call cpc_set_tail_schedule=1" "early_prio|ENV:CPC_SIDE_PRIORITY=2 --call cpc_set_tail_schedule=1"
bash tools/trace_variant.sh $O open
