#!/bin/bash
# round 5, call 18: the batched weight-gradient reduction with a capped grid
ROUNDS=3 bash tools/ab_bench.sh gpurun_out/r5p "uncapped|" "red256|--call cpc_set_wgrad_reduce_groups=256" "red512|--call cpc_set_wgrad_reduce_groups=512" "red1024|--call cpc_set_wgrad_reduce_groups=1024"
