#!/bin/bash
# round 6: the 128 x 128 wave tile (one wave per SIMD) on layer 1's forward (pipeline 7) and data gradient (+ 8): parity, then in-step A/B
export TMPDIR=/tmp
O=gpurun_out/r6_w128
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_encoder.py -q -x -k "dma_pipelines" 2>&1 | tail -3
ROUNDS=2 bash tools/ab_bench.sh $O/ab "pipe2|--no-b256 --no-config4 --dma-pipeline 2" "pipe7|--no-b256 --no-config4 --dma-pipeline 7" "pipe10|--no-b256 --no-config4 --dma-pipeline 10" "pipe15|--no-b256 --no-config4 --dma-pipeline 15" | tee $O/ab_step.txt
