#!/bin/bash
# round 5, call 12b: 20 prediction steps on hardware
mkdir -p gpurun_out/r5l
python -m pytest tests/test_gpu_fused_step.py -x -q -m gpu -k "20_prediction" > gpurun_out/r5l/pytest2.log 2>&1
echo "pytest rc $?"
tail -5 gpurun_out/r5l/pytest2.log
