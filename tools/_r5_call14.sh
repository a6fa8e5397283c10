#!/bin/bash
# round 5, call 14: how repeatable are the in-step marker timings (three processes, rows printed)
mkdir -p gpurun_out/r5m
for i in 1 2 3; do
  CPC_BENCH_IN_STEP_ROWS=1 python bench.py --no-cpu-baseline --no-b256 --steps 30 --warmup 8 --sustained-seconds 1 > gpurun_out/r5m/bench$i.json 2> gpurun_out/r5m/err$i.txt
  grep "in-step marker" gpurun_out/r5m/err$i.txt
  python -c "
import json; d=json.load(open('gpurun_out/r5m/bench$i.json')); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['in_step_us'], d['roofline']['frac'], d['roofline']['standalone'])"
done
