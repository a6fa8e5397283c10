#!/bin/bash
# Regenerate the round's committed profile files in one go (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh [tag=r2]
# Writes to gpurun_out/profiles_<tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
TAG=${1:-r2}
O=gpurun_out/profiles_$TAG
mkdir -p $O
export TMPDIR=/tmp
# 1. the default bench line (probes, cpu baseline, bf16 secondary)
timeout 400 python bench.py > $O/bench_$TAG.json 2> $O/bench.err
# 2. kernel trace of the fp32 step (eager launch, no probes): per-kernel totals, last full step, step period
rm -rf /tmp/prof_fp32
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fp32 -o res -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probes --launch eager --sustained-seconds 0 > $O/bench_trace.json 2> $O/trace.err
db=$(find /tmp/prof_fp32 -name "*.db" | head -1)
python tools/rocpd_stats.py $db $O/${TAG}_step_kernel_stats.csv > /dev/null
python tools/step_timeline.py $db --stats > $O/${TAG}_step_stats.txt
python tools/step_timeline.py $db > $O/${TAG}_step_timeline.txt
python tools/step_timeline.py $db --period > $O/${TAG}_step_period.txt
# 3. the same for the bf16-storage variant
rm -rf /tmp/prof_bf16
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o res -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-probes --launch eager --sustained-seconds 0 > $O/bench_trace_bf16.json 2>> $O/trace.err
db=$(find /tmp/prof_bf16 -name "*.db" | head -1)
python tools/step_timeline.py $db --stats > $O/${TAG}_bf16_step_stats.txt
python tools/step_timeline.py $db > $O/${TAG}_bf16_step_timeline.txt
# 4. kernel trace of the roofline probes themselves (the averages bench.py's hip events must agree with)
rm -rf /tmp/prof_probe
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_probe -o res -- python tools/probe_kernels.py 64 > /dev/null 2>> $O/trace.err
db=$(find /tmp/prof_probe -name "*.db" | head -1)
python tools/rocpd_stats.py $db $O/${TAG}_probe_kernel_stats.csv > /dev/null
python -c "
import json; d=json.load(open('$O/bench_$TAG.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['host_enqueue_ms_per_step'], d['config']['launch'])
for k in ('roofline','roofline_hbm_layer','roofline_scoring','roofline_gru'):
    r=d.get(k,{}); print(k, r.get('achieved'), r.get('frac'), r.get('ms_per_launch'), r.get('forward'), r.get('backward'))
print(d['cpu_baseline']['value'], d.get('bf16_storage_variant',{}).get('value'))"
tail -1 $O/${TAG}_step_stats.txt; cat $O/${TAG}_step_period.txt | head -2
