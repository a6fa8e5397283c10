#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_dist.py tests/test_gpu_train_step.py tests/test_gpu_harness.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
ROUNDS=2 bash tools/ab_bench.sh $O/ab "open|" "closed|--no-pipeline-tail"
bash tools/trace_variant.sh $O open
timeout 500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"; tail -3 $O/bench_full.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5a/bench_full.json'))
for k in ('value','ms_per_step','ms_per_step_median','host_enqueue_ms_per_step','timed_region','in_step_us','time_dominant_kernel','b256_variant','bf16_storage_variant','sustained'):
    print(k, d.get(k))
print(d['roofline']); print(d['roofline_hbm_layer'])
PY
