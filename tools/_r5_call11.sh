#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5k/bench.json'))
print(d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['in_step_us'], d['time_dominant_kernel']['us_in_step'], d['b256_variant'].get('ms_per_step'))
PY
