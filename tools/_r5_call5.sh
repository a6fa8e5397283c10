#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_full_configs.py tests/test_gpu_train_step.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log; grep -E "relu ties|product:|misround:" $O/pytest.log | head -60
for v in "nsplit|--call cpc_set_dgrad_nsplit=256" "small64|--call cpc_set_conv_small_tile=64" "base|"; do
  label=${v%%|*}; args=${v#*|}
  timeout 300 python bench.py --no-cpu-baseline --no-probes --steps 40 --warmup 10 --sustained-seconds 2 $args 2>/dev/null | grep '^{' > $O/loss_$label.json
  python -c "
import json; d=json.load(open('$O/loss_$label.json')); print('$label', d['config']['loss_mean_over_heads'], d['ms_per_step'], d['sustained']['ms_per_step'])"
done
