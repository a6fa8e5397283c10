#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_modules.py tests/test_gpu_full_configs.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
for v in "b256_nt2|" "b256_nt1|--call cpc_set_gru_tiles_per_wg=1" "b256_nt2b|" "b256_nt1b|--call cpc_set_gru_tiles_per_wg=1"; do
  label=${v%%|*}; args=${v#*|}
  timeout 300 python bench.py --batch 256 --no-cpu-baseline --no-probes --steps 20 --warmup 6 --sustained-seconds 2 $args 2>/dev/null | grep '^{' > $O/$label.json
  python -c "
import json; d=json.load(open('$O/$label.json')); print('$label', d['config']['loss_mean_over_heads'], d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])"
done
bash tools/trace_variant.sh $O b256 --batch 256
