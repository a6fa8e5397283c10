#!/bin/bash
# usage (GPU box): tools/ab_bench.sh OUTDIR "label1|bench args" "label2|bench args" ...   -- alternates the variants ROUNDS times
# (default 3) in fresh processes on the same box and prints ms/step (mean / median / sustained) per run.
out=$1; shift
mkdir -p "$out"
rounds=${ROUNDS:-3}
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%|*}; args=${spec#*|}
    env $(echo "$args" | grep -o "ENV:[^ ]*" | sed "s/ENV://") timeout 300 python bench.py --no-cpu-baseline --no-probes --steps 40 --warmup 10 --sustained-seconds 3 $(echo "$args" | sed "s/ENV:[^ ]*//g") 2>/dev/null | grep '^{' > "$out/${label}_$r.json"
    python - "$out/${label}_$r.json" "$label" "$r" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:>14s} run {sys.argv[3]}: mean {d['ms_per_step']:.3f}  median {d['ms_per_step_median']:.3f}  sustained {d['sustained']['ms_per_step']:.3f}  host {d['host_enqueue_ms_per_step']:.3f}  loss {d['config']['loss_mean_over_heads']}")
except Exception as e:
    print(sys.argv[2], "run", sys.argv[3], "FAILED", e)
PY
  done
done
