#!/bin/bash
# usage (GPU box): tools/trace_variant.sh OUTDIR TAG [bench args...]  -- rocprofv3 kernel trace of 20 eager steps; writes the
# last step's timeline, its per-kernel totals and the per-kernel averages over all traced steps
out=$1; tag=$2; shift 2
mkdir -p $out
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o res -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes --launch eager --sustained-seconds 0 "$@" > $out/${tag}_bench.json 2> $out/${tag}_trace.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv > /dev/null
python tools/step_timeline.py $db --stats > $out/${tag}_step_stats.txt
python tools/step_timeline.py $db > $out/${tag}_step_timeline.txt
python tools/step_timeline.py $db --period > $out/${tag}_step_period.txt
tail -1 $out/${tag}_step_stats.txt; head -1 $out/${tag}_step_period.txt
