#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6_dgrad_split
mkdir -p $O
for rot in 5 1005 2005; do
rm -rf /tmp/p5; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o res -- python bench.py --no-cpu-baseline --no-probes --no-b256 --no-config4 --steps 20 --warmup 5 --sustained-seconds 0 --call cpc_set_dma_rotation=$rot > $O/trace_$rot.log 2>&1
db=$(find /tmp/p5 -name "*.db" | head -1); python tools/rocpd_stats.py $db $O/stats_$rot.csv > /dev/null
grep "conv_dgrad_dma_kernel\|conv_fwd_dma_kernel" $O/stats_$rot.csv | sed "s/^/rot $rot: /"
done
