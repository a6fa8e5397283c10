#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6_run10
mkdir -p $O
timeout 300 python tools/run_config4.py 10 64 2>&1 | tail -1
rm -rf /tmp/p4; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o res -- python tools/run_config4.py 5 64 > $O/c4.log 2>&1
db=$(find /tmp/p4 -name "*.db" | head -1)
python tools/rocpd_stats.py $db $O/r6_config4_kernel_stats_before.csv | head -40
