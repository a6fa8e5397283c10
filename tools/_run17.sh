#!/bin/bash
# round 6: timeline of one config-4 step
export TMPDIR=/tmp
O=gpurun_out/r6_c4tl
mkdir -p $O
rm -rf /tmp/p4; timeout 400 rocprofv3 --kernel-trace -d /tmp/p4 -o res -- python tools/run_config4.py 5 64 > $O/c4.log 2>&1
db=$(find /tmp/p4 -name "*.db" | head -1); python tools/step_timeline.py $db > $O/c4_timeline.txt 2>&1
wc -l $O/c4_timeline.txt
