#!/bin/bash
# round 5: everything the committed profiles/r5_* files come from, in one call on the GPU box
export TMPDIR=/tmp
O=gpurun_out/profiles_r5
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
bash tools/collect_profiles.sh r5
bash tools/pmc_step.sh r5 > $O/pmc_step.log 2>&1; tail -25 $O/pmc_step.log
rm -rf /tmp/prof_c4
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o res -- python tools/run_config4.py 5 64 > $O/config4_run.txt 2>> $O/trace.err
db=$(find /tmp/prof_c4 -name "*.db" | head -1)
python tools/rocpd_stats.py $db $O/r5_config4_kernel_stats.csv > /dev/null
cat $O/config4_run.txt
timeout 300 python tools/run_config4.py 10 64 >> $O/config4_run.txt 2>&1; tail -1 $O/config4_run.txt
