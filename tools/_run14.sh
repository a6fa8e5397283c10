#!/bin/bash
# round-6 final evidence: the whole -m gpu suite, the step profiles (kernel stats, timeline, bf16), PMC, config-4 stats, the bench line
export TMPDIR=/tmp
O=gpurun_out/r6_final6
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.log | tail -4
bash tools/collect_profiles.sh r6 > $O/collect.log 2>&1; tail -6 $O/collect.log
bash tools/pmc_step.sh r6 > $O/pmc.log 2>&1; cp profiles/r6_pmc_step_* $O/
rm -rf /tmp/p4; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o res -- python tools/run_config4.py 5 64 > $O/c4.log 2>&1
db=$(find /tmp/p4 -name "*.db" | head -1); python tools/rocpd_stats.py $db $O/r6_config4_kernel_stats.csv > /dev/null
timeout 200 python tools/run_config4.py 10 64 2>&1 | tail -1
