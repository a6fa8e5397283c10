#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6_run8
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
bash tools/collect_profiles.sh r6 > $O/collect.log 2>&1; tail -8 $O/collect.log
bash tools/pmc_step.sh r6 > $O/pmc.log 2>&1; tail -30 $O/pmc.log; cp profiles/r6_pmc_step_* gpurun_out/r6_run8/
