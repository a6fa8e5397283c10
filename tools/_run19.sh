#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6_suite2
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.log | tail -6
