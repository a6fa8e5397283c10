#!/bin/bash
# round 6: FFN dropout on 16-bit Philox fields + faster weight re-layout kernels: parity, config-4 step and kernel stats
export TMPDIR=/tmp
O=gpurun_out/r6_philox16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_transformer.py tests/test_gpu_full_configs.py -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for i in 1 2 3; do timeout 200 python tools/run_config4.py 20 64 2>&1 | tail -1; done | tee $O/steps.txt
rm -rf /tmp/p4; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o res -- python tools/run_config4.py 5 64 > $O/c4.log 2>&1
db=$(find /tmp/p4 -name "*.db" | head -1); python tools/rocpd_stats.py $db $O/c4_stats.csv > /dev/null
grep "gemm_nt_dma_kernel<1\|col_l1\|gemm_weight_h2" $O/c4_stats.csv
