#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6_run11
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_transformer.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8
for m in 1 0 1 0; do echo "gemm_dma=$m"; CPC_GEMM_DMA=$m timeout 300 python tools/run_config4.py 10 64 2>&1 | tail -1; done
rm -rf /tmp/p4; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o res -- python tools/run_config4.py 5 64 > $O/c4.log 2>&1
db=$(find /tmp/p4 -name "*.db" | head -1)
python tools/rocpd_stats.py $db $O/r6_config4_kernel_stats.csv | head -32
