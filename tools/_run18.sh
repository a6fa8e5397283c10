#!/bin/bash
# round 6: the criterion's prediction product on the DMA-fed tile: parity, then in-step A/B
export TMPDIR=/tmp
O=gpurun_out/r6_heads
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_fused_step.py tests/test_gpu_modules.py -q -x 2>&1 | tail -3
ROUNDS=3 bash tools/ab_bench.sh $O/ab "generic|--no-b256 --no-config4 --call cpc_set_nce_heads_dma=0" "dma|--no-b256 --no-config4 --call cpc_set_nce_heads_dma=1" | tee $O/ab_step.txt
rm -rf /tmp/p5; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o res -- python bench.py --no-cpu-baseline --no-probes --no-b256 --no-config4 --steps 20 --warmup 5 --sustained-seconds 0 > $O/trace.log 2>&1
db=$(find /tmp/p5 -name "*.db" | head -1); python tools/step_timeline.py $db > $O/step_timeline.txt 2>&1
grep -n "nt_gemm\|gemm_nt_dma\|rows_to_h2\|nce_fwd_h2\|gemm_weight" $O/step_timeline.txt | cut -c1-150
