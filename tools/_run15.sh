#!/bin/bash
# round 6: A/B of the 128 x 128 wave tile (one wave per SIMD) for the DMA-fed plain NT products, config 4
export TMPDIR=/tmp
O=gpurun_out/r6_w128
mkdir -p $O
for r in 64 128; do
  rm -rf /tmp/p4; CPC_CALLS="cpc_set_dma_wave_rows=$r" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o res -- python tools/run_config4.py 5 64 > $O/c4_$r.log 2>&1
  db=$(find /tmp/p4 -name "*.db" | head -1); python tools/rocpd_stats.py $db $O/c4_stats_$r.csv > /dev/null
  grep "gemm_nt_dma" $O/c4_stats_$r.csv | sed "s/^/rows $r: /"
done
