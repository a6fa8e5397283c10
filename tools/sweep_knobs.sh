#!/bin/bash
# round 6: one pass over the tuning setters around their defaults (bench.py --call), the default interleaved every few configurations
# as the drift reference; sustained ms per step.  usage (GPU box): bash tools/sweep_knobs.sh OUTDIR
export TMPDIR=/tmp
O=${1:-gpurun_out/r6_sweep}
mkdir -p $O
run() {
  label=$1; shift
  timeout 200 python bench.py --no-cpu-baseline --no-probes --no-b256 --no-config4 --steps 30 --warmup 8 --sustained-seconds 3 "$@" 2>/dev/null | grep '^{' > $O/$label.json
  python - $O/$label.json "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:>28s}: mean {d['ms_per_step']:.3f}  median {d['ms_per_step_median']:.3f}  sustained {d['sustained']['ms_per_step']:.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default_a
run rot0 --call cpc_set_dma_rotation=0
run rot3 --call cpc_set_dma_rotation=3
run rot7 --call cpc_set_dma_rotation=7
run rot11 --call cpc_set_dma_rotation=11
run nce_grid_m1 --call cpc_set_nce_grid=-1
run nce_grid_512 --call cpc_set_nce_grid=512
run default_b
run wgrad_stages2 --call cpc_set_wgrad_dma_stages=2
run wgrad_groups384 --call cpc_set_wgrad_dma_groups=384
run wgrad_groups512 --call cpc_set_wgrad_dma_groups=512
run dma_layer2 --call cpc_set_dma_layer2=1
run small_tile64 --call cpc_set_conv_small_tile=64
run dgrad_nsplit0 --call cpc_set_dgrad_nsplit=0
run default_c
run sched3 --call cpc_set_step_schedule=3,0
run sched0 --call cpc_set_step_schedule=0,0
run prep192 --call cpc_set_index_prep_groups=192
run prep384 --call cpc_set_index_prep_groups=384
run gru_wgrad_s0 --call cpc_set_gru_wgrad_stream=0
run conv0_g2 --call cpc_set_conv0_tuning=2,0
run conv0_g8 --call cpc_set_conv0_tuning=8,0
run conv0_nt --call cpc_set_conv0_tuning=4,1
run default_d
run pace_2_4 --call cpc_set_gru_poll_pacing=2,4
run pace_4_8 --call cpc_set_gru_poll_pacing=4,8
run pace_8_16 --call cpc_set_gru_poll_pacing=8,16
run poll_plain0 --call cpc_set_gru_poll_plain=0
run poll_plain5 --call cpc_set_gru_poll_plain=5
run xcd_local0 --call cpc_set_gru_xcd_local=0
run nce_rows_apart0 --call cpc_set_nce_rows_apart=0
run heads_dma0 --call cpc_set_nce_heads_dma=0
run default_e
