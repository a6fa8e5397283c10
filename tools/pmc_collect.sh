#!/bin/bash
# PMC counters of the roofline kernels (run on the GPU box, from the repo root):  bash tools/pmc_collect.sh [B=64]
# One counter group per rocprofv3 pass, kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; no --pmc together with the sys/hip/hsa trace domains).  Writes the per-kernel means
# to gpurun_out/pmc/pmc_counters.csv and the HBM traffic summary bench.py reads to profiles/pmc_traffic.json.
B=${1:-64}
TAG=${2:-r3}
export TMPDIR=/tmp
OUT=gpurun_out/pmc
mkdir -p $OUT
: > $OUT/pmc_counters.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $d -o res -- python tools/probe_kernels.py $B > $OUT/run.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db conv_fwd_dma_kernel conv0_fwd_kernel nce_fwd_kernel | tail -n +2 >> $OUT/pmc_counters.csv; else echo "no db for $grp" >> $OUT/run.log; fi
done
python tools/pmc_to_json.py $OUT/pmc_counters.csv $B profiles/pmc_traffic.json
cp $OUT/pmc_counters.csv profiles/${TAG}_pmc_counters.csv
cat profiles/pmc_traffic.json
