#!/bin/bash
# PMC counters of conv0_fwd_kernel alone (tools/sweep_conv0.py, default tuning only): where its cycles go.
# One counter group per rocprofv3 pass, kernel-trace only (MI355X_MICROARCH.md).  Run on the GPU box from the repo root.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv0
mkdir -p $OUT
: > $OUT/pmc_conv0.csv
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS"; do
  d=/tmp/pmcc0_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $d -o res -- python tools/sweep_conv0.py 64 one > $OUT/run.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db conv0_fwd_kernel | tail -n +2 >> $OUT/pmc_conv0.csv; else echo "no db for $grp" >> $OUT/run.log; tail -3 $OUT/run.log; fi
done
cat $OUT/pmc_conv0.csv
