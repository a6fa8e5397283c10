#!/bin/bash
# PMC counters of named kernels under a given command: where their cycles go.  One counter group per rocprofv3 pass, kernel-trace
# only (MI355X_MICROARCH.md).  usage (GPU box, repo root): bash tools/pmc_kernels.sh <out-name> "<command>" kernel-substring...
export TMPDIR=/tmp
NAME=$1; CMD=$2; shift 2
OUT=gpurun_out/pmc_$NAME
mkdir -p $OUT
: > $OUT/pmc.csv
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmck_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $d -o res -- $CMD > $OUT/run.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db "$@" | tail -n +2 >> $OUT/pmc.csv; else echo "no db for $grp" >> $OUT/run.log; tail -3 $OUT/run.log; fi
done
cat $OUT/pmc.csv
