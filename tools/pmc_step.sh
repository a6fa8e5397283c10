#!/bin/bash
# PMC counters of EVERY kernel of the train step (run on the GPU box, from the repo root):  bash tools/pmc_step.sh
# One counter group per rocprofv3 pass over `bench.py --steps 3` (kernel-trace only, see tools/pmc_collect.sh); per-kernel means
# over all launches of the run go to gpurun_out/pmc_step/pmc_step_counters.csv and, as HBM-side bytes per launch
# ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction) + matrix-pipe busy fraction, to pmc_step_summary.txt.
export TMPDIR=/tmp
TAG=${1:-r3}
OUT=gpurun_out/pmc_step
mkdir -p $OUT
: > $OUT/pmc_step_counters.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  d=/tmp/pmcs_$(echo $grp | tr ' ' '_')
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $d -o res -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-probes --launch eager --sustained-seconds 0 > $OUT/run.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db cpc:: | tail -n +2 >> $OUT/pmc_step_counters.csv; else echo "no db for $grp" >> $OUT/run.log; fi
done
python - <<'PY'
import csv, collections
rows = [r for r in csv.reader(open("gpurun_out/pmc_step/pmc_step_counters.csv")) if len(r) == 4]
k = collections.defaultdict(dict)
for name, c, n, v in rows:
    k[name][c] = float(v); k[name]["n"] = int(n)
out = []
for name, c in k.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        mb = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6
        busy = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0) if c.get("GRBM_GUI_ACTIVE") else float("nan")
        out.append((mb, busy, c["n"], name))
with open("gpurun_out/pmc_step/pmc_step_summary.txt", "w") as f:
    f.write("HBM-side MB per launch ((2*FETCH_SIZE + WRITE_SIZE) KB), matrix-pipe busy fraction, launches in the run, kernel\n")
    for mb, busy, n, name in sorted(out, reverse=True):
        f.write(f"{mb:10.1f} MB  mfma_busy {busy:5.2f}  n={n:3d}  {name[:110]}\n")
print(open("gpurun_out/pmc_step/pmc_step_summary.txt").read()[:3500])
PY
cp $OUT/pmc_step_summary.txt profiles/${TAG}_pmc_step_summary.txt; cp $OUT/pmc_step_counters.csv profiles/${TAG}_pmc_step_counters.csv
