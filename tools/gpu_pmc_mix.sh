#!/bin/bash
# instruction / stall counters of the decision kernel of the given workloads (two counter passes each)
#   bash tools/gpu_pmc_mix.sh <tag> W[:batches] ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/__graft_entry__.py > $OUT/build.log 2>&1
for spec in "$@"; do
  W=${spec%%:*}; NB=${spec##*:}; [ "$NB" = "$W" ] && NB=2
  run() { name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${W}_$name -o $name -- python $R/bench.py --workload $W --batches $NB --steps 3 --warmup 1 --no-cpu-baseline --no-side-legs > $OUT/${W}_$name.log 2>&1
  }
  run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES
  run cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
done
cd $R
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*/*counter_collection.csv')):
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'cbh_check' not in r.get('Kernel_Name', ''): continue
        agg.setdefault((r['Kernel_Name'].split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
    print(f[len(sys.argv[1]) + 1:])
    waves = {k: sum(v) / len(v) for (k, c), v in agg.items() if c == 'SQ_WAVES'}
    for (k, c), v in sorted(agg.items()):
        print("   %-34s %-22s n=%d mean=%.5g" % (k, c, len(v), sum(v) / len(v)))
PY
find $OUT -name '*.csv' -size +200k -delete
