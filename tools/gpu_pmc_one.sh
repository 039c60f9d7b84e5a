#!/bin/bash
# dynamic instruction / cycle counters of the decision kernel of one workload (flat kernel unless CBH_NO_FLAT=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-C2}; NB=${2:-12}
OUT=$R/gpurun_out/pmc1_$W
mkdir -p $OUT
python $R/__graft_entry__.py > $OUT/build.log 2>&1
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/bench.py --workload $W --batches $NB --steps 3 --warmup 1 --no-cpu-baseline --no-side-legs > $OUT/$name.log 2>&1
}
run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES
run cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
cd $R
python - $OUT <<'PY'
import csv, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*counter_collection.csv')):
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'cbh_check' not in r.get('Kernel_Name', ''): continue
        agg.setdefault((r['Kernel_Name'].split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
    waves = None
    for (k, c), v in sorted(agg.items()):
        if c == 'SQ_WAVES': waves = sum(v) / len(v)
    for (k, c), v in sorted(agg.items()):
        m = sum(v) / len(v)
        print("   %-30s %-22s mean=%.5g%s" % (k, c, m, "  per wave %.1f" % (m / waves) if waves else ""))
PY
