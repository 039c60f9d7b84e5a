#!/bin/bash
# GPU box: instruction counts of the decision kernel with parts of the walk switched off
# (library must be built with CBH_ABLATION=1).  Usage: gpu_pmc_ablate.sh [C2|C3]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-C2}
OUT=$R/gpurun_out/pmc_abl
rm -rf $OUT; mkdir -p $OUT
for f in 0 0x400 0x800 0x200; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH \
    --output-format csv -d $OUT/f$f -o f$f -- python $R/tools/gpu_ablate.py $WL $f > $OUT/f$f.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, os
for f in sorted(glob.glob('gpurun_out/pmc_abl/*/*counter_collection.csv')):
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'cbh_check_kernel' not in r.get('Kernel_Name',''): continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    w = sum(agg['SQ_WAVES'])/len(agg['SQ_WAVES'])
    print(os.path.basename(f).split('_')[0], "waves %d" % w, " ".join("%s %.0f" % (k.replace('SQ_INSTS_',''), sum(v)/len(v)/w) for k, v in sorted(agg.items()) if k != 'SQ_WAVES'))
PY
