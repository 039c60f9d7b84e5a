#!/bin/bash
# Instruction mix / wait counters of the decision kernels (rocprofv3 --pmc, separate passes, kernel trace only).
#   usage: gpu_pmc_mix.sh TAG "T C4" [ENV=VALUE]
set -u
TAG=${1:-r04_pmc_mix}; WL=${2:-"T"}; SW=${3:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
[ -n "$SW" ] && export "$SW"
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU")
for w in $WL; do
  BENCH="python $R/bench.py --workload $w --batches 4 --steps 3 --warmup 1 --no-cpu-baseline --no-side-legs"
  i=0
  for p in "${PASSES[@]}"; do
    ( cd /tmp && CBH_BENCH_NO_DIST=1 timeout 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $OUT/pmc_$w/p$i -o p$i -- $BENCH > $OUT/pmc_${w}_p$i.log 2>&1 )
    i=$((i+1))
  done
  python - <<P
import csv, glob, collections
vals = collections.defaultdict(list)
for f in glob.glob('$OUT/pmc_$w/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '').split('(')[0]
        if n.startswith('cbh_'):
            vals[(n, r['Counter_Name'])].append(float(r['Counter_Value']))
waves = {k[0]: sum(v) / len(v) for k, v in vals.items() if k[1] == 'SQ_WAVES'}
with open('$OUT/pmc_mix_$w.txt', 'w') as out:
    for (n, c), v in sorted(vals.items()):
        m = sum(v) / len(v)
        line = '%-4s %-34s %-22s launches=%-4d mean=%-12.5g per_wave(%d)=%.1f' % ('$w', n, c, len(v), m, waves.get(n, 0), m / max(1.0, waves.get(n, 1.0)))
        print(line); out.write(line + '\n')
P
  rm -rf $OUT/pmc_$w
done
