#!/bin/bash
# Runs on the GPU box: PMC counter passes for the decision kernel (separate passes; kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
cd $R
python - <<'PY'
import csv, glob, os
for f in sorted(glob.glob('gpurun_out/pmc/*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        if 'cbh_check_kernel' not in r.get('Kernel_Name',''): continue
        k = r['Counter_Name']; agg.setdefault(k, []).append(float(r['Counter_Value']))
    print(os.path.basename(f))
    for k, v in agg.items():
        print("   %-24s n=%d mean=%.4g" % (k, len(v), sum(v)/len(v)))
PY
