#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
}
run lvl SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES
run sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL
run cyc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
cd $R
python - <<'PY'
import csv, glob, os
for f in sorted(glob.glob('gpurun_out/pmc2/*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        if 'cbh_check_kernel' not in r.get('Kernel_Name',''): continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print(os.path.basename(f))
    for k, v in agg.items():
        print("   %-24s n=%d mean=%.4g" % (k, len(v), sum(v)/len(v)))
PY
