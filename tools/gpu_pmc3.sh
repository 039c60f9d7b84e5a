#!/bin/bash
# GPU box: instruction-mix + cache + HBM traffic counter passes for the decision kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
run sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python - <<'PY'
import csv, glob, os
for f in sorted(glob.glob('gpurun_out/pmc3/*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        if 'cbh_check_kernel' not in r.get('Kernel_Name',''): continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print(os.path.basename(f))
    for k, v in agg.items():
        print("   %-24s n=%d mean=%.6g" % (k, len(v), sum(v)/len(v)))
PY
