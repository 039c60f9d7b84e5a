#!/bin/bash
# Instruction mix / wait counters of the WIRE kernels (rocprofv3 --pmc, separate passes, kernel trace only) on the device road of
# tools/e2e_wire_bench.cpp.    usage: gpu_pmc_mix_wire.sh TAG "C2 C5"
set -u
TAG=${1:-r05_pmc_wire}; WL=${2:-"C2 C5"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM")
for w in $WL; do
  python tools/export_wire.py $w 131072 /tmp/wire_$w > $OUT/export_$w.log 2>&1
  i=0
  for p in "${PASSES[@]}"; do
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $OUT/pmc_$w/p$i -o p$i -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 0.3 1 device_out > $OUT/pmc_${w}_p$i.log 2>&1 )
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
with open('$OUT/pmc_mix_wire_$w.txt', 'w') as out:
    for (n, c), v in sorted(vals.items()):
        m = sum(v) / len(v)
        line = '%-4s %-34s %-22s launches=%-4d mean=%-12.5g per_wave(%d)=%.1f' % ('$w', n, c, len(v), m, waves.get(n, 0), m / max(1.0, waves.get(n, 1.0)))
        out.write(line + '\n')
        if 'fill' in n or 'out_write' in n or 'out_size' in n: print(line)
P
  rm -rf $OUT/pmc_$w
done
