#!/bin/bash
# resident kernel time against occupancy: the same bench with extra dynamic LDS per workgroup (CBH_LDS_PAD)
#   bash tools/gpu_occupancy_sweep.sh <tag> "<workloads>" <pad bytes...>
export TMPDIR=/tmp
TAG=$1; WLS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
python __graft_entry__.py > $OUT/build.log 2>&1
for pad in "$@"; do
  line="pad $pad:"
  for w in $WLS; do
    NB=$(case $w in C2) echo 12;; C3) echo 2;; C4) echo 3;; *) echo 4;; esac)
    r=$(CBH_LDS_PAD=$pad timeout 200 python bench.py --workload $w --batches $NB --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs 2>$OUT/err_$w.log | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('%.4f' % d['roofline']['kernel_ms'])" 2>/dev/null)
    line="$line  $w ${r:-FAIL}"
  done
  echo "$line" | tee -a $OUT/summary.txt
done
