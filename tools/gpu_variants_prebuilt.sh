#!/bin/bash
# kernel lab: time prebuilt library variants (build_variants/lib_<name>.so, built off-box) on the resident benches
#   bash tools/gpu_variants_prebuilt.sh <tag> "<workloads>" <names...>
export TMPDIR=/tmp
TAG=$1; WLS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
python __graft_entry__.py > $OUT/build.log 2>&1
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp build_variants/lib_$v.so cerbos_amd/libcerbos_hip.so
  line="variant $v:"
  for w in $WLS; do
    NB=$(case $w in C2) echo 12;; C3) echo 2;; C4) echo 3;; *) echo 4;; esac)
    r=$(timeout 200 python bench.py --workload $w --batches $NB --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs 2>$OUT/err_$v_$w.log | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('%.4f' % d['roofline']['kernel_ms'])" 2>/dev/null)
    line="$line  $w ${r:-FAIL}"
  done
  echo "$line" | tee -a $OUT/summary.txt
done
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
