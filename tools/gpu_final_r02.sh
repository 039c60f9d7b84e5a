#!/bin/bash
# Runs on the GPU box (via gpurun): the judged round-2 artefacts, in an order that lets the bench lines carry the PMC
# traffic measured in THIS call.
#   1. per workload C2..C5: rocprofv3 --kernel-trace --stats of `bench.py --workload W` (no side legs)   -> kernel_stats_W.txt
#      and rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one pass each                         -> pmc_W/
#   2. counter calibration on known byte counts (tools/pmc_calib.hip)                                    -> pmc_calibration.json
#   3. tools/pmc_summary.py -> pmc_traffic.json, copied to profiles/r02_pmc_traffic.json ON THE BOX (what bench.py reads)
#   4. python bench.py (driver flags: --steps 20 --warmup 3, CPU baseline + side legs)                   -> bench_C2.json
#      python bench.py --workload C3|C4|C5 --steps 10 --warmup 2                                        -> bench_W.json
#   5. per-wave phase profile of the flat kernel from a prebuilt profiling library (build_variants/lib_profile.so)
#   6. the wire-inclusive end-to-end rate (tools/e2e_bench.py)                                           -> e2e_wire_inclusive.json
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r02h}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
for w in C2 C3 C4 C5; do
  NB=$(case $w in C2) echo 12;; C3) echo 3;; C4) echo 6;; C5) echo 8;; esac)
  BENCH="python $R/bench.py --workload $w --batches $NB --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- $BENCH > $OUT/prof_$w.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$w.txt | head -3
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$w/$c -o $c -- $BENCH > $OUT/pmc_${w}_$c.log 2>&1 )
  done
  rm -rf $OUT/prof_$w
done
hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib > $OUT/calib_build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_calib/$c -o $c -- /tmp/pmc_calib > $OUT/pmc_calib_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT > /dev/null
cp $OUT/pmc_traffic.json profiles/r02_pmc_traffic.json
(timeout 900 python bench.py --steps 20 --warmup 3 2>$OUT/bench_C2.err | grep '^{' | tail -1) > $OUT/bench_C2.json
cut -c1-330 $OUT/bench_C2.json; echo
for w in C3 C4 C5; do
  (timeout 600 python bench.py --workload $w --steps 10 --warmup 2 2>$OUT/bench_$w.err | grep '^{' | tail -1) > $OUT/bench_$w.json
done
if [ -f build_variants/lib_profile.so ]; then
  cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
  cp build_variants/lib_profile.so cerbos_amd/libcerbos_hip.so
  for w in C2 C3 C4; do timeout 250 python tools/gpu_cycles_flat.py $w > $OUT/cycles_flat_$w.txt 2>&1; done
  cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
fi
timeout 300 python tools/e2e_bench.py C2 131072 8192 3 1,8,32,64 > $OUT/e2e_wire_inclusive.json 2> $OUT/e2e.err
tail -4 $OUT/e2e_wire_inclusive.json
find $OUT -name '*.csv' -size +300k -delete
