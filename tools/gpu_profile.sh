#!/bin/bash
# Runs on the GPU box (via gpurun): the judged artefacts of one round.
#   1. python bench.py (default flags, with the CPU baseline)        -> gpurun_out/bench_full.json
#   2. rocprofv3 --kernel-trace --stats of the same command          -> gpurun_out/kernel_stats.txt
#   3. rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, one pass each)
#                                                                    -> gpurun_out/pmc_traffic.json
# Copy the three files into profiles/ (tracked) afterwards.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
(timeout 600 python bench.py 2>$OUT/bench_full.err | grep '^{' | tail -1) > $OUT/bench_full.json

cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof $OUT/pmc_t
BENCH="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- $BENCH > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py "$DB" $OUT/kernel_stats.txt > /dev/null
mkdir -p $OUT/pmc_t
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_t/$c -o $c -- $BENCH > $OUT/pmc_t/$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json
vals, kernel = {}, None
for f in glob.glob('gpurun_out/pmc_t/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'cbh_check_kernel' not in r.get('Kernel_Name', ''):
            continue
        kernel = r['Kernel_Name'].split('(')[0]
        vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
if vals:
    fetch_kb = sum(vals['FETCH_SIZE']) / len(vals['FETCH_SIZE'])
    write_kb = sum(vals['WRITE_SIZE']) / len(vals['WRITE_SIZE'])
    out = {"kernel": kernel, "fetch_kb_per_launch": fetch_kb, "write_kb_per_launch": write_kb,
           "bytes_per_launch": (fetch_kb + write_kb) * 1024.0, "launches": len(vals['FETCH_SIZE']),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), separate passes over `bench.py --steps 100 --warmup 10`; "
                   "unscaled: the kernel reads with dword / 8-byte loads, not the 16 B/lane streaming reads the "
                   "MI355X guide's x2 correction was calibrated on (see DESIGN.md, Measurement)"}
    json.dump(out, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
    print(json.dumps(out))
PY
cat $OUT/kernel_stats.txt
cat $OUT/bench_full.json
