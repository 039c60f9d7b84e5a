#!/bin/bash
# Round 6: tools/pmc_shapes.hip under the profiler - FETCH_SIZE per launch and kernel durations of the shapes the column cache could be filled in.
set -u
TAG=${1:-r06shapes}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/pmc_shapes.hip -o /tmp/pmc_shapes > $OUT/build.log 2>&1
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o f -- /tmp/pmc_shapes > $OUT/pmc.log 2>&1 )
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- /tmp/pmc_shapes > $OUT/stats.log 2>&1 )
python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob('$OUT/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        v[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k, x in sorted(v.items()):
    print('%-24s FETCH_SIZE kb per launch' % k, [round(y) for y in x])
for f in glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print('%-24s calls %s avg ns %s' % (r.get('Name', '?').split('(')[0], r.get('Calls'), r.get('AverageNs', r.get('Average'))))
PY
find $OUT -name '*.csv' -size +300k -delete
