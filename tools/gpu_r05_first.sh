#!/bin/bash
# The first GPU call of the next round: everything round 4 wrote after its GPU minutes were spent, measured - and the artefacts
# of the last build whole (round 4's bench lines of C3 .. C5W and its PMC traffic are an earlier build's).
#   1. the GPU tier with the files that sort last (request road, effective policies)
#   2. tools/gpu_final_r04.sh (kernel statistics, PMC traffic - C3 with the tags as bytes -, bench lines, all on this build)
#   3. tools/gpu_requests_and_trail.py: the request road against the input road, the trail walk against cbh_check_batch
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
tools/gpu_final_r04.sh $TAG
for w in C2 C5; do
  timeout 300 python tools/gpu_requests_and_trail.py $w 250000 10 > $OUT/requests_and_trail_$w.txt 2>&1; tail -8 $OUT/requests_and_trail_$w.txt
done
