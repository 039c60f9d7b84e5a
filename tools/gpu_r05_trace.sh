#!/bin/bash
# Timeline of one cbh_wire_check_pb call (copies and kernels, per stream): where a 2 ms call spends what its kernels do not.
set -u
TAG=${1:-r05t}; W=${2:-C2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace_$W -o t -- python $R/tools/gpu_wire_onecall.py $W 250000 > $OUT/trace_$W.log 2>&1 )
tail -2 $OUT/trace_$W.log
python tools/trace_timeline.py /tmp/trace_$W 2.6 > $OUT/timeline_$W.txt 2>&1; head -120 $OUT/timeline_$W.txt
