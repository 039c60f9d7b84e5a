cd $GRAFT_REPO_ROOT
python __graft_entry__.py > /dev/null 2>&1
W=${1:-C2}
for q in 4 8 16 24; do
GPU_MAX_HW_QUEUES=$q CBH_TRACE=1 python tools/gpu_wire_onecall.py $W 250000 2>&1 | tail -5 | sed "s/^/hwq=$q /" | cut -c1-400
done
