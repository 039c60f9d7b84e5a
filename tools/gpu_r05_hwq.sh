cd $GRAFT_REPO_ROOT
python __graft_entry__.py > /dev/null 2>&1
for q in 4 8 16; do for s in 4 8; do
  GPU_MAX_HW_QUEUES=$q CBH_WIRE_SLICES=$s python tools/gpu_wire_onecall.py C2 250000 2>&1 | tail -1 | sed "s/^/hwq=$q /"
done; done
GPU_MAX_HW_QUEUES=8 CBH_WIRE_SLICES=4 python tools/gpu_wire_onecall.py C5 250000 2>&1 | tail -1 | sed "s/^/hwq=8 /"
CBH_WIRE_SLICES=4 python tools/gpu_wire_onecall.py C5 250000 2>&1 | tail -1 | sed "s/^/hwq=default /"
