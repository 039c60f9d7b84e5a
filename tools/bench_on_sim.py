"""Developer aid: bench.py's control flow against the simulator build of the library (torch.cuda patched away).  Figures are meaningless;
what it shows is that the script runs end to end and prints its line.
    CBH_BENCH_NO_DIST=1 python tools/bench_on_sim.py --requests 3000 --batches 2 --steps 2 --warmup 1 --cpu-sample 2000 [--audit-trail]
(the side legs keep sizes of their own - minutes on the simulator; --no-side-legs for a quick look)"""
import runpy, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import torch
from sim_engine import sim_engine
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *_a, **_k: None
torch.cuda.synchronize = lambda *_a, **_k: None
torch.cuda.device_count = lambda: 1
_tensor, _empty = torch.tensor, torch.empty
torch.tensor = lambda *a, **k: _tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
torch.empty = lambda *a, **k: _empty(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
torch.Tensor.pin_memory = lambda self: self
with sim_engine():
    sys.argv = ["bench.py"] + sys.argv[1:]
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
