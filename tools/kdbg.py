"""Run the sweep kernel of the DEBUG library (make debug) a few times at C2 with whatever OMR_RUNS_DBG the
environment sets (bit 1: no compute, 2: no window fetch/commit, 4: no column flush; results are then wrong --
this is for attributing counters / time to phases under rocprofv3).  Usage: OMR_RUNS_DBG=2 python tools/kdbg.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from oics import _lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "libomrdeskew_dbg.so")
from oics import projection, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ROWS, COLS = 3508, 2480
g, th = synth.make_card(ROWS, COLS, 2)
d = torch.from_numpy(g).to("cuda:0")
dbg = os.environ.pop("OMR_RUNS_DBG", "0")
plan = projection.SweepPlan(ROWS, COLS, 10, 0.05)   # plan creation (dry run) with all stages on
os.environ["OMR_RUNS_DBG"] = dbg
plan.set_timing(True)
vs = torch.zeros(400, dtype=torch.float64, device="cuda:0"); hs = torch.zeros_like(vs)
ts = []
for _ in range(reps):
    plan.run_device(d.data_ptr(), COLS, 127, None, None, None, vs.data_ptr(), hs.data_ptr(), None)
    ts.append(plan.last_kernel_ms())
print("OMR_RUNS_DBG=%s sweep kernel ms:" % dbg, " ".join("%.3f" % t for t in ts))
