"""Phase shares of runs_kernel.  Needs the DEBUG library (`make -C omr-img-corrector_amd/csrc debug`,
-DOMR_RUNS_DEBUG): the release library has neither the stamp variant nor any environment switch.
This script points the ctypes loader at lib/libomrdeskew_dbg.so explicitly."""
import ctypes as C, os, sys
os.environ["OMR_RUNS_DBG"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from oics import _lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "libomrdeskew_dbg.so")  # before the first lib() call
from oics import projection, synth
L = C.CDLL(_l.LIB_PATH)
ROWS, COLS = 3508, 2480
g, th = synth.make_card(ROWS, COLS, 2)
d = torch.from_numpy(g).to("cuda:0")
plan = projection.SweepPlan(ROWS, COLS, 10, 0.05)
vs = torch.zeros(400, dtype=torch.float64, device="cuda:0"); hs = torch.zeros_like(vs)
out = (C.c_ulonglong * 8)()
L.omr_debug_runs_stamps(out, 1)
for _ in range(5):
    plan.run_device(d.data_ptr(), COLS, 127, None, None, None, vs.data_ptr(), hs.data_ptr(), None)
L.omr_debug_runs_stamps(out, 1)
n = out[6]
names = ["tables+setup", "commit+barriers", "prefetch issue", "compute", "flush", "total"]
tot = out[5]
for i, nm in enumerate(names):
    print("%-16s %8.0f cycles/block  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
print("blocks", n)
