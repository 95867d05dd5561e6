"""Phase shares of runs_kernel in batch mode (8 scans per launch).  Needs the DEBUG library
(`make -C omr-img-corrector_amd/csrc debug`, -DOMR_RUNS_DEBUG): the release library has no stamps.
This script points the ctypes loader at lib/libomrdeskew_dbg.so explicitly."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from oics import _lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "libomrdeskew_dbg.so")  # before the first lib() call
from oics import projection, synth
L = C.CDLL(_l.LIB_PATH)
ROWS, COLS, G, REPS = 3508, 2480, 8, 6
cards = [synth.make_card(ROWS, COLS, 3 + i)[0] for i in range(G)]
buf = torch.empty((G * REPS, ROWS, COLS), dtype=torch.uint8, device="cuda:0")
for i in range(G * REPS):
    buf[i] = torch.from_numpy(cards[i % G]).to("cuda:0")
best = torch.zeros(G * REPS, dtype=torch.int32, device="cuda:0")
b = projection.Batch(ROWS, COLS, 10, 0.05, n_streams=1)
b.set_group(G)
b.set_timing(True)
out = (C.c_ulonglong * 8)()
b.run_device(buf.data_ptr(), ROWS * COLS, COLS, G * REPS, 127, best.data_ptr()); b.sync(); b.kernel_ms()
L.omr_debug_runs_stamps(out, 1)
b.run_device(buf.data_ptr(), ROWS * COLS, COLS, G * REPS, 127, best.data_ptr()); b.sync()
ms, k = b.kernel_ms()
L.omr_debug_runs_stamps(out, 1)
n = out[6]
names = ["prologue", "wait for the window", "compute", "issue of the next window", "flush (park, barriers, reduce)", "total"]
tot = out[5]
print("sweep stage %.3f ms per launch of %d scans (debug build, stamps on)" % (ms / k, G))
for i, nm in enumerate(names):
    print("%-16s %9.0f cycles/block  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
print("blocks", n)
