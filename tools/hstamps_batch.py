"""Phase clocks of the sequential Hough stage (ppht_kernel) of scan 0 WHILE a whole batch is in flight: which phase of the
dependent chain stretches when 256 scans share the memory system.  Needs the debug library
(`make -C omr-img-corrector_amd/csrc debug`).  Usage: python tools/hstamps_batch.py [batch sizes ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from oics import _lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "libomrdeskew_dbg.so")
from oics import omr, synth
L = C.CDLL(_l.LIB_PATH)
ROWS, COLS = 3508, 2480
sizes = [int(a) for a in sys.argv[1:]] or [1, 64, 256]
cards = [synth.make_card(ROWS, COLS, 2 + i)[0] for i in range(8)]
dev = torch.device("cuda:0")
names = ["next point", "vote + arg-max", "walk pass 1", "pass 2 + un-vote", "re-test (registers)"]
out = (C.c_ulonglong * 12)()
for B in sizes:
    d = torch.from_numpy(np.stack([cards[i % 8] for i in range(B)])).to(dev)
    omr.edges_detection_batch_device(d.data_ptr(), min(B, 8), ROWS * COLS, ROWS, COLS, 1, COLS, 150.0, 50.0)  # warm-up
    L.omr_debug_ppht_stamps(out, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    omr.edges_detection_batch_device(d.data_ptr(), B, ROWS * COLS, ROWS, COLS, 1, COLS, 150.0, 50.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.omr_debug_ppht_stamps(out, 1)
    tot = sum(out[i] for i in range(5))
    print("batch %4d: %.3f s (%.0f scans/s); scan 0: %.1f Mcycles, %.0f cycles per served point (%d points)" % (
        B, dt, B / dt, tot / 1e6, tot / max(1, out[5]), out[5]))
    print("   " + "  ".join("%s %.0f" % (nm, out[i] / max(1, out[5])) for i, nm in enumerate(names)) + "  [cycles per served point]; un-vote part %.0f" % (out[10] / max(1, out[5])))
    del d
    torch.cuda.empty_cache()
