"""Phase clocks of the sequential Hough stage (ppht_kernel) for ONE scan.  Needs the debug library
(`make -C omr-img-corrector_amd/csrc debug`).  Usage: python tools/hstamps.py [rows cols]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from oics import _lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "libomrdeskew_dbg.so")
from oics import omr, synth
L = C.CDLL(_l.LIB_PATH)
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3508, 2480)
g, th = synth.make_card(rows, cols, 2)
out = (C.c_ulonglong * 12)()
omr.get_result_from_edges_detection(g, 150.0, 50.0)  # warm-up
L.omr_debug_ppht_stamps(out, 1)
t0 = time.perf_counter()
r = omr.get_result_from_edges_detection(g, 150.0, 50.0)
dt = time.perf_counter() - t0
L.omr_debug_ppht_stamps(out, 1)
names = ["next point", "vote + arg-max", "walk pass 1", "pass 2 + un-vote", "re-test (registers)"]
tot = sum(out[i] for i in range(5))
print("%dx%d: call %.3f s, angle %.3f, status %d; ppht clocks total %.1f Mcycles" % (cols, rows, dt, r.angle, int(r.status), tot / 1e6))
for i, nm in enumerate(names):
    print("  %-18s %8.1f Mcycles %5.1f %%" % (nm, out[i] / 1e6, 100.0 * out[i] / max(1, tot)))
print("  served points %d, pass-1 rounds %d (%.2f per point), pass-2 rounds %d" % (out[5], out[6], out[6] / max(1, out[5]), out[7]))
print("  accepted segments %d, un-voted points %d, un-vote clocks %.1f Mcycles (part of pass 2)" % (out[9], out[8], out[10] / 1e6))
print("  cycles per served point %.0f" % (tot / max(1, out[5])))
print("  walks that end within 64 steps in both directions: %d" % out[11])
