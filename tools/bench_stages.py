"""Stage benchmark (rows f1 / f2 of SURVEY.md section 8): front-end and final-warp kernels on
device-resident A4 scans, HIP-event timed, reported as GB/s of compulsory traffic against the
HBM roofline.  Usage: python tools/bench_stages.py [reps]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import synth
from oics._lib import check, lib, u8p

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ROWS, COLS = 3508, 2480
dev = torch.device("cuda:0")
L = lib()
g, th = synth.make_card(ROWS, COLS, 2)
d_gray = torch.from_numpy(g).to(dev)
d_rgb = torch.from_numpy(np.repeat(g[:, :, None], 3, axis=2).copy()).to(dev)
white = np.array([255, 255, 255, 0], np.uint8)
out = {}


def timeit(name, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    out[name] = {"ms": ms, "bytes": nbytes, "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000}
    print("%-34s %8.3f ms  %8.1f GB/s  (%.3f of 8 TB/s)" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000))


N = ROWS * COLS
d_a = torch.empty((ROWS, COLS), dtype=torch.uint8, device=dev)
d_b = torch.empty((ROWS, COLS), dtype=torch.uint8, device=dev)
# resize needs divisible sizes: crop to 3505 x 2480 (k = 5)
R5, C5 = 3505, 2480
d_small = torch.empty((R5 // 5, C5 // 5), dtype=torch.uint8, device=dev)
timeit("rgb2gray (3ch -> 1ch)", lambda: check(L.omr_rgb_to_gray_device(d_rgb.data_ptr(), COLS * 3, ROWS, COLS, 3, d_a.data_ptr(), COLS, None)), 4 * N)
timeit("erode 3x3 cross x3 (fused)", lambda: check(L.omr_erode3_device(d_gray.data_ptr(), COLS, ROWS, COLS, d_b.data_ptr(), COLS, None)), 2 * N)
timeit("resize INTER_AREA /5", lambda: check(L.omr_resize_area_device(d_gray.data_ptr(), COLS, R5, C5, 1, d_small.data_ptr(), C5 // 5, R5 // 5, C5 // 5, None)), R5 * C5 + R5 * C5 // 25)
timeit("threshold(127,255)", lambda: check(L.omr_threshold_binary_device(d_gray.data_ptr(), COLS, ROWS, COLS, d_a.data_ptr(), COLS, None)), 2 * N)
for interp, nm in ((0, "NEAREST"), (1, "LINEAR")):
    for clip, cn in ((0, "DEFAULT"), (1, "CONTAIN")):
        dr, dc = C.c_int32(), C.c_int32()
        check(L.omr_rotate_size(ROWS, COLS, -th, clip, C.byref(dr), C.byref(dc)))
        d_out = torch.empty((dr.value, dc.value), dtype=torch.uint8, device=dev)
        timeit("deskew warp %s %s gray" % (nm, cn),
               lambda: check(L.omr_rotate_device(d_gray.data_ptr(), COLS, ROWS, COLS, 1, -th, 1.0, interp, white.ctypes.data_as(u8p), clip, d_out.data_ptr(), dc.value, dr.value, dc.value, None)),
               N + dr.value * dc.value)
print(json.dumps(out))
