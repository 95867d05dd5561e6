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
# the production final warp runs on the 3-channel (BGR) scan: omr.rs:408-445 (NEAREST, CONTAIN), core/src/main.rs:72-81 (LINEAR, CONTAIN)
for interp, nm in ((0, "NEAREST"), (1, "LINEAR")):
    dr, dc = C.c_int32(), C.c_int32()
    check(L.omr_rotate_size(ROWS, COLS, -th, 1, C.byref(dr), C.byref(dc)))
    d_out3 = torch.empty((dr.value, dc.value, 3), dtype=torch.uint8, device=dev)
    timeit("deskew warp %s CONTAIN BGR" % nm,
           lambda: check(L.omr_rotate_device(d_rgb.data_ptr(), COLS * 3, ROWS, COLS, 3, -th, 1.0, interp, white.ctypes.data_as(u8p), 1, d_out3.data_ptr(), dc.value * 3, dr.value, dc.value, None)),
           3 * (N + dr.value * dc.value))
# the same kernels on nine scans stacked into one 2480 x 31572 image (launch ramp and tail amortised: the
# per-byte rate a batch pipeline sees)
T = 9
d_tall = d_gray.repeat(T, 1).contiguous()
d_tall_o = torch.empty_like(d_tall)
TR = ROWS * T
timeit("[x9 tall] erode 3x3 cross x3", lambda: check(L.omr_erode3_device(d_tall.data_ptr(), COLS, TR, COLS, d_tall_o.data_ptr(), COLS, None)), 2 * N * T)
timeit("[x9 tall] threshold(127,255)", lambda: check(L.omr_threshold_binary_device(d_tall.data_ptr(), COLS, TR, COLS, d_tall_o.data_ptr(), COLS, None)), 2 * N * T)
TR5 = TR - TR % 5
d_tall_s = torch.empty((TR5 // 5, C5 // 5), dtype=torch.uint8, device=dev)
timeit("[x9 tall] resize INTER_AREA /5", lambda: check(L.omr_resize_area_device(d_tall.data_ptr(), COLS, TR5, C5, 1, d_tall_s.data_ptr(), C5 // 5, TR5 // 5, C5 // 5, None)), TR5 * C5 + TR5 * C5 // 25)
dr, dc = C.c_int32(), C.c_int32()
check(L.omr_rotate_size(TR, COLS, -1.0, 0, C.byref(dr), C.byref(dc)))
for interp, nm in ((0, "NEAREST"), (1, "LINEAR")):
    timeit("[x9 tall] warp %s DEFAULT gray (1 deg)" % nm,
           lambda: check(L.omr_rotate_device(d_tall.data_ptr(), COLS, TR, COLS, 1, -1.0, 1.0, interp, white.ctypes.data_as(u8p), 0, d_tall_o.data_ptr(), COLS, TR, COLS, None)),
           2 * N * T)
print(json.dumps(out))
