"""Per-call latency of the host-image drivers (what the Rust shim would call once per file): upload, device
pipeline, download, including every allocation the call makes.  Usage: python tools/bench_calls.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch  # noqa: F401

from oics import hough, omr, projection, synth, transfer


def timeit(f, n=10):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3)


out = {}
g, _ = synth.make_card(1754, 1240, 5)          # a dataset-sized sheet (1240 x 1754)
bgr = np.stack([g, g, g], axis=2)
out["get_result_from_projection 1240x1754 bgr (45, 0.2, 248, 230) ms"] = timeit(
    lambda: omr.get_result_from_projection(bgr, 45, 0.2, 248, 230))
out["correct_default 1240x1754 bgr ms"] = timeit(lambda: omr.correct_default(bgr, 45, 0.2, 248, 230, 150.0, 50.0), 5)
out["get_result_from_edges_detection 1240x1754 bgr ms"] = timeit(
    lambda: omr.get_result_from_edges_detection(bgr, 150.0, 50.0), 3)
g2, _ = synth.make_card(3508, 2480, 2)
c3 = np.stack([g2, g2, g2], axis=2)
out["get_angle_with_projections 2480x3508 3-channel (10, 0.05, scale 1) ms"] = timeit(
    lambda: projection.get_angle_with_projections(c3, 10, 0.05, 1.0, 1), 5)
out["rotate_mat CONTAIN NEAREST 2480x3508 ms"] = timeit(
    lambda: transfer.rotate_mat(g2, 3.3, 1.0, 0, 0, (255, 255, 255, 0), transfer.RotateClipStrategy.CONTAIN), 5)
out["canny 2480x3508 ms"] = timeit(lambda: hough.canny(g2), 5)
# the Tauri host runs correct_default on a pool of OS threads, one file each (thread_pool.rs:41-88): files/s with
# T host threads calling the re-entrant entry point at once (dataset-sized sheets; ctypes releases the GIL)
from concurrent.futures import ThreadPoolExecutor
sheets = []
for i in range(8):
    gi, _ = synth.make_card(1150, 1240, 40 + i)
    sheets.append(np.stack([gi, gi, gi], axis=2))
for T in (1, 4, 16):
    n = 64 * T if T > 1 else 64
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(lambda k: omr.correct_default(sheets[k % 8], 45, 0.2, 248, 230, 150.0, 50.0), range(T)))  # warm-up
        t0 = time.perf_counter()
        list(ex.map(lambda k: omr.correct_default(sheets[k % 8], 45, 0.2, 248, 230, 150.0, 50.0), range(n)))
        dt = time.perf_counter() - t0
    out["correct_default 1240x1150 bgr, %d host thread(s): files/s" % T] = n / dt
print(json.dumps(out, indent=1))
