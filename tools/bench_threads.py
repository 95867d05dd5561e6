import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from oics import omr, synth
sheets = []
for i in range(8):
    gi, _ = synth.make_card(1150, 1240, 40 + i)
    sheets.append(np.stack([gi, gi, gi], axis=2))
st = [int(omr.get_result_from_projection(s, 45, 0.2, 248, 230).status) for s in sheets]
print("projection status per sheet", st)
def run(fn, T, n):
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(fn, range(T)))
        t0 = time.perf_counter(); list(ex.map(fn, range(n))); return n / (time.perf_counter() - t0)
for T in (1, 4, 16):
    print("T", T, "projection only files/s %.0f" % run(lambda k: omr.get_result_from_projection(sheets[k % 8], 45, 0.2, 248, 230), T, 200 * T),
          " edges only files/s %.1f" % run(lambda k: omr.get_result_from_edges_detection(sheets[k % 8], 150.0, 50.0), T, 8 * T),
          " correct_default files/s %.1f" % run(lambda k: omr.correct_default(sheets[k % 8], 45, 0.2, 248, 230, 150.0, 50.0, want_image=False), T, 32 * T))
for wi in (False, True):
    for T in (1, 4):
        print("want_image", wi, "T", T, "correct_default files/s %.1f" % run(lambda k: omr.correct_default(sheets[k % 8], 45, 0.2, 248, 230, 150.0, 50.0, want_image=wi), T, 32 * T))
import ctypes as C
from oics._lib import lib
t0 = time.perf_counter()
for k in range(32):
    a, c, img = omr.correct_default(sheets[k % 8], 45, 0.2, 248, 230, 150.0, 50.0, want_image=True)
print("single thread with image: %.2f ms per call, output shape %s" % ((time.perf_counter() - t0) / 32 * 1e3, img.shape))
