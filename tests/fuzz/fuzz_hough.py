"""Randomised parity run of HoughLinesP alone (development aid): small and odd shapes, sparse to dense point sets
(including fewer points than one draw round and exact multiples of 64), every threshold / length / gap regime and
several accumulator resolutions -- segments must equal the CPU oracle's, in order.
Usage: python tools/fuzz_hough.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch  # noqa: F401

from oics import hough
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 7))
orc.build()
bad = 0
for c in range(cases):
    rows, cols = int(rng.integers(1, 420)), int(rng.integers(1, 420))
    img = np.zeros((rows, cols), np.uint8)
    kind = int(rng.integers(0, 4))
    if kind == 0:    # exact number of points (draw-round edges)
        n = min(rows * cols, int(rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 640, 1000])))
        idx = rng.choice(rows * cols, n, replace=False)
        img.reshape(-1)[idx] = 255
    elif kind == 1:  # random density
        img[rng.random((rows, cols)) < float(rng.choice([0.002, 0.02, 0.1, 0.4]))] = 255
    elif kind == 2:  # lines with gaps, plus noise
        for _ in range(int(rng.integers(1, 8))):
            y0, x0 = rng.integers(0, rows), rng.integers(0, cols)
            th = rng.random() * np.pi
            for t in range(int(rng.integers(5, 400))):
                y, x = int(round(y0 + t * np.sin(th))), int(round(x0 + t * np.cos(th)))
                if 0 <= y < rows and 0 <= x < cols and rng.random() < 0.85:
                    img[y, x] = 255
        img[rng.random((rows, cols)) < 0.005] = 255
    else:            # full rows / columns (dense runs along the walk)
        img[:: int(rng.integers(2, 9)), :] = 255
        img[:, :: int(rng.integers(3, 17))] = 255
    thr = int(rng.choice([0, 0, 1, 5, 20, 80]))
    mll = int(rng.choice([0, 1, 5, 30, 150]))
    mlg = int(rng.choice([0, 1, 3, 10, 50, 63, 64, 100, 300]))
    div, rho = [(180, 1.0), (180, 1.0), (90, 1.0), (250, 1.0), (180, 2.0), (64, 0.5)][int(rng.integers(0, 6))]
    exp = orc.hough_lines_p(img, mll, mlg, threshold=thr, rho=rho, theta=np.pi / div)
    got = hough.hough_lines_p(img, rho, np.pi / div, thr, mll, mlg)
    ok = got.shape == exp.shape and (got == exp).all()
    if not ok:
        bad += 1
        print("MISMATCH case %d: %dx%d kind %d thr %d mll %d mlg %d div %d rho %.1f: %d vs %d segments" % (
            c, cols, rows, kind, thr, mll, mlg, div, rho, len(got), len(exp)), flush=True)
print("hough fuzz: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
