"""Randomised parity run (development aid, not part of the test suite): random shapes, candidate ranges and
cards through the automatic kernel selection against the CPU oracle, bit for bit.
Usage: python tools/fuzz_sweep.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch  # noqa: F401

from oics import projection, synth
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
orc.build()
bad = 0
for c in range(cases):
    rows = int(rng.integers(8, 900))
    cols = int(rng.integers(8, 900))
    if c % 10 == 0:
        rows, cols = int(rng.integers(1500, 4200)), int(rng.integers(8, 300))   # tall: several flushes
    if c % 10 == 5:
        rows, cols = int(rng.integers(8, 300)), int(rng.integers(1500, 4200))   # wide: many word groups
    max_angle = int(rng.choice([2, 5, 10, 15, 45]))
    step = float(rng.choice([1.0, 0.5, 0.25, 0.2]))
    scale = float(rng.choice([1.0, 1.0, 1.0, 0.5, 0.2]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        b, _ = synth.make_binary_card(rows, cols, int(rng.integers(1, 10**6)))
    elif kind == 1:
        b = np.where(rng.random((rows, cols)) < rng.uniform(0.01, 0.6), 0, 255).astype(np.uint8)
    else:
        b = np.full((rows, cols), 255, np.uint8)
        b[rng.integers(0, rows, 50), rng.integers(0, cols, 50)] = 0
        b[0, :] = 0
        b[:, -1] = 0
    plan = projection.SweepPlan(rows, cols, max_angle, step, scale)
    vp, hp, vs, hs, best = plan.run(b)
    n_runs, n_gather = plan.info()[:2] if hasattr(plan, "info") else (-1, -1)
    plan.close()
    evp, ehp, evs, ehs = orc.sweep(b, max_angle, step, scale)
    ok = (vp == evp).all() and (hp == ehp).all() and (vs.view(np.uint64) == evs.view(np.uint64)).all() and \
        (hs.view(np.uint64) == ehs.view(np.uint64)).all()
    if not ok:
        bad += 1
        print("MISMATCH case", c, rows, cols, max_angle, step, scale, kind)
print("cases %d mismatches %d" % (cases, bad))
