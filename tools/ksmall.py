"""Batch sweep of the app's DEFAULT parameters (getLibParams.ts:30-60; core/src/main.rs:69-70): +-45 deg @ 0.2 deg = 450 candidates on
248 x 230 working images, `n` sheets resident in HBM (development aid).  Usage: python tools/ksmall.py [sheets] [group] [max_angle] [step]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

import oics
from oics import projection, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
group = int(sys.argv[2]) if len(sys.argv) > 2 else 64
max_angle = int(sys.argv[3]) if len(sys.argv) > 3 else 45
step = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
ROWS, COLS = 230, 248
cards = [synth.make_card(ROWS, COLS, 3 + i, skew=float(-40 + 5 * i)) for i in range(16)]
dev = torch.device("cuda:0")
buf = torch.from_numpy(np.stack([cards[i % 16][0] for i in range(n)])).to(dev)
best = torch.zeros(n, dtype=torch.int32, device=dev)
b = projection.Batch(ROWS, COLS, max_angle, step, n_streams=1)
mode = "run-merging + gather"
try:
    b.set_lanes(min(n, 512))
    mode = "scan-lane"
except oics.OmrError as e:
    b.set_group(group)
print("candidates on run-merging / gather kernels:", b.info(), "mode:", mode)
N, A = projection.candidate_count(max_angle, step)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.run_device(buf.data_ptr(), ROWS * COLS, COLS, n, 127, best.data_ptr())
    b.sync()
    dt = time.perf_counter() - t0
    print("pass %d: %d sheets %dx%d, %d candidates: %.2f ms = %.0f sheets/s" % (it, n, COLS, ROWS, A, dt * 1e3, n / dt))
bi = best.cpu().numpy()
print("max |detected - injected| = %.2f deg" % max(abs((bi[i] - N) * step - cards[i % 16][1]) for i in range(n)))
b.close()
