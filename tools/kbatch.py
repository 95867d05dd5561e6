"""Batch-mode sweep timing (development aid): 8 distinct seeded A4 cards, `reps` launch groups of 8 scans through
omr_batch_run_device with HIP-event timing of the sweep stage, plus the wall-clock rate of the whole batch.
Usage: python tools/kbatch.py [launch groups] [scans per launch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import projection, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ROWS, COLS = 3508, 2480
cards = [synth.make_card(ROWS, COLS, 3 + i) for i in range(8)]
dev = torch.device("cuda:0")
n = reps * G
buf = torch.empty((n, ROWS, COLS), dtype=torch.uint8, device=dev)
for i in range(n):
    buf[i] = torch.from_numpy(cards[i % 8][0]).to(dev)
best = torch.zeros(n, dtype=torch.int32, device=dev)
b = projection.Batch(ROWS, COLS, 10, 0.05, n_streams=1)
b.set_group(G)
b.set_timing(True)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.run_device(buf.data_ptr(), ROWS * COLS, COLS, n, 127, best.data_ptr())
    b.sync()
    wall = time.perf_counter() - t0
    ms, k = b.kernel_ms()
    print("pass %d: sweep stage %.3f ms per launch of %d scans (%d launches); whole batch %.1f images/s" % (it, ms / k, G, k, n / wall))
bi = best.cpu().numpy()
err = max(abs((bi[i] - 200) * 0.05 - cards[i % 8][1]) for i in range(n))
print("max |detected - injected| = %.3f deg; candidates run-merged / gathered: %s" % (err, b.info()))
b.close()
