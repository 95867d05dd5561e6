"""Scan-lane sweep timing (development aid): `n` A4 scans resident in HBM (8 distinct seeded cards), omr_batch_set_lanes,
HIP-event time of the sweep kernel per launch and the wall-clock rate of the whole batch (pack, sweep, column counts,
std-dev, arg-max).  Usage: python tools/klanes.py [scans, default 512] [scans per launch, default = scans] [passes] [rows cols, default A4 at 300 dpi]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import projection, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else n
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ROWS, COLS = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (3508, 2480)
cards = [synth.make_card(ROWS, COLS, 3 + i) for i in range(8)]
dev = torch.device("cuda:0")
buf = torch.empty((n, ROWS, COLS), dtype=torch.uint8, device=dev)
for i in range(n):
    buf[i] = torch.from_numpy(cards[i % 8][0]).to(dev)
best = torch.zeros(n, dtype=torch.int32, device=dev)
b = projection.Batch(ROWS, COLS, 10, 0.05, n_streams=1)
t0 = time.perf_counter()
b.set_lanes(lanes)
print("plan + scratch for %d scans per launch: %.3f s (programs generated on the device, %.2f GB)" % (lanes, time.perf_counter() - t0, b.lanes_program_bytes() / 1e9), flush=True)
b.set_timing(True)
for it in range(passes):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.run_device(buf.data_ptr(), ROWS * COLS, COLS, n, 127, best.data_ptr())
    b.sync()
    wall = time.perf_counter() - t0
    ms, k = b.kernel_ms()
    print("pass %d: sweep kernel %.3f ms per launch of %d scans (%d launches) = %.3f ms per 8 scans; whole batch %.1f images/s"
          % (it, ms / k, min(lanes, n), k, ms / k * 8 / min(lanes, n), n / wall), flush=True)
bi = best.cpu().numpy()
err = max(abs((bi[i] - 200) * 0.05 - cards[i % 8][1]) for i in range(n))
print("max |detected - injected| = %.3f deg" % err)
b.close()
