"""The batch warp on its own (development aid): 8 A4 scans per launch, the batch context synchronised after every
call so that no sweep of a following group shares the chip with the warp.  Run under
`rocprofv3 --kernel-trace --stats` to read deskew_warp_kernel's duration; prints the wall-clock figures.
Usage: python tools/bench_deskew.py [reps] [scans per launch, default 8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import projection, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ROWS, COLS = 3508, 2480
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
base = [synth.make_card(ROWS, COLS, 3 + i)[0] for i in range(8)]
cards = [base[i % 8] for i in range(G)]
dev = torch.device("cuda:0")
scans = torch.from_numpy(np.stack(cards)).to(dev)
b = projection.Batch(ROWS, COLS, 10, 0.05, n_streams=1)
b.set_group(G) if G <= 64 else b.set_lanes(G)
dr, dc = b.deskew_canvas()
out = torch.empty((G, dr, dc), dtype=torch.uint8, device=dev)
size = torch.zeros((G, 2), dtype=torch.int32, device=dev)
best = torch.zeros(G, dtype=torch.int32, device=dev)
for name, interp in (("nearest", 0), ("linear", 1)):
    ts = []
    for it in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.deskew_device(scans.data_ptr(), ROWS * COLS, COLS, G, 127, interp, 255, out.data_ptr(), dr * dc, dc, size.data_ptr(),
                        best.data_ptr())
        b.sync()
        if it >= 2:
            ts.append(time.perf_counter() - t0)
    sz = size.cpu().numpy().astype(np.int64)
    byt = float((sz[:, 0] * sz[:, 1]).sum() + G * ROWS * COLS)
    print("%s: sweep + warp of %d scans %.3f ms per call (synchronised); warp traffic %.1f MB per call" % (name, G, np.mean(ts) * 1e3, byt / 1e6))
b.close()
