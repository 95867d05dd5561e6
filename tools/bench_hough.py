"""Hough-line path (BASELINE config 4) measurement: a batch of 2480x3508 scans resident in HBM through
omr_edges_detection_batch_device (Canny + sequential-order HoughLinesP, one workgroup per scan).  GPU only:
bit-exact agreement with the oracle is what tests/test_gpu_hough.py and tests/test_gpu_c3.py check (nothing
outside tests/, smoke() and bench.py's cpu_baseline touches oracle/).  The rate is the MEAN over the repetitions.
With a fourth argument (comma-separated, e.g. 64,128,192,256,0) the batch is repeated for each setting of
omr_hough_set_scans_in_flight (0 = the library's default) and the rates are listed under "in_flight_sweep".
Usage: python tools/bench_hough.py [batch] [distinct] [reps] [in-flight list]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import omr, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DISTINCT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SWEEP = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else []
ROWS, COLS = 3508, 2480
MLL, MLG = 150.0, 50.0  # the reference's defaults (lib.rs:220-226 parameters)

cards = [synth.make_card(ROWS, COLS, 3 + i)[0] for i in range(DISTINCT)]
dev = torch.device("cuda:0")
host = np.stack([cards[i % DISTINCT] for i in range(B)])
d = torch.from_numpy(host).to(dev)
torch.cuda.synchronize()

# warm-up
ang, st, nl = omr.edges_detection_batch_device(d.data_ptr(), min(B, DISTINCT), ROWS * COLS, ROWS, COLS, 1, COLS, MLL, MLG)

ts = []
for _ in range(REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ang, st, nl = omr.edges_detection_batch_device(d.data_ptr(), B, ROWS * COLS, ROWS, COLS, 1, COLS, MLL, MLG)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
gpu = B / float(np.mean(ts))
rep_same = bool((ang.view(np.uint64).reshape(-1, DISTINCT) == ang.view(np.uint64)[:DISTINCT]).all()) if B % DISTINCT == 0 else None

sweep = {}
for k in SWEEP:
    omr.hough_set_scans_in_flight(k)
    tk = []
    for _ in range(REPS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a2, _, _ = omr.edges_detection_batch_device(d.data_ptr(), B, ROWS * COLS, ROWS, COLS, 1, COLS, MLL, MLG)
        torch.cuda.synchronize()
        tk.append(time.perf_counter() - t0)
    assert (a2.view(np.uint64) == ang.view(np.uint64)).all(), "the result must not depend on the scans in flight"
    sweep[str(k)] = B / float(np.mean(tk))
    print("in flight %4d: %.0f scans/s" % (k, sweep[str(k)]), flush=True)
omr.hough_set_scans_in_flight(0)

out = {
    "workload": "C4: %d scans 2480x3508 u8 resident in HBM, Canny(50,150,3) + HoughLinesP(1, pi/180, 0, %g, %g) + vote"
                % (B, MLL, MLG),
    "gpu_scans_per_s": gpu, "gpu_batch_seconds_mean": float(np.mean(ts)), "gpu_batch_seconds_best": min(ts), "batch": B,
    "distinct_cards": DISTINCT, "reps": REPS, "mean_edge_segments_per_scan": float(np.mean(nl)),
    "repeats_of_a_card_identical": rep_same,
}
if sweep:
    out["in_flight_sweep"] = sweep
print(json.dumps(out))
