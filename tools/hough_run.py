"""GPU-only Hough-path run for profiling (no oracle, no child processes: safe directly after `rocprofv3 ... --`):
a batch of B 2480x3508 scans resident in HBM through omr_edges_detection_batch_device, R repetitions.
Usage: python3 tools/hough_run.py [batch] [distinct] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np, torch
from oics import omr, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ROWS, COLS = 3508, 2480
cards = [synth.make_card(ROWS, COLS, 3 + i)[0] for i in range(D)]
d = torch.from_numpy(np.stack([cards[i % D] for i in range(B)])).to("cuda:0")
omr.edges_detection_batch_device(d.data_ptr(), min(B, 2), ROWS * COLS, ROWS, COLS, 1, COLS, 150.0, 50.0)  # warm-up
ts = []
for _ in range(R):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ang, st, nl = omr.edges_detection_batch_device(d.data_ptr(), B, ROWS * COLS, ROWS, COLS, 1, COLS, 150.0, 50.0)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({"batch": B, "distinct": D, "seconds": min(ts), "scans_per_s": B / min(ts), "mean_segments": float(np.mean(nl))}))
