"""Hough-line path (BASELINE config 4) measurement: a batch of 2480x3508 scans resident in HBM through
omr_edges_detection_batch_device (Canny + sequential-order HoughLinesP, one workgroup per scan), beside
the CPU oracle on the host cores; the first scans' results are checked against the oracle bit for bit.
Usage: python tools/bench_hough.py [batch] [distinct] [reps]"""
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
from oracle import oracle as orc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DISTINCT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ROWS, COLS = 3508, 2480
MLL, MLG = 150.0, 50.0  # the reference's defaults (lib.rs:220-226 parameters)

orc.build()
cards = [synth.make_card(ROWS, COLS, 3 + i)[0] for i in range(DISTINCT)]
dev = torch.device("cuda:0")
host = np.stack([cards[i % DISTINCT] for i in range(B)])
d = torch.from_numpy(host).to(dev)
torch.cuda.synchronize()

# warm-up + parity on a small batch
ang, st, nl = omr.edges_detection_batch_device(d.data_ptr(), min(B, DISTINCT), ROWS * COLS, ROWS, COLS, 1, COLS, MLL, MLG)
t0 = time.perf_counter()
exp = [orc.get_result_from_edges_detection(cards[i], MLL, MLG, fast=True) for i in range(min(B, DISTINCT, 4))]
t_cpu1 = (time.perf_counter() - t0) / len(exp)
parity = all(np.float64(ang[i]).view(np.uint64) == np.float64(e[0]).view(np.uint64) and st[i] == e[1] and nl[i] == e[3]
             for i, e in enumerate(exp))

ts = []
for _ in range(REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ang, st, nl = omr.edges_detection_batch_device(d.data_ptr(), B, ROWS * COLS, ROWS, COLS, 1, COLS, MLL, MLG)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
gpu = B / min(ts)

cores = min(os.cpu_count() or 1, 64)
work = [cards[i % DISTINCT] for i in range(2 * cores)]
t0 = time.perf_counter()
with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: one scan per core
    list(ex.map(lambda c: orc.get_result_from_edges_detection(c, MLL, MLG, fast=True), work))
cpu_all = len(work) / (time.perf_counter() - t0)

out = {
    "workload": "C4: %d scans 2480x3508 u8 resident in HBM, Canny(50,150,3) + HoughLinesP(1, pi/180, 0, %g, %g) + vote"
                % (B, MLL, MLG),
    "gpu_scans_per_s": gpu, "gpu_batch_seconds": min(ts), "batch": B, "distinct_cards": DISTINCT,
    "mean_edge_segments_per_scan": float(np.mean(nl)),
    "cpu_oracle_scans_per_s_1_thread": 1.0 / t_cpu1, "cpu_oracle_scans_per_s_all_cores": cpu_all, "cores": cores,
    "parity_vs_oracle": bool(parity),
}
print(json.dumps(out))
