"""FFT path (BASELINE config 5) measurement: magnitude_log pictures of a batch of 4096x4096 (and
2480x3508) scans resident in HBM through omr_fft_image_batch_device.  GPU only: the agreement of the pictures
with the oracle is what tests/test_gpu_fft.py and tests/test_gpu_c3.py check (nothing outside tests/, smoke()
and bench.py's cpu_baseline touches oracle/).  Rates are the MEAN over the repetitions.
A3 at 600 dpi (9921 x 14032: both axes beyond 8192 points, the global-memory chirp-z of fft_big.hip) runs 2 scans.
Usage: python tools/bench_fft.py [batch] [reps] [c5|a4|a3|both]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import fft, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
WHICH = sys.argv[3] if len(sys.argv) > 3 else "both"  # c5 | a4 | both
dev = torch.device("cuda:0")
out = {}
for name, rows, cols in (("C5 4096x4096", 4096, 4096), ("A4 2480x3508", 3508, 2480), ("A3 600 dpi 9921x14032", 14032, 9921)):
    if (WHICH == "both" and name.startswith("A3")) or (WHICH != "both" and not name.lower().startswith(WHICH)):
        continue
    if name.startswith("A3"):
        B = 2
    cards = [synth.make_card(rows, cols, 3 + i)[0] for i in range(2)]
    d = torch.from_numpy(np.stack([cards[i % 2] for i in range(B)])).to(dev)
    o = torch.zeros((B, rows, cols), dtype=torch.uint8, device=dev)
    fft.fft_image_batch_device(d.data_ptr(), B, rows * cols, rows, cols, cols, o.data_ptr())  # warm-up: the same launches as the timed calls, so that per-kernel profiles average equal launches
    ts = []
    for _ in range(REPS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fft.fft_image_batch_device(d.data_ptr(), B, rows * cols, rows, cols, cols, o.data_ptr())
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    px = rows * cols
    t = float(np.mean(ts))
    # algorithmic bytes per scan = SURVEY.md Appendix C's count for config 5 (round-3 / round-4 verdicts: price the path on the
    # survey's bytes, not on this builder's own formulation): the u8 scan in + four passes over the complex f32 spectrum =
    # 1 + 4 x 8 = 33 B per pixel (553 MB per 4096 x 4096 scan)
    alg = px * (1 + 4 * 8)
    # what the kernels actually move per scan: u8 in, the half spectrum (complex f32, cols / 2 + 1 columns) written
    # and read once, |F| of the half spectrum (f32) written and read once, two 8-bit pictures out
    half = rows * (cols // 2 + 1)
    moved = px * 1 + half * 8 * 2 + half * 4 * 2 + px * 2
    out[name] = {"batch": B, "reps": REPS, "scans_per_s": B / t, "ms_per_scan": t / B * 1e3, "best_ms_per_scan": min(ts) / B * 1e3,
                 "algorithmic_bytes_per_scan": alg, "algorithmic_GBps": alg * B / t / 1e9,
                 "kernel_bytes_per_scan": moved, "kernel_GBps": moved * B / t / 1e9}
    del d, o
print(json.dumps(out))
