"""The reference's own comparative benchmark (packages/core/src/main.rs:17-252) through the drop-in API, on its own
dataset: every sheet is skewed by a random angle in [-10, 10) (rotate_mat(-angle, 1.0, INTER_LINEAR, white, CONTAIN),
main.rs:42-52), then each method estimates the angle and the sheet is rotated back (CONTAIN, INTER_LINEAR):
  projection  get_angle_with_projections(img, 45, 0.2, 0.2, 1)            main.rs:67-97
  hough       get_angle_with_hough(gray, 125.0, 15.0)                      main.rs:99-134
  fft         get_angle_with_fft(gray, 125.0, 150.0, 150.0, 75.0)          main.rs:136-172
Printed like main.rs:177-250: mean run time per sheet (ms; the reference's figure also holds its JPEG write, which
stays on the host side of the boundary and is left out here) and mean / std-dev / max of |estimate - injected|.
The reference publishes no numbers for this protocol; the angles are seeded here (PCG64) instead of
rand::thread_rng.  Usage: python tools/core_protocol.py [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

import dataset_pin as dp
import oics
from oics import fft, hough, projection, transfer
from oics.types import RotateClipStrategy

WHITE = (255.0, 255.0, 255.0, 0.0)


def run(seed=2024):
    rng = np.random.Generator(np.random.PCG64(seed))
    res = {"projection": ([], []), "hough": ([], []), "fft": ([], [])}
    fails = {"projection": 0, "hough": 0, "fft": 0}
    names = dp.sheets()
    for nm in names:
        _one(nm, rng, res, fails)
    out = {"sheets": len(names), "seed": seed}
    for method, (ts, dv) in res.items():
        d = np.array(dv)
        out[method] = {"mean_ms_per_sheet": float(np.mean(ts)), "deviation_mean_deg": float(d.mean()),
                       "deviation_std_deg": float(d.std()), "deviation_max_deg": float(d.max()),
                       "within_0.5_deg": int((d < 0.5).sum()), "answered": len(dv), "no_answer": fails[method]}
    return out


def _one(nm, rng, res, fails):
    img = dp.imread_color(nm)
    ang = float(rng.uniform(-10.0, 10.0))
    skew = transfer.rotate_mat(img, -ang, 1.0, 1, 0, WHITE, RotateClipStrategy.CONTAIN)
    gray = transfer.transfer_rgb_image_to_gray_image(skew)
    for method in ("projection", "hough", "fft"):
        t0 = time.perf_counter()
        try:
            if method == "projection":
                est = projection.get_angle_with_projections(skew, 45, 0.2, 0.2, 1)
            elif method == "hough":
                est = hough.get_angle_with_hough(gray, 125.0, 15.0)
            else:
                est = fft.get_angle_with_fft(gray, 125.0, 150.0, 150.0, 75.0)
            est = float(est[0] if isinstance(est, tuple) else est)
            transfer.rotate_mat(skew, est, 1.0, 1, 0, WHITE, RotateClipStrategy.CONTAIN)
        except oics.OmrError:
            fails[method] += 1  # (no segment found: the reference panics on angles[0] here)
            continue
        res[method][0].append((time.perf_counter() - t0) * 1e3)
        res[method][1].append(abs(est - ang))


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 2024), indent=1))
