"""The reference's enabled library test in full (packages/lib/src/lib.rs:130-245): 104 sheets x 900 injected angles
(-45.0 .. +44.9 by 0.1, lib.rs:153-154) through the drop-in API on the GPU -- a tool, not a test (93 600 full
pipeline runs).  Per case, as lib.rs:156-205: imread COLOR -> rotate_mat(-angle, 1.0, INTER_LINEAR, BORDER_CONSTANT
white, DEFAULT) -> RGB2GRAY -> GRAY2RGB -> JPEG quality 100 round trip -> correct_default(45, 0.2, 248, 230, 150.0,
50.0).  The rotation and the grey conversion run through omr_rotate / omr_rgb_to_gray (both exact against the
oracle in tests/), the codec is PIL on the host (the codec stays on the host side of the boundary).
Writes the lib.rs:220-226 histogram, the error statistics of the believed cases and the sheets' NATIVE skew
(get_angle_with_projections(2, 0.05, 0.5) on the un-rotated sheets) as markdown + JSON.
Usage (GPU box): python tools/dataset_full.py [out_prefix] [angles per sheet, default 900] [threads, default 16] [first sheet] [end sheet]
       python tools/dataset_full.py --merge out_prefix part1.json part2.json ..   (a gpurun call lasts 20 minutes at most: the
       104 sheets go in two calls of 52, the parts are merged on any machine)"""
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
from PIL import Image

from oics import omr, projection, transfer
from oics.types import RotateClipStrategy

DATASET = os.path.join(ROOT, "tests", "golden", "dataset")
PARAMS = (45, 0.2, 248, 230, 150.0, 50.0)  # lib.rs:192-205
MERGE = len(sys.argv) > 1 and sys.argv[1] == "--merge"
if MERGE:
    sys.argv.pop(1)
out_prefix = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "dataset_full")
n_angles = int(sys.argv[2]) if len(sys.argv) > 2 and not MERGE else 900
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 and not MERGE else 16
idxs = list(range(-450, 450)) if n_angles >= 900 else sorted(
    int(v) for v in np.random.Generator(np.random.PCG64(7)).choice(900, n_angles, replace=False) - 450)


def imread_color(name):
    g = np.array(Image.open(os.path.join(DATASET, name)).convert("L"))
    return np.ascontiguousarray(np.stack([g, g, g], axis=2))


def one_case(bgr, idx):
    angle = idx * 0.1
    sk = transfer.rotate_mat(bgr, -angle, 1.0, 1, 0, (255.0, 255.0, 255.0, 0.0), RotateClipStrategy.DEFAULT).get_mat()
    gray = transfer.transfer_rgb_image_to_gray_image(sk).get_mat()
    buf = io.BytesIO()
    Image.fromarray(np.stack([gray, gray, gray], axis=2)).save(buf, format="JPEG", quality=100)
    buf.seek(0)
    back = np.ascontiguousarray(np.array(Image.open(buf).convert("RGB"))[:, :, ::-1])
    detected, need_check = omr.correct_default(back, *PARAMS)[:2]
    return idx, angle, float(detected), bool(need_check)


sheets = sorted(f for f in os.listdir(DATASET) if f.lower().endswith(".jpg"))
t0 = time.time()
rows = []
native = {}
seconds = 0.0
if MERGE:
    for part in sys.argv[2:]:
        rec = json.load(open(part))
        rows += [tuple(r) for r in rec["rows"]]
        native.update(rec["native_skew"])
        seconds += rec["seconds"]
    sheets = sorted(native)
    idxs = sorted({r[1] for r in rows})
else:
    s_begin = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    s_end = int(sys.argv[5]) if len(sys.argv) > 5 else len(sheets)
    sheets = sheets[s_begin:s_end]
    with ThreadPoolExecutor(n_threads) as pool:
        for si, s in enumerate(sheets):
            bgr = imread_color(s)
            native[s] = float(projection.get_angle_with_projections(bgr, 2, 0.05, 0.5, 1))
            for idx, angle, det, chk in pool.map(lambda i: one_case(bgr, i), idxs):
                rows.append((s, idx, angle, det, chk))
            print("%3d / %d sheets, %d cases, %.0f s" % (si + 1, len(sheets), len(rows), time.time() - t0), flush=True)
    seconds = time.time() - t0

cls = {"SUCCESS": 0, "NOT_SO_RIGHT": 0, "ERROR": 0, "NOT_BELIEVED": 0}
diffs = []
worst = []
for s, idx, angle, det, chk in rows:
    d = abs(angle - det)
    if chk:
        c = "NOT_BELIEVED"
    elif d > 0.5:
        c = "ERROR"
    elif d > 0.4:
        c = "NOT_SO_RIGHT"
    else:
        c = "SUCCESS"
    cls[c] += 1
    if not chk:
        diffs.append(angle - det)
        if d > 0.4:
            worst.append((d, s, angle, det))
diffs = np.array(diffs)
nat = np.array(list(native.values()))
n = len(rows)
summary = {
    "cases": n, "sheets": len(sheets), "angles_per_sheet": len(idxs), "seconds": seconds, "classes": cls,
    "believed": int(diffs.size), "believed_lt_0.5": int((np.abs(diffs) < 0.5).sum()), "believed_le_0.4": int((np.abs(diffs) <= 0.4).sum()),
    "signed_error_mean": float(diffs.mean()), "signed_error_std": float(diffs.std()),
    "signed_error_positive_share": float((diffs > 0).mean()), "signed_error_negative_share": float((diffs < 0).mean()),
    "abs_error_mean": float(np.abs(diffs).mean()), "abs_error_max": float(np.abs(diffs).max()),
    "native_skew_mean": float(nat.mean()), "native_skew_min": float(nat.min()), "native_skew_max": float(nat.max()),
    "native_skew_negative_sheets": int((nat < 0).sum()),
    "worst": sorted(worst, reverse=True)[:20],
}
os.makedirs(os.path.dirname(out_prefix), exist_ok=True)
json.dump({"summary": summary, "native_skew": native, "seconds": seconds, "rows": rows if not MERGE else []},
          open(out_prefix + ".json", "w"))
with open(out_prefix + ".md", "w") as f:
    f.write("# The reference's library test in full (lib.rs:130-245) through the drop-in API on the GPU\n\n")
    f.write("%d sheets x %d injected angles = %d runs of `correct_default(45, 0.2, 248, 230, 150.0, 50.0)`, %.0f s "
            "(tools/dataset_full.py, %d host threads; skew injection and grey conversion on the GPU, JPEG q100 round trip by PIL).\n\n"
            % (len(sheets), len(idxs), n, summary["seconds"], n_threads))
    f.write("| class (lib.rs:220-226) | cases | share |\n|---|---|---|\n")
    for k in ("SUCCESS", "NOT_SO_RIGHT", "ERROR", "NOT_BELIEVED"):
        f.write("| %s | %d | %.3f %% |\n" % (k, cls[k], 100.0 * cls[k] / n))
    f.write("\nBelieved cases: %d; |injected - detected| < 0.5 deg (the criterion of lib.rs:103-113): %d (%.4f %%); <= 0.4 deg: %.3f %% "
            "(the reference's comment claims 99.9 %%, lib.rs:108).\n" % (diffs.size, summary["believed_lt_0.5"],
                                                                        100.0 * summary["believed_lt_0.5"] / diffs.size,
                                                                        100.0 * summary["believed_le_0.4"] / diffs.size))
    f.write("\nSigned error injected - detected over the believed cases: mean %+.3f deg, std %.3f, positive in %.1f %%, negative in %.1f %%; "
            "|error| mean %.3f, max %.3f.\n" % (summary["signed_error_mean"], summary["signed_error_std"],
                                                100 * summary["signed_error_positive_share"], 100 * summary["signed_error_negative_share"],
                                                summary["abs_error_mean"], summary["abs_error_max"]))
    f.write("\nNative skew of the un-rotated sheets (`get_angle_with_projections(2, 0.05, 0.5)`): mean %+.3f deg, range %+.2f .. %+.2f, "
            "negative on %d of %d sheets -- the scans themselves are tilted, so `injected - detected` is biased by about minus that.\n"
            % (summary["native_skew_mean"], summary["native_skew_min"], summary["native_skew_max"], summary["native_skew_negative_sheets"], len(sheets)))
    if worst:
        f.write("\nLargest errors among the believed cases:\n\n| |error| | sheet | injected | detected |\n|---|---|---|---|\n")
        for d, s, a, det in summary["worst"]:
            f.write("| %.3f | %s | %.1f | %.2f |\n" % (d, s, a, det))
print(json.dumps(summary)[:600])
