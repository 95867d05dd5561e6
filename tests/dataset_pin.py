"""Shared pieces of the dataset pin (tests/test_dataset_pin.py, tests/golden/make_dataset_pin.py).

The one numeric acceptance criterion the reference itself holds (packages/lib/src/lib.rs):
  * protocol lib.rs:132-205 -- every sheet of dataset/dataset, skew injected at angles on the grid
    test_iter_idx * 0.1, idx in -450..450 (:153-154): imread COLOR -> rotate_mat(-angle, 1.0, INTER_LINEAR,
    BORDER_CONSTANT white, DEFAULT) (:156-166) -> RGB2GRAY -> GRAY2RGB (:168-184) -> tmp.jpg quality 100
    (:186-188) -> correct_default(tmp, out, 45, 0.2, 248, 230, 150.0, 50.0) (:192-205);
  * criterion lib.rs:103-113 -- |injected - detected| < 0.5 deg whenever !need_check ("99.9 % < 0.4");
  * classes lib.rs:220-226 -- NOT_BELIEVED / ERROR (> 0.5) / NOT_SO_RIGHT (> 0.4) / SUCCESS.
The sheets under tests/golden/dataset/ are the reference's own test data files (byte-identical copies of
/root/reference/dataset/dataset/*.jpg, made by make_dataset_pin.py); imread/imwrite are PIL here (the
codec stays on the host side of the boundary)."""
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATASET = os.path.join(HERE, "golden", "dataset")
EXPECTED = os.path.join(HERE, "golden", "dataset_pin_expected.json")
PARAMS = (45, 0.2, 248, 230, 150.0, 50.0)  # lib.rs:192-205
ANGLES_PER_SHEET = 9


def sheets():
    return sorted(f for f in os.listdir(DATASET) if f.lower().endswith(".jpg"))


def cases():
    """[(sheet, test_iter_idx)]: 9 seeded picks per sheet from lib.rs:153's -450..450 -> 936 cases."""
    rng = np.random.Generator(np.random.PCG64(2019))
    out = []
    for s in sheets():
        for idx in sorted(int(v) for v in rng.choice(900, ANGLES_PER_SHEET, replace=False) - 450):
            out.append((s, idx))
    return out


def imread_color(name):
    """imgcodecs::imread(IMREAD_COLOR): a grey JPEG decodes to three equal channels."""
    from PIL import Image
    g = np.array(Image.open(os.path.join(DATASET, name)).convert("L"))
    return np.ascontiguousarray(np.stack([g, g, g], axis=2))


def inject(bgr, angle, orc):
    """lib.rs:156-188 with the oracle's warp / grey conversion as the fixture maker, JPEG q100 round trip by PIL."""
    from PIL import Image
    sk = orc.rotate_mat(bgr, -angle, 1.0, interp=1, border=(255, 255, 255, 0), clip=0)
    gray = orc.rgb2gray(sk)
    rgb = np.stack([gray, gray, gray], axis=2)  # COLOR_GRAY2RGB
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", quality=100)
    buf.seek(0)
    back = np.array(Image.open(buf).convert("RGB"))  # imread COLOR of tmp.jpg (channels equal up to codec noise)
    return np.ascontiguousarray(back[:, :, ::-1])    # BGR order, as imread yields


def oracle_correct_default(bgr, orc, params=PARAMS):
    """omr.rs:339-402 composed from the oracle's parts -> (angle, need_check, projection status)."""
    ma, st, mw, mh, ml, mg = params
    pa, pst, pc = orc.get_result_from_projection(bgr, ma, st, mw, mh)
    if pst == 0:
        return pa, False, pst
    ea, _, _, nl = orc.get_result_from_edges_detection(bgr, ml, mg)
    if nl == 0:
        raise RuntimeError("no Hough line (the reference panics here, omr.rs:272)")
    ang, chk = orc.correct_default_decision(pa, pst, pc, ea)
    return ang, chk, pst


def classify(injected, detected, need_check):
    """lib.rs:220-226"""
    d = abs(injected - detected)
    if need_check:
        return "NOT_BELIEVED"
    if d > 0.5:
        return "ERROR"
    if d > 0.4:
        return "NOT_SO_RIGHT"
    return "SUCCESS"


def load_expected():
    return json.load(open(EXPECTED))
