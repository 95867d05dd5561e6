"""Seeded synthetic OMR answer cards (SURVEY.md section 8d / BASELINE.md section 2).

White page, black frame, timing marks down the left edge, four corner registration squares,
a grid of answer bubbles (outlined; filled black with p = 0.25), a few text-like strokes;
then a skew theta* ~ U(-9.5, 9.5) deg injected by bilinear rotation with white fill and
N(0, 4) grey noise.  Pure numpy; no reference code involved.
"""
import math

import numpy as np


def _card(rows, cols, rng):
    img = np.full((rows, cols), 255, np.uint8)
    s = min(cols / 2480.0, rows / 3508.0)

    def px(v, lo=1):
        return max(lo, int(round(v * s)))

    m = px(90)  # page margin
    t = px(6)   # frame thickness
    img[m:m + t, m:cols - m] = 0
    img[rows - m - t:rows - m, m:cols - m] = 0
    img[m:rows - m, m:m + t] = 0
    img[m:rows - m, cols - m - t:cols - m] = 0
    q = px(60)  # registration squares
    o = m + px(30)
    for (y0, x0) in ((o, o), (o, cols - o - q), (rows - o - q, o), (rows - o - q, cols - o - q)):
        img[y0:y0 + q, x0:x0 + q] = 0
    # bubble grid
    n_r, n_c = 60, 40
    gx0, gx1 = m + px(220), cols - m - px(120)
    gy0, gy1 = m + px(160), rows - m - px(160)
    bw, bh = px(28, 2), px(18, 2)
    pitch_x = (gx1 - gx0) / n_c
    pitch_y = (gy1 - gy0) / n_r
    filled = rng.random((n_r, n_c)) < 0.25
    ot = px(2)
    for r in range(n_r):
        y = int(gy0 + r * pitch_y)
        # timing mark for this row
        img[y:y + bh, m + px(40):m + px(40) + px(50, 2)] = 0
        for c in range(n_c):
            x = int(gx0 + c * pitch_x)
            if filled[r, c]:
                img[y:y + bh, x:x + bw] = 0
            else:
                img[y:y + ot, x:x + bw] = 0
                img[y + bh - ot:y + bh, x:x + bw] = 0
                img[y:y + bh, x:x + ot] = 0
                img[y:y + bh, x + bw - ot:x + bw] = 0
    # text-like strokes in the header band
    for _ in range(40):
        ylo, xlo = m + px(40), m + px(200)
        y = int(rng.integers(ylo, max(ylo + 1, gy0 - px(20))))
        x = int(rng.integers(xlo, max(xlo + 1, cols - m - px(300))))
        img[y:y + px(4), x:x + int(rng.integers(px(20), px(200) + 1))] = 0
    return img


def rotate_bilinear(img, angle_deg, fill=255.0):
    """Rotate about the image centre (positive = counter-clockwise with y down, like OpenCV)."""
    rows, cols = img.shape
    a = math.radians(angle_deg)
    ca, sa = math.cos(a), math.sin(a)
    cx, cy = cols / 2.0, rows / 2.0
    out = np.empty((rows, cols), np.float32)
    src = img.astype(np.float32)
    band = max(1, (1 << 22) // cols)
    xs = np.arange(cols, dtype=np.float32) - cx
    for y0 in range(0, rows, band):
        ys = np.arange(y0, min(rows, y0 + band), dtype=np.float32)[:, None] - cy
        # inverse map of the forward rotation [[ca, sa], [-sa, ca]]
        sx = ca * xs[None, :] - sa * ys + cx
        sy = sa * xs[None, :] + ca * ys + cy
        x0 = np.floor(sx).astype(np.int32)
        y0i = np.floor(sy).astype(np.int32)
        fx, fy = sx - x0, sy - y0i

        def tap(yy, xx):
            ok = (xx >= 0) & (xx < cols) & (yy >= 0) & (yy < rows)
            v = np.full(xx.shape, fill, np.float32)
            v[ok] = src[yy[ok], xx[ok]]
            return v

        v = (tap(y0i, x0) * (1 - fx) * (1 - fy) + tap(y0i, x0 + 1) * fx * (1 - fy)
             + tap(y0i + 1, x0) * (1 - fx) * fy + tap(y0i + 1, x0 + 1) * fx * fy)
        out[y0:y0 + v.shape[0]] = v
    return out


def make_card(rows, cols, seed, skew=None, noise_sigma=4.0):
    """Returns (gray u8 [rows, cols], injected skew in degrees)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    card = _card(rows, cols, rng)
    theta = float(rng.uniform(-9.5, 9.5)) if skew is None else float(skew)
    # same protocol as packages/core/src/main.rs:41-51: the sheet is rotated by -theta, so the
    # corrector is expected to report +theta
    rot = rotate_bilinear(card, -theta)
    if noise_sigma > 0:
        rot = rot + rng.normal(0.0, noise_sigma, rot.shape).astype(np.float32)
    return np.clip(np.rint(rot), 0, 255).astype(np.uint8), theta


def make_binary_card(rows, cols, seed, skew=None):
    """Card already binarised like transfer.rs:294-301 (0 / 255)."""
    g, theta = make_card(rows, cols, seed, skew)
    return np.where(g > 127, 255, 0).astype(np.uint8), theta
