"""Independent numpy restatement of the sweep (second opinion on oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/oracle.h.  Written separately from oracle.c (vectorised
instead of per-pixel loops) so that a transcription slip in one of the two shows up as a
disagreement in tests/test_oracle_*.py.  Citations: /root/reference paths, or [OpenCV 4.6.0].
"""
import math

import numpy as np

CV_PI = 3.1415926535897932384626433832795


def get_rotation_matrix_2d(cx, cy, angle_deg, scale):
    """[OpenCV 4.6.0 getRotationMatrix2D]; centre is Point2f (transfer.rs:474)."""
    cx, cy = float(np.float32(cx)), float(np.float32(cy))
    a = angle_deg * (CV_PI / 180)
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    return np.array([alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy], np.float64)


def invert_affine(M):
    """[OpenCV 4.6.0 cv::warpAffine] inversion, same operation order."""
    M = [float(x) for x in M]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return np.array(M, np.float64)


def nn_coords(Minv, drows, dcols):
    """Integer source coordinates of every destination pixel, INTER_NEAREST."""
    x = np.arange(dcols, dtype=np.float64)
    y = np.arange(drows, dtype=np.float64)
    adelta = np.rint(Minv[0] * x * 1024).astype(np.int64)
    bdelta = np.rint(Minv[3] * x * 1024).astype(np.int64)
    X0 = np.rint((Minv[1] * y + Minv[2]) * 1024).astype(np.int64) + 512
    Y0 = np.rint((Minv[4] * y + Minv[5]) * 1024).astype(np.int64) + 512
    X = (X0[:, None] + adelta[None, :]) >> 10
    Y = (Y0[:, None] + bdelta[None, :]) >> 10
    return np.clip(X, -32768, 32767), np.clip(Y, -32768, 32767)


def warp_affine_nn(src, M, border=255):
    """Single-channel nearest warp onto the same canvas (transfer.rs:477-485)."""
    rows, cols = src.shape
    X, Y = nn_coords(invert_affine(M), rows, cols)
    inb = (X >= 0) & (X < cols) & (Y >= 0) & (Y < rows)
    out = np.full((rows, cols), border, np.uint8)
    out[inb] = src[Y[inb], X[inb]]
    return out


def standard_deviation(v):
    """calculate.rs:13-23, strictly sequential."""
    v = [float(t) for t in v]
    s = v[0]
    for t in v[1:]:
        s = s + t
    mean = s / float(len(v))
    acc = (v[0] - mean) * (v[0] - mean)
    for t in v[1:]:
        d = t - mean
        acc = acc + d * d
    return math.sqrt(acc / float(len(v)))


def candidate_count(max_angle, step):
    q = float(max_angle) / step
    N = max(0, min(65535, int(q)))
    return N, 2 * N


def sweep(bin_img, max_angle, step, matrix_scale=1.0):
    """projection.rs:47-65: per candidate (vproj, hproj, v_sd, h_sd)."""
    rows, cols = bin_img.shape
    N, A = candidate_count(max_angle, step)
    cx, cy = np.float32(cols) / np.float32(2), np.float32(rows) / np.float32(2)
    vp = np.zeros((A, cols), np.uint32)
    hp = np.zeros((A, rows), np.uint32)
    vs, hs = np.zeros(A), np.zeros(A)
    for i in range(A):
        M = get_rotation_matrix_2d(cx, cy, float(i - N) * step, matrix_scale)
        black = warp_affine_nn(bin_img, M) == 0
        vp[i] = black.sum(axis=0)
        hp[i] = black.sum(axis=1)
        vs[i] = standard_deviation(vp[i])
        hs[i] = standard_deviation(hp[i])
    return vp, hp, vs, hs


def argmax_path1(v, h):
    """projection.rs:125-190; returns the set of indices the reference may return."""
    n = len(v)
    vl, vmax = [0], v[0]
    for i in range(n):
        if v[i] > vmax:
            vmax, vl = v[i], [i]
        elif v[i] == vmax:
            vl.append(i)
    hl, hmax = [0], h[0]
    for i in range(n):
        if h[i] > hmax:
            hmax, hl = h[i], [i]
        elif h[i] == hmax:
            hl.append(i)
    if len(vl) == 1 and len(hl) == 1 and vl[0] == hl[0]:
        return {vl[0]}
    cand = sorted(set(vl) | set(hl))
    sdp = {i: v[i] * v[i] + h[i] * h[i] for i in cand}
    best = max(sdp.values())
    if not best > 0.0:
        return {n // 2}
    return {i for i in cand if sdp[i] == best}


# ---- Hough-line path: independent restatement (vectorised Canny, component-based hysteresis,
# ---- scalar-Python PPHT in numpy float32).  Small images only.
def canny_np(img, low=50.0, high=150.0):
    """Canny(…, 3, false) of OpenCV 4.6.0: Sobel 3x3 replicate, L1 magnitude, fixed-point sector
    test, hysteresis = connected components (8-neighbourhood) of the surviving pixels that hold a
    pixel above `high`.  hough.rs:27, omr.rs:239."""
    from scipy import ndimage
    a = np.asarray(img, np.int32)
    if a.ndim == 2:
        a = a[:, :, None]
    p = np.pad(a, ((1, 1), (1, 1), (0, 0)), mode="edge")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    mag_c = np.abs(dx) + np.abs(dy)
    best = np.argmax(mag_c, axis=2)  # first maximum
    ii, jj = np.indices(best.shape)
    gx, gy, mag = dx[ii, jj, best], dy[ii, jj, best], mag_c[ii, jj, best]
    lo, hi = int(np.floor(min(low, high))), int(np.floor(max(low, high)))
    m = np.pad(mag, 1)
    c = m[1:-1, 1:-1]
    ax, ay = np.abs(gx), np.abs(gy) << 15
    t22 = ax * 13573
    t67 = t22 + (ax << 16)
    horiz = (c > m[1:-1, :-2]) & (c >= m[1:-1, 2:])
    vert = (c > m[:-2, 1:-1]) & (c >= m[2:, 1:-1])
    same = (c > m[:-2, :-2]) & (c > m[2:, 2:])   # gradient signs equal: up-left / down-right
    anti = (c > m[:-2, 2:]) & (c > m[2:, :-2])
    diag = np.where((gx ^ gy) < 0, anti, same)
    keep = (c > lo) & np.where(ay < t22, horiz, np.where(ay > t67, vert, diag))
    strong = keep & (c > hi)
    lab, n = ndimage.label(keep, structure=np.ones((3, 3), int))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[strong])] = True
    good[0] = False
    return (good[lab] * 255).astype(np.uint8)


def hough_lines_p_py(edges, min_line_length, max_line_gap, rho=1.0, theta=np.pi / 180.0, threshold=0):
    """HoughLinesP of OpenCV 4.6.0 (progressive probabilistic Hough, RNG seed 2^64-1), scalar
    Python; float32 arithmetic via numpy scalars.  hough.rs:31-43, omr.rs:245-253."""
    f32 = np.float32
    e = np.asarray(edges)
    height, width = e.shape
    rho32, theta32 = f32(rho), f32(theta)
    irho = f32(1.0) / rho32
    numangle = int(np.rint(np.pi / float(theta32)))
    numrho = int(np.rint(f32((width + height) * 2 + 1) / rho32))
    cs = [f32(np.cos(float(n) * float(theta32)) * float(irho)) for n in range(numangle)]
    sn = [f32(np.sin(float(n) * float(theta32)) * float(irho)) for n in range(numangle)]
    line_length, line_gap = int(np.rint(min_line_length)), int(np.rint(max_line_gap))
    accum = np.zeros((numangle, numrho), np.int64)
    mask = (e != 0)
    mask = mask.copy()
    ys, xs = np.nonzero(mask)  # raster order
    nz = [(int(x), int(y)) for x, y in zip(xs, ys)]
    state = (1 << 64) - 1
    lines = []
    count = len(nz)
    half = (numrho - 1) // 2

    def rbin(n, x, y):
        return int(np.rint(f32(x) * cs[n] + f32(y) * sn[n])) + half

    while count > 0:
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        idx = (state & 0xFFFFFFFF) % count
        j, i = nz[idx]
        nz[idx] = nz[count - 1]
        count -= 1
        if not mask[i, j]:
            continue
        max_val, max_n = threshold - 1, 0
        for n in range(numangle):
            r = rbin(n, j, i)
            accum[n, r] += 1
            if max_val < accum[n, r]:
                max_val, max_n = int(accum[n, r]), n
        if max_val < threshold:
            continue
        a, b = -sn[max_n], cs[max_n]
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag = True
            dx0 = 1 if a > 0 else -1
            dy0 = int(np.rint(b * f32(65536) / abs(a)))
            y0 = (y0 << 16) + (1 << 15)
        else:
            xflag = False
            dy0 = 1 if b > 0 else -1
            dx0 = int(np.rint(a * f32(65536) / abs(b)))
            x0 = (x0 << 16) + (1 << 15)
        ends = [None, None]
        for k in range(2):
            gap, x, y = 0, x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if j1 < 0 or j1 >= width or i1 < 0 or i1 >= height:
                    break
                if mask[i1, j1]:
                    gap = 0
                    ends[k] = (j1, i1)
                else:
                    gap += 1
                    if gap > line_gap:
                        break
                x += dx
                y += dy
        good = abs(ends[1][0] - ends[0][0]) >= line_length or abs(ends[1][1] - ends[0][1]) >= line_length
        for k in range(2):
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> 16) if xflag else (x >> 16, y)
                if mask[i1, j1]:
                    if good:
                        for n in range(numangle):
                            accum[n, rbin(n, j1, i1)] -= 1
                    mask[i1, j1] = False
                if (j1, i1) == ends[k]:
                    break
                x += dx
                y += dy
        if good:
            lines.append((ends[0][0], ends[0][1], ends[1][0], ends[1][1]))
    return np.array(lines, np.int32).reshape(-1, 4)


def line_angles_f32_np(lines):
    """hough.rs:50-68 in numpy float32 (np.arctan2 on float32 calls the same libm atan2f)."""
    l = np.asarray(lines, np.float32).reshape(-1, 4)
    ang = np.arctan2(l[:, 3] - l[:, 1], l[:, 2] - l[:, 0]) * np.float32(180.0) / np.float32(np.pi)
    return np.fmod(ang, np.float32(45.0)).astype(np.float32)
