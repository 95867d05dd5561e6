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
