"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)


def build(force=False):
    """Compile liboracle.so / liboracle_fast.so with gcc (building the checker is not using it)."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))  # no-op when up to date


def lib(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_get_rotation_matrix_2d.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double, f64p]
        L.orc_get_rotation_matrix_2d.restype = None
        L.orc_invert_affine.argtypes = [f64p, f64p]
        L.orc_invert_affine.restype = None
        L.orc_warp_tables.argtypes = [f64p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
        L.orc_warp_tables.restype = None
        for n in ("orc_warp_affine_nn", "orc_warp_affine_linear"):
            f = getattr(L, n)
            f.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, u8p, C.c_int, C.c_int, C.c_int64, f64p, u8p]
            f.restype = C.c_int
        L.orc_threshold_binary.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, u8p, C.c_int64, C.c_int, C.c_int]
        L.orc_threshold_binary.restype = None
        L.orc_rgb2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, u8p, C.c_int64]
        L.orc_rgb2gray.restype = None
        L.orc_erode_cross3.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, u8p, C.c_int64, C.c_int]
        L.orc_erode_cross3.restype = None
        L.orc_resize_area.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, u8p, C.c_int, C.c_int, C.c_int64]
        L.orc_resize_area.restype = C.c_int
        L.orc_resize_linear.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, u8p, C.c_int, C.c_int, C.c_int64, C.c_int]
        L.orc_resize_linear.restype = C.c_int
        L.orc_arithmetic_mean.argtypes = [f64p, C.c_size_t]
        L.orc_arithmetic_mean.restype = C.c_double
        L.orc_standard_deviation.argtypes = [f64p, C.c_size_t]
        L.orc_standard_deviation.restype = C.c_double
        for n in ("orc_vertical_projection", "orc_horizontal_projection"):
            f = getattr(L, n)
            f.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, f64p]
            f.restype = None
        L.orc_mat_projection_data.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, f64p, f64p]
        L.orc_mat_projection_data.restype = None
        L.orc_projection_standard_deviations.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, f64p, f64p]
        L.orc_projection_standard_deviations.restype = None
        L.orc_rotate_mat_size.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_rotate_mat_size.restype = None
        L.orc_rotate_mat.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_int, u8p,
                                     C.c_int, u8p, C.c_int, C.c_int, C.c_int64]
        L.orc_rotate_mat.restype = C.c_int
        L.orc_candidate_count.argtypes = [C.c_uint16, C.c_double, C.POINTER(C.c_int)]
        L.orc_candidate_count.restype = C.c_int
        L.orc_sweep.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, C.c_uint16, C.c_double, C.c_double, C.c_int,
                                u32p, u32p, f64p, f64p]
        L.orc_sweep.restype = C.c_int
        L.orc_sweep_matrices.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, f64p, C.c_int, C.c_int, u32p, u32p, f64p, f64p]
        L.orc_sweep_matrices.restype = C.c_int
        L.orc_argmax_path1.argtypes = [f64p, f64p, C.c_size_t, u8p]
        L.orc_argmax_path1.restype = C.c_size_t
        L.orc_select_path2.argtypes = [f64p, f64p, C.c_size_t, C.c_int, C.c_double, C.POINTER(C.c_int), f64p,
                                       C.POINTER(C.c_int)]
        L.orc_select_path2.restype = C.c_double
        L.orc_get_angle_with_projections.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint16, C.c_double,
                                                     C.c_double, f64p, C.POINTER(C.c_size_t)]
        L.orc_get_angle_with_projections.restype = C.c_int
        L.orc_get_result_from_projection.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint16, C.c_double,
                                                     C.c_int, C.c_int, f64p, C.POINTER(C.c_int), f64p, C.c_int,
                                                     C.POINTER(C.c_int)]
        L.orc_get_result_from_projection.restype = C.c_int
        # ---- Hough-line path (oracle_hough.c)
        i16p = C.POINTER(C.c_int16)
        ip = C.POINTER(C.c_int)
        L.orc_sobel3_16s.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, i16p, i16p]
        L.orc_sobel3_16s.restype = None
        L.orc_canny.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, u8p, C.c_int64]
        L.orc_canny.restype = C.c_int
        L.orc_hough_trigtab.argtypes = [C.c_double, C.c_double, ip, C.POINTER(C.c_float)]
        L.orc_hough_trigtab.restype = None
        L.orc_hough_lines_p.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_double,
                                        C.c_double, i32p, C.c_int, ip]
        L.orc_hough_lines_p.restype = C.c_int
        L.orc_line_angle_f32.argtypes = [i32p]
        L.orc_line_angle_f32.restype = C.c_float
        L.orc_vote_hough_rs.argtypes = [C.POINTER(C.c_float), C.c_int, f64p]
        L.orc_vote_hough_rs.restype = C.c_int
        L.orc_vote_omr_rs.argtypes = [C.POINTER(C.c_float), C.c_int, f64p, ip, f64p, C.c_int, ip]
        L.orc_vote_omr_rs.restype = C.c_int
        L.orc_get_angle_with_hough.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, f64p, ip]
        L.orc_get_angle_with_hough.restype = C.c_int
        L.orc_get_result_from_edges_detection.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double,
                                                          C.c_double, f64p, ip, f64p, C.c_int, ip, ip]
        L.orc_get_result_from_edges_detection.restype = C.c_int
        L.orc_correct_default_decision.argtypes = [C.c_double, C.c_int, f64p, C.c_int, C.c_double, f64p, ip]
        L.orc_correct_default_decision.restype = None
        _LIBS[name] = L
    return _LIBS[name]


def _u8(a):
    return a.ctypes.data_as(u8p)


def _f64(a):
    return a.ctypes.data_as(f64p)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 2:
        return a, a.shape[0], a.shape[1], 1, a.strides[0]
    return a, a.shape[0], a.shape[1], a.shape[2], a.strides[0]


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)


def get_rotation_matrix_2d(cx, cy, angle, scale):
    M = np.zeros(6, np.float64)
    lib().orc_get_rotation_matrix_2d(cx, cy, angle, scale, _f64(M))
    return M


def rotation_matrices(rows, cols, max_angle, step, scale=1.0):
    """The A forward matrices of the sweep (transfer.rs:473-475 per candidate)."""
    N, A = candidate_count(max_angle, step)
    Ms = np.zeros((A, 6), np.float64)
    cx, cy = np.float32(cols) / np.float32(2.0), np.float32(rows) / np.float32(2.0)
    for i in range(A):
        Ms[i] = get_rotation_matrix_2d(float(cx), float(cy), float(i - N) * step, scale)
    return Ms


def invert_affine(M):
    M = np.ascontiguousarray(M, np.float64)
    out = np.zeros(6, np.float64)
    lib().orc_invert_affine(_f64(M), _f64(out))
    return out


def warp_tables(Minv, dcols, drows, round_delta=512):
    Minv = np.ascontiguousarray(Minv, np.float64)
    ad, bd = np.zeros(dcols, np.int32), np.zeros(dcols, np.int32)
    X0, Y0 = np.zeros(drows, np.int32), np.zeros(drows, np.int32)
    lib().orc_warp_tables(_f64(Minv), dcols, drows, round_delta, ad.ctypes.data_as(i32p), bd.ctypes.data_as(i32p),
                          X0.ctypes.data_as(i32p), Y0.ctypes.data_as(i32p))
    return ad, bd, X0, Y0


def warp_affine(src, M, dsize=None, interp=0, border=(255, 255, 255, 0)):
    a, r, c, cn, st = _img(src)
    dr, dc = (r, c) if dsize is None else dsize
    dst = np.zeros((dr, dc) if a.ndim == 2 else (dr, dc, cn), np.uint8)
    M = np.ascontiguousarray(M, np.float64)
    b = np.array(border, np.uint8)
    f = lib().orc_warp_affine_nn if interp == 0 else lib().orc_warp_affine_linear
    _check(f(_u8(a), r, c, cn, st, _u8(dst), dr, dc, dst.strides[0], _f64(M), _u8(b)))
    return dst


def threshold_binary(gray, thresh=127, maxval=255):
    a, r, c, _, st = _img(gray)
    dst = np.zeros((r, c), np.uint8)
    lib().orc_threshold_binary(_u8(a), r, c, st, _u8(dst), c, thresh, maxval)
    return dst


def rgb2gray(img):
    a, r, c, cn, st = _img(img)
    dst = np.zeros((r, c), np.uint8)
    lib().orc_rgb2gray(_u8(a), r, c, cn, st, _u8(dst), c)
    return dst


def erode_cross3(gray, iterations=3):
    a, r, c, _, st = _img(gray)
    dst = np.zeros((r, c), np.uint8)
    lib().orc_erode_cross3(_u8(a), r, c, st, _u8(dst), c, iterations)
    return dst


def resize_area(img, drows, dcols):
    a, r, c, cn, st = _img(img)
    dst = np.zeros((drows, dcols) if a.ndim == 2 else (drows, dcols, cn), np.uint8)
    _check(lib().orc_resize_area(_u8(a), r, c, cn, st, _u8(dst), drows, dcols, dst.strides[0]))
    return dst


def resize_linear(img, drows, dcols, area_mode=False):
    """resize(INTER_LINEAR), or INTER_AREA's bilinear emulation when an axis enlarges (area_mode)"""
    a, r, c, cn, st = _img(img)
    dst = np.zeros((drows, dcols) if a.ndim == 2 else (drows, dcols, cn), np.uint8)
    _check(lib().orc_resize_linear(_u8(a), r, c, cn, st, _u8(dst), drows, dcols, dst.strides[0], 1 if area_mode else 0))
    return dst


def scale_self(img, scale):
    """transfer.rs:66-91"""
    if scale == 1.0:
        return np.array(img, copy=True)
    r, c = img.shape[:2]
    dc, dr = int(c * scale), int(r * scale)
    return resize_linear(img, dr, dc, False) if scale > 1.0 else resize_area(img, dr, dc)


def shrink_to(img, max_width, max_height):
    """transfer.rs:93-126: never enlarges"""
    r, c = img.shape[:2]
    ws = 1.0 if max_width <= 0 else max_width / c
    hs = 1.0 if max_height <= 0 else max_height / r
    t = ws if ws < hs else hs
    return np.array(img, copy=True) if t >= 1.0 else scale_self(img, t)


def standard_deviation(v):
    v = np.ascontiguousarray(v, np.float64)
    return lib().orc_standard_deviation(_f64(v), v.size)


def arithmetic_mean(v):
    v = np.ascontiguousarray(v, np.float64)
    return lib().orc_arithmetic_mean(_f64(v), v.size)


def vertical_projection(img):
    a, r, c, _, st = _img(img)
    out = np.zeros(c, np.float64)
    lib().orc_vertical_projection(_u8(a), r, c, st, _f64(out))
    return out


def horizontal_projection(img):
    a, r, c, _, st = _img(img)
    out = np.zeros(r, np.float64)
    lib().orc_horizontal_projection(_u8(a), r, c, st, _f64(out))
    return out


def mat_projection_data(img):
    a, r, c, _, st = _img(img)
    h, v = np.zeros(r, np.float64), np.zeros(c, np.float64)
    lib().orc_mat_projection_data(_u8(a), r, c, st, _f64(h), _f64(v))
    return h, v


def projection_standard_deviations(img):
    a, r, c, _, st = _img(img)
    v, h = C.c_double(), C.c_double()
    lib().orc_projection_standard_deviations(_u8(a), r, c, st, C.byref(v), C.byref(h))
    return v.value, h.value


def rotate_mat(src, angle, scale=1.0, interp=0, border=(255, 255, 255, 0), clip=0):
    a, r, c, cn, st = _img(src)
    dr, dc = C.c_int(), C.c_int()
    lib().orc_rotate_mat_size(r, c, angle, clip, C.byref(dr), C.byref(dc))
    dst = np.zeros((dr.value, dc.value) if a.ndim == 2 else (dr.value, dc.value, cn), np.uint8)
    b = np.array(border, np.uint8)
    _check(lib().orc_rotate_mat(_u8(a), r, c, cn, st, angle, scale, interp, _u8(b), clip, _u8(dst), dr.value, dc.value,
                                dst.strides[0]))
    return dst


def candidate_count(max_angle, step):
    n = C.c_int()
    A = lib().orc_candidate_count(max_angle, step, C.byref(n))
    return n.value, A


def sweep(bin_img, max_angle, step, matrix_scale=1.0, threads=1, want_proj=True, fast=False):
    a, r, c, _, st = _img(bin_img)
    _, A = candidate_count(max_angle, step)
    vp = np.zeros((A, c), np.uint32) if want_proj else None
    hp = np.zeros((A, r), np.uint32) if want_proj else None
    vs, hs = np.zeros(A, np.float64), np.zeros(A, np.float64)
    _check(lib(fast).orc_sweep(_u8(a), r, c, st, max_angle, step, matrix_scale, threads,
                               vp.ctypes.data_as(u32p) if want_proj else None,
                               hp.ctypes.data_as(u32p) if want_proj else None, _f64(vs), _f64(hs)))
    return vp, hp, vs, hs


def sweep_matrices(bin_img, Ms, threads=1, want_proj=True, fast=False):
    a, r, c, _, st = _img(bin_img)
    Ms = np.ascontiguousarray(Ms, np.float64).reshape(-1, 6)
    A = Ms.shape[0]
    vp = np.zeros((A, c), np.uint32) if want_proj else None
    hp = np.zeros((A, r), np.uint32) if want_proj else None
    vs, hs = np.zeros(A, np.float64), np.zeros(A, np.float64)
    _check(lib(fast).orc_sweep_matrices(_u8(a), r, c, st, _f64(Ms), A, threads,
                                        vp.ctypes.data_as(u32p) if want_proj else None,
                                        hp.ctypes.data_as(u32p) if want_proj else None, _f64(vs), _f64(hs)))
    return vp, hp, vs, hs


def argmax_path1(v_sd, h_sd):
    v = np.ascontiguousarray(v_sd, np.float64)
    h = np.ascontiguousarray(h_sd, np.float64)
    acc = np.zeros(v.size, np.uint8)
    idx = lib().orc_argmax_path1(_f64(v), _f64(h), v.size, _u8(acc))
    return int(idx), np.nonzero(acc)[0]


def select_path2(v_sd, h_sd, N, step):
    v = np.ascontiguousarray(v_sd, np.float64)
    h = np.ascontiguousarray(h_sd, np.float64)
    cand = np.zeros(max(v.size, 1), np.float64)
    status, n = C.c_int(), C.c_int()
    ang = lib().orc_select_path2(_f64(v), _f64(h), v.size, N, step, C.byref(status), _f64(cand), C.byref(n))
    return ang, status.value, cand[: n.value].copy()


def get_angle_with_projections(img, max_angle, step, resize_scale):
    a, r, c, cn, st = _img(img)
    ang, idx = C.c_double(), C.c_size_t()
    _check(lib().orc_get_angle_with_projections(_u8(a), r, c, cn, st, max_angle, step, resize_scale, C.byref(ang),
                                                C.byref(idx)))
    return ang.value, idx.value


def get_result_from_projection(img, max_angle, step, max_w, max_h):
    a, r, c, cn, st = _img(img)
    _, A = candidate_count(max_angle, step)
    cand = np.zeros(max(A, 1), np.float64)
    ang, status, n = C.c_double(), C.c_int(), C.c_int()
    _check(lib().orc_get_result_from_projection(_u8(a), r, c, cn, st, max_angle, step, max_w, max_h, C.byref(ang),
                                                C.byref(status), _f64(cand), cand.size, C.byref(n)))
    return ang.value, status.value, cand[: n.value].copy()


# ---- Hough-line path (oracle_hough.c) -----------------------------------
def sobel3_16s(img):
    a, rows, cols, cn, step = _img(img)
    dx = np.zeros((rows, cols, cn), np.int16)
    dy = np.zeros((rows, cols, cn), np.int16)
    i16p = C.POINTER(C.c_int16)
    lib().orc_sobel3_16s(_u8(a), rows, cols, cn, step, dx.ctypes.data_as(i16p), dy.ctypes.data_as(i16p))
    return dx, dy


def canny(img, low=50.0, high=150.0, fast=False):
    a, rows, cols, cn, step = _img(img)
    out = np.zeros((rows, cols), np.uint8)
    _check(lib(fast).orc_canny(_u8(a), rows, cols, cn, step, low, high, _u8(out), out.strides[0]))
    return out


def hough_trigtab(theta=np.pi / 180.0, rho=1.0):
    n = C.c_int(0)
    lib().orc_hough_trigtab(theta, rho, C.byref(n), None)
    t = np.zeros(2 * n.value, np.float32)
    lib().orc_hough_trigtab(theta, rho, C.byref(n), t.ctypes.data_as(C.POINTER(C.c_float)))
    return n.value, t


def hough_lines_p(edges, min_line_length, max_line_gap, rho=1.0, theta=np.pi / 180.0, threshold=0, fast=False):
    a, rows, cols, cn, step = _img(edges)
    assert cn == 1
    n = C.c_int(0)
    L = lib(fast)
    _check(L.orc_hough_lines_p(_u8(a), rows, cols, step, rho, theta, threshold, min_line_length, max_line_gap, None, 0,
                               C.byref(n)))
    lines = np.zeros((max(n.value, 1), 4), np.int32)
    _check(L.orc_hough_lines_p(_u8(a), rows, cols, step, rho, theta, threshold, min_line_length, max_line_gap,
                               lines.ctypes.data_as(i32p), n.value, C.byref(n)))
    return lines[: n.value]


def line_angles_f32(lines):
    lines = np.ascontiguousarray(lines, np.int32)
    return np.array([lib().orc_line_angle_f32(lines[i].ctypes.data_as(i32p)) for i in range(len(lines))], np.float32)


def vote_hough_rs(angles):
    angles = np.ascontiguousarray(angles, np.float32)
    out = C.c_double(0)
    _check(lib().orc_vote_hough_rs(angles.ctypes.data_as(C.POINTER(C.c_float)), len(angles), C.byref(out)))
    return out.value


def vote_omr_rs(angles):
    angles = np.ascontiguousarray(angles, np.float32)
    out, st, nc = C.c_double(0), C.c_int(0), C.c_int(0)
    cand = np.zeros(max(len(angles), 1), np.float64)
    _check(lib().orc_vote_omr_rs(angles.ctypes.data_as(C.POINTER(C.c_float)), len(angles), C.byref(out), C.byref(st),
                                 _f64(cand), len(cand), C.byref(nc)))
    return out.value, st.value, cand[: nc.value]


def get_angle_with_hough(gray, min_line_length, max_line_gap, fast=False):
    a, rows, cols, cn, step = _img(gray)
    out, n = C.c_double(0), C.c_int(0)
    _check(lib(fast).orc_get_angle_with_hough(_u8(a), rows, cols, cn, step, min_line_length, max_line_gap, C.byref(out),
                                              C.byref(n)))
    return out.value, n.value


def get_result_from_edges_detection(src, min_line_length, max_line_gap, fast=False):
    a, rows, cols, cn, step = _img(src)
    out, st, nc, nl = C.c_double(0), C.c_int(0), C.c_int(0), C.c_int(0)
    cap = 1 << 16
    cand = np.zeros(cap, np.float64)
    _check(lib(fast).orc_get_result_from_edges_detection(_u8(a), rows, cols, cn, step, min_line_length, max_line_gap,
                                                         C.byref(out), C.byref(st), _f64(cand), cap, C.byref(nc),
                                                         C.byref(nl)))
    return out.value, st.value, cand[: min(nc.value, cap)], nl.value


def correct_default_decision(proj_angle, proj_status, proj_candidates, edges_angle):
    c = np.ascontiguousarray(proj_candidates, np.float64)
    ang, chk = C.c_double(0), C.c_int(0)
    lib().orc_correct_default_decision(proj_angle, proj_status, _f64(c), len(c), edges_angle, C.byref(ang), C.byref(chk))
    return ang.value, bool(chk.value)
