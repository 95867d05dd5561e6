"""oics::omr (packages/lib/src/omr.rs) through the C ABI."""
import ctypes as C

import numpy as np

from ._lib import OmrImageOwned, check, f64p, i32p, lib
from .transfer import _mat, as_image
from .types import ResultStatus


class OmrResult:
    """omr.rs:46-50"""

    def __init__(self, angle, status, candidates):
        self.angle = angle
        self.status = status
        self.candidates = candidates


def get_mat_projection_data(mat):
    """omr.rs:8-39 -> (horizontal[rows], vertical[cols])"""
    a, im = as_image(_mat(mat))
    h, v = np.empty(im.rows, np.float64), np.empty(im.cols, np.float64)
    check(lib().omr_get_mat_projection_data(C.byref(im), h.ctypes.data_as(f64p), v.ctypes.data_as(f64p)))
    return h, v


def get_result_from_projection(src_mat, projection_max_angle, projection_angle_step, projection_max_width,
                               projection_max_height):
    """omr.rs:52-229"""
    a, im = as_image(_mat(src_mat))
    n = C.c_int32()
    A = lib().omr_candidate_count(int(projection_max_angle), float(projection_angle_step), C.byref(n))
    cand = np.zeros(max(A, 1), np.float64)
    angle, status, clen = C.c_double(), C.c_int32(), C.c_int32()
    check(lib().omr_get_result_from_projection(C.byref(im), int(projection_max_angle), float(projection_angle_step),
                                               int(projection_max_width), int(projection_max_height), C.byref(angle),
                                               C.byref(status), cand.ctypes.data_as(f64p), cand.size, C.byref(clen)))
    return OmrResult(angle.value, ResultStatus(status.value), cand[: clen.value].copy())


def select_projection_result(v_sd, h_sd, N, step):
    """omr.rs:147-221 on host arrays"""
    v = np.ascontiguousarray(v_sd, np.float64)
    h = np.ascontiguousarray(h_sd, np.float64)
    cand = np.zeros(max(v.size, 1), np.float64)
    angle, status, clen = C.c_double(), C.c_int32(), C.c_int32()
    check(lib().omr_select_projection_result(v.ctypes.data_as(f64p), h.ctypes.data_as(f64p), v.size, N, float(step),
                                             C.byref(angle), C.byref(status), cand.ctypes.data_as(f64p), cand.size,
                                             C.byref(clen)))
    return OmrResult(angle.value, ResultStatus(status.value), cand[: clen.value].copy())


def get_result_from_edges_detection(src_mat, edges_min_line_length, edges_max_line_gap):
    """omr.rs:231-302"""
    a, im = as_image(_mat(src_mat))
    cap = 1 << 16
    cand = np.zeros(cap, np.float64)
    angle, status, clen = C.c_double(), C.c_int32(), C.c_int32()
    check(lib().omr_get_result_from_edges_detection(C.byref(im), float(edges_min_line_length), float(edges_max_line_gap),
                                                    C.byref(angle), C.byref(status), cand.ctypes.data_as(f64p), cap,
                                                    C.byref(clen)))
    return OmrResult(angle.value, ResultStatus(status.value), cand[: min(clen.value, cap)].copy())


def edges_detection_batch_device(d_scans_ptr, n, scan_stride, rows, cols, channels, step, min_line_length, max_line_gap,
                                 stream=None):
    """get_result_from_edges_detection on n device-resident scans -> (angles, status, n_lines)"""
    angles = np.zeros(n, np.float64)
    status = np.zeros(n, np.int32)
    nl = np.zeros(n, np.int32)
    check(lib().omr_edges_detection_batch_device(d_scans_ptr, n, scan_stride, rows, cols, channels, step,
                                                 float(min_line_length), float(max_line_gap), angles.ctypes.data_as(f64p),
                                                 status.ctypes.data_as(i32p), nl.ctypes.data_as(i32p), stream))
    return angles, status, nl


def hough_set_scans_in_flight(scans):
    """Tuning knob of edges_detection_batch_device (include/omrdeskew.h): scans the sequential Hough stage works on at
    once; 0 = the library's default.  Returns the previous setting."""
    return int(lib().omr_hough_set_scans_in_flight(int(scans)))


def correct_default_decision(projection_result, edges_angle):
    """omr.rs:351-399 -> (rotate_angle, need_check)"""
    c = np.ascontiguousarray(projection_result.candidates, np.float64)
    ang, chk = C.c_double(), C.c_int32()
    lib().omr_correct_default_decision(float(projection_result.angle), int(projection_result.status),
                                       c.ctypes.data_as(f64p), c.size, float(edges_angle), C.byref(ang), C.byref(chk))
    return ang.value, bool(chk.value)


def correct_default(src_mat, projection_max_angle, projection_angle_step, projection_max_width, projection_max_height,
                    hough_min_line_length, hough_max_line_gap, want_image=True):
    """omr.rs:339-448 on a decoded BGR image (imread / imwrite are the caller's) ->
    (rotate_angle, need_check, rotated image or None)"""
    from .hough import _take
    a, im = as_image(_mat(src_mat))
    ang, chk = C.c_double(), C.c_int32()
    owned = OmrImageOwned()
    check(lib().omr_correct_default(C.byref(im), int(projection_max_angle), float(projection_angle_step),
                                    int(projection_max_width), int(projection_max_height), float(hough_min_line_length),
                                    float(hough_max_line_gap), C.byref(ang), C.byref(chk),
                                    C.byref(owned) if want_image else None))
    return ang.value, bool(chk.value), (_take(owned) if want_image else None)


def get_result_from_fourier_transform(src_mat, canny_threshold_weak, canny_threshold_strong, fourier_min_line_length,
                                      fourier_max_line_gap):
    """omr.rs:304-337"""
    a, im = as_image(_mat(src_mat))
    cap = 1 << 16
    cand = np.zeros(cap, np.float64)
    angle, status, clen = C.c_double(), C.c_int32(), C.c_int32()
    check(lib().omr_get_result_from_fourier_transform(C.byref(im), float(canny_threshold_weak),
                                                      float(canny_threshold_strong), float(fourier_min_line_length),
                                                      float(fourier_max_line_gap), C.byref(angle), C.byref(status),
                                                      cand.ctypes.data_as(f64p), cap, C.byref(clen)))
    return OmrResult(angle.value, ResultStatus(status.value), cand[: min(clen.value, cap)].copy())
