"""oics::omr (packages/lib/src/omr.rs) through the C ABI."""
import ctypes as C

import numpy as np

from ._lib import check, f64p, lib
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
