"""oics::hough (packages/lib/src/hough.rs) through the C ABI, plus the two OpenCV calls it is made of."""
import ctypes as C

import numpy as np

from ._lib import OmrImageOwned, check, i32p, lib
from .transfer import _mat, as_image


def _take(owned):
    n = owned.rows * owned.step_bytes
    out = np.ctypeslib.as_array(C.cast(owned.data, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    out = out.reshape(owned.rows, owned.step_bytes)[:, : owned.cols * owned.channels]
    out = out.reshape(owned.rows, owned.cols) if owned.channels == 1 else out.reshape(owned.rows, owned.cols, owned.channels)
    lib().omr_image_free(C.byref(owned))
    return np.ascontiguousarray(out)


def canny(src, low_thresh=50.0, high_thresh=150.0):
    """imgproc::canny(src, &mut edges, low, high, 3, false) -- hough.rs:27, omr.rs:239"""
    a, im = as_image(_mat(src))
    owned = OmrImageOwned()
    check(lib().omr_canny(C.byref(im), float(low_thresh), float(high_thresh), C.byref(owned)))
    return _take(owned)


def hough_lines_p(edges, rho, theta, threshold, min_line_length, max_line_gap):
    """imgproc::hough_lines_p -- hough.rs:31-43.  Returns int32 [n, 4] (x0, y0, x1, y1)."""
    a, im = as_image(_mat(edges))
    n = C.c_int32(0)
    cap = 4096
    while True:
        lines = np.zeros((cap, 4), np.int32)
        check(lib().omr_hough_lines_p(C.byref(im), float(rho), float(theta), int(threshold), float(min_line_length),
                                      float(max_line_gap), lines.ctypes.data_as(i32p), cap, C.byref(n)))
        if n.value <= cap:
            return lines[: n.value].copy()
        cap = n.value


def get_angle_with_hough(gray_tm, min_line_length, max_line_gap):
    """hough.rs:17-100 without the debug picture (file_name / edge_image_output_dir stay host-side)."""
    a, im = as_image(_mat(gray_tm))
    out = C.c_double()
    check(lib().omr_get_angle_with_hough(C.byref(im), float(min_line_length), float(max_line_gap), C.byref(out)))
    return out.value
