"""oics::transfer (packages/lib/src/transfer.rs) through the C ABI.

`TransformableMatrix` wraps a numpy u8 array the way the reference wraps a cv::Mat
(transfer.rs:16-18); every helper returns a fresh object, like the reference's.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OmrImage, OmrImageOwned, check, f64p, lib, u8p
from .types import RotateClipStrategy

INTER_NEAREST = 0  # == imgproc::WARP_POLAR_LINEAR's numeric value, what the reference passes
INTER_LINEAR = 1


def as_image(a):
    """numpy [rows, cols] or [rows, cols, cn] u8 -> (keep-alive array, OmrImage)."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 2:
        rows, cols, cn = a.shape[0], a.shape[1], 1
    elif a.ndim == 3:
        rows, cols, cn = a.shape
    else:
        raise ValueError("image must be 2-D or 3-D")
    return a, OmrImage(a.ctypes.data, rows, cols, cn, a.strides[0] if rows else cols * cn)


class TransformableMatrix:
    def __init__(self, matrix):
        self.matrix = np.ascontiguousarray(matrix, dtype=np.uint8)

    @classmethod
    def from_matrix(cls, mat):
        return cls(np.array(mat, dtype=np.uint8, copy=True))  # transfer.rs:55-59 deep clone

    def get_mat(self):
        return self.matrix

    def clone(self):
        return TransformableMatrix.from_matrix(self.matrix)

    # the three in-place resizers return self like the reference's `&mut Self` chain (transfer.rs:66-145)
    def scale_self(self, scale):
        """transfer.rs:66-91: INTER_LINEAR when scale > 1, INTER_AREA otherwise, size truncated `as i32`"""
        if scale == 1.0:
            return self
        self.matrix = _owned_call(lib().omr_scale, self.matrix, C.c_double(float(scale)))
        return self

    def shrink_to(self, max_width, max_height):
        """transfer.rs:93-126: never enlarges"""
        self.matrix = _owned_call(lib().omr_shrink_to, self.matrix, C.c_int32(int(max_width)), C.c_int32(int(max_height)))
        return self

    def resize_self(self, width, height):
        """transfer.rs:128-145: INTER_AREA to (width, height)"""
        self.matrix = _owned_call(lib().omr_resize, self.matrix, C.c_int32(int(width)), C.c_int32(int(height)))
        return self


def _take_owned(out):
    try:
        shape = (out.rows, out.cols) if out.channels == 1 else (out.rows, out.cols, out.channels)
        n = out.rows * out.step_bytes
        return np.frombuffer((C.c_uint8 * n).from_address(out.data), np.uint8).reshape(shape).copy()
    finally:
        lib().omr_image_free(C.byref(out))


def _owned_call(fn, mat, *args):
    a, im = as_image(mat)
    out = OmrImageOwned()
    check(fn(C.byref(im), *args, C.byref(out)))
    return _take_owned(out)


def _mat(src):
    return src.matrix if isinstance(src, TransformableMatrix) else np.asarray(src)


def transfer_rgb_image_to_gray_image(src):
    """transfer.rs:283-290"""
    a, im = as_image(_mat(src))
    out = np.empty((im.rows, im.cols), np.uint8)
    check(lib().omr_rgb_to_gray(C.byref(im), out.ctypes.data_as(u8p), out.strides[0]))
    return TransformableMatrix(out)


def transfer_gray_image_to_thresh_binary(src):
    """transfer.rs:294-301"""
    a, im = as_image(_mat(src))
    out = np.empty((im.rows, im.cols), np.uint8)
    check(lib().omr_threshold_binary(C.byref(im), out.ctypes.data_as(u8p), out.strides[0]))
    return TransformableMatrix(out)


def get_horizontal_projection(src):
    """transfer.rs:305-333"""
    a, im = as_image(_mat(src))
    out = np.empty(im.rows, np.float64)
    check(lib().omr_get_horizontal_projection(C.byref(im), out.ctypes.data_as(f64p)))
    return out


def get_vertical_projection(src):
    """transfer.rs:380-405"""
    a, im = as_image(_mat(src))
    out = np.empty(im.cols, np.float64)
    check(lib().omr_get_vertical_projection(C.byref(im), out.ctypes.data_as(f64p)))
    return out


def get_projection_standard_deviations(src):
    """transfer.rs:527-536 -> (vertical sd, horizontal sd)"""
    a, im = as_image(_mat(src))
    v, h = C.c_double(), C.c_double()
    check(lib().omr_get_projection_standard_deviations(C.byref(im), C.byref(v), C.byref(h)))
    return v.value, h.value


def rotate_mat(src, angle, scale, flags, border_mode=0, border_value=(255.0, 255.0, 255.0, 0.0),
               clip_strategy=RotateClipStrategy.DEFAULT):
    """transfer.rs:459-523.  border_mode must be BORDER_CONSTANT (0), the only one the reference uses."""
    if border_mode != 0:
        raise _lib.OmrError(-213, "only BORDER_CONSTANT is implemented")
    a, im = as_image(_mat(src))
    b = np.array([int(v) for v in border_value], np.uint8)
    out = OmrImageOwned()
    check(lib().omr_rotate(C.byref(im), float(angle), float(scale), int(flags), b.ctypes.data_as(u8p),
                           int(clip_strategy), C.byref(out)))
    try:
        shape = (out.rows, out.cols) if out.channels == 1 else (out.rows, out.cols, out.channels)
        n = out.rows * out.step_bytes
        arr = np.frombuffer((C.c_uint8 * n).from_address(out.data), np.uint8).reshape(shape).copy()
    finally:
        lib().omr_image_free(C.byref(out))
    return TransformableMatrix(arr)
