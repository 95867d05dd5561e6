"""oics::fft (packages/lib/src/fft.rs) through the C ABI."""
import ctypes as C

from ._lib import OmrImageOwned, check, lib
from .hough import _take
from .transfer import _mat, as_image


def get_fft_image(gray_tm):
    """fft.rs:124-141 -> (magnitude_image, magnitude_log_image), uint8"""
    a, im = as_image(_mat(gray_tm))
    m, lg = OmrImageOwned(), OmrImageOwned()
    check(lib().omr_get_fft_image(C.byref(im), C.byref(m), C.byref(lg)))
    return _take(m), _take(lg)


def fft_image_batch_device(d_scans_ptr, n, scan_stride, rows, cols, step, d_out_ptr, stream=None):
    """magnitude_log pictures of n device-resident scans (config 5)"""
    check(lib().omr_fft_image_batch_device(d_scans_ptr, n, scan_stride, rows, cols, step, d_out_ptr, stream))


def get_angle_with_fft(gray_tm, canny_threshold_1, canny_threshold_2, min_line_length, max_line_gap):
    """fft.rs:145-256 without the debug picture"""
    a, im = as_image(_mat(gray_tm))
    out = C.c_double()
    check(lib().omr_get_angle_with_fft(C.byref(im), float(canny_threshold_1), float(canny_threshold_2),
                                       float(min_line_length), float(max_line_gap), C.byref(out)))
    return out.value
