"""oics::calculate (packages/lib/src/calculate.rs:2-10, :13-23) through the C ABI."""
import ctypes as C

import numpy as np

from ._lib import check, f64p, lib


def get_arithmetic_mean(vec):
    v = np.ascontiguousarray(vec, np.float64)
    out = C.c_double()
    check(lib().omr_get_arithmetic_mean(v.ctypes.data_as(f64p), v.size, C.byref(out)))
    return out.value


def get_standard_deviation(vec):
    v = np.ascontiguousarray(vec, np.float64)
    out = C.c_double()
    check(lib().omr_get_standard_deviation(v.ctypes.data_as(f64p), v.size, C.byref(out)))
    return out.value
