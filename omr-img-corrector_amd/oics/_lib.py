"""ctypes loader of libomrdeskew.so (the C ABI of include/omrdeskew.h).

The product path: every compute call goes through this shared library and from there to the
hand-written HIP kernels.  There is no Python/numpy fallback -- a missing library or a missing
GPU raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libomrdeskew.so")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)


class OmrImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("channels", C.c_int32),
                ("step_bytes", C.c_int64)]


class OmrImageOwned(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("channels", C.c_int32),
                ("step_bytes", C.c_int64)]


class OmrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("omrdeskew error %d: %s" % (code, message))
        self.code = code
        self.message = message


# name -> (restype, argtypes): every symbol include/omrdeskew.h declares
SYMBOLS = {
    "omr_version": (C.c_int, []),
    "omr_device_count": (C.c_int, []),
    "omr_last_error": (C.c_char_p, []),
    "omr_image_free": (None, [C.POINTER(OmrImageOwned)]),
    "omr_get_rotation_matrix_2d": (C.c_int, [C.c_float, C.c_float, C.c_double, C.c_double, f64p]),
    "omr_candidate_count": (C.c_int, [C.c_uint16, C.c_double, i32p]),
    "omr_sweep_matrices": (C.c_int, [C.c_int32, C.c_int32, C.c_uint16, C.c_double, C.c_double, f64p, C.c_int32]),
    "omr_projection_sweep": (C.c_int, [C.POINTER(OmrImage), f64p, C.c_int32, u32p, u32p, f64p, f64p]),
    "omr_sweep_plan_create": (C.c_int, [C.c_int32, C.c_int32, f64p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "omr_sweep_plan_create_angles": (C.c_int, [C.c_int32, C.c_int32, C.c_uint16, C.c_double, C.c_double, C.c_int32,
                                               C.POINTER(C.c_void_p)]),
    "omr_sweep_plan_destroy": (None, [C.c_void_p]),
    "omr_sweep_plan_candidates": (C.c_int, [C.c_void_p]),
    "omr_sweep_plan_run_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "omr_sweep_plan_run": (C.c_int, [C.c_void_p, C.POINTER(OmrImage), C.c_int32, u32p, u32p, f64p, f64p, i32p]),
    "omr_sweep_plan_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "omr_sweep_plan_set_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_sweep_plan_set_kernel": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_sweep_plan_info": (C.c_int, [C.c_void_p, i32p, i32p]),
    "omr_sweep_plan_tables": (C.c_int, [C.c_void_p, C.c_int32, i32p, i32p, i32p, i32p]),
    "omr_slane_strip_program": (C.c_int, [C.c_int32, C.c_int32, f64p, C.c_int32, u32p, u32p, i32p, i32p, i32p, i32p, i32p, i32p]),
    "omr_batch_set_lanes": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_batch_lanes_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), i32p, i32p]),
    "omr_batch_lanes_keep": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_batch_lanes_projections": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, u32p, u32p]),
    "omr_batch_lanes_check_programs": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "omr_batch_create": (C.c_int, [C.c_int32, C.c_int32, C.c_uint16, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_void_p)]),
    "omr_batch_destroy": (None, [C.c_void_p]),
    "omr_batch_run_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "omr_batch_sync": (C.c_int, [C.c_void_p]),
    "omr_batch_set_group": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_batch_info": (C.c_int, [C.c_void_p, i32p, i32p]),
    "omr_batch_set_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_batch_kernel_ms": (C.c_int, [C.c_void_p, f64p, i32p]),
    "omr_batch_deskew_canvas": (C.c_int, [C.c_void_p, i32p, i32p]),
    "omr_batch_deskew_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_uint8, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "omr_call_pool_stats": (C.c_int, [C.c_int32, i32p, i32p, C.POINTER(C.c_int64)]),
    "omr_host_batch_create": (C.c_int, [C.c_int32, C.c_int32, C.c_uint16, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "omr_host_batch_destroy": (None, [C.c_void_p]),
    "omr_host_batch_info": (C.c_int, [C.c_void_p, i32p, i32p, i32p]),
    "omr_host_batch_run": (C.c_int, [C.c_void_p, C.POINTER(OmrImage), C.c_int32, C.c_int32, i32p, f64p, f64p, f64p]),
    "omr_host_batch_set_launch": (C.c_int, [C.c_void_p, C.c_int32]),
    "omr_sweep_batch": (C.c_int, [C.POINTER(OmrImage), C.c_int32, C.c_uint16, C.c_double, C.c_int32, i32p, f64p, f64p,
                                  f64p]),
    "omr_get_angle_with_projections": (C.c_int, [C.POINTER(OmrImage), C.c_uint16, C.c_double, C.c_double, C.c_size_t,
                                                 f64p]),
    "omr_find_target_angle": (C.c_int, [C.c_uint16, C.c_double, C.POINTER(OmrImage), C.c_size_t, f64p]),
    "omr_get_result_from_projection": (C.c_int, [C.POINTER(OmrImage), C.c_uint16, C.c_double, C.c_int32, C.c_int32, f64p,
                                                 i32p, f64p, C.c_int32, i32p]),
    "omr_argmax_projection": (C.c_int, [f64p, f64p, C.c_int32, i32p]),
    "omr_argmax_projection_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "omr_select_projection_result": (C.c_int, [f64p, f64p, C.c_int32, C.c_int32, C.c_double, f64p, i32p, f64p,
                                               C.c_int32, i32p]),
    "omr_threshold_binary": (C.c_int, [C.POINTER(OmrImage), u8p, C.c_int64]),
    "omr_rgb_to_gray": (C.c_int, [C.POINTER(OmrImage), u8p, C.c_int64]),
    "omr_rotate": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, C.c_int32, u8p, C.c_int32,
                             C.POINTER(OmrImageOwned)]),
    "omr_get_vertical_projection": (C.c_int, [C.POINTER(OmrImage), f64p]),
    "omr_get_horizontal_projection": (C.c_int, [C.POINTER(OmrImage), f64p]),
    "omr_get_mat_projection_data": (C.c_int, [C.POINTER(OmrImage), f64p, f64p]),
    "omr_get_projection_standard_deviations": (C.c_int, [C.POINTER(OmrImage), f64p, f64p]),
    "omr_rgb_to_gray_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                         C.c_void_p]),
    "omr_erode3_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "omr_resize_area_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                         C.c_int32, C.c_int32, C.c_void_p]),
    "omr_threshold_binary_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                              C.c_void_p]),
    "omr_rotate_size": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_int32, i32p, i32p]),
    "omr_rotate_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                    C.c_int32, u8p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "omr_canny": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, C.POINTER(OmrImageOwned)]),
    "omr_hough_lines_p": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double, i32p,
                                    C.c_int32, i32p]),
    "omr_get_angle_with_hough": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, f64p]),
    "omr_get_result_from_edges_detection": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, f64p, i32p, f64p,
                                                      C.c_int32, i32p]),
    "omr_edges_detection_batch_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                   C.c_int64, C.c_double, C.c_double, f64p, i32p, i32p, C.c_void_p]),
    "omr_hough_set_scans_in_flight": (C.c_int32, [C.c_int32]),
    "omr_correct_default_decision": (None, [C.c_double, C.c_int32, f64p, C.c_int32, C.c_double, f64p, i32p]),
    "omr_correct_default": (C.c_int, [C.POINTER(OmrImage), C.c_uint16, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                      C.c_double, f64p, i32p, C.POINTER(OmrImageOwned)]),
    "omr_get_fft_image": (C.c_int, [C.POINTER(OmrImage), C.POINTER(OmrImageOwned), C.POINTER(OmrImageOwned)]),
    "omr_fft_image_batch_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                             C.c_void_p, C.c_void_p]),
    "omr_get_angle_with_fft": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, C.c_double, C.c_double, f64p]),
    "omr_get_result_from_fourier_transform": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.c_double, C.c_double,
                                                        C.c_double, f64p, i32p, f64p, C.c_int32, i32p]),
    "omr_scale": (C.c_int, [C.POINTER(OmrImage), C.c_double, C.POINTER(OmrImageOwned)]),
    "omr_shrink_to": (C.c_int, [C.POINTER(OmrImage), C.c_int32, C.c_int32, C.POINTER(OmrImageOwned)]),
    "omr_resize": (C.c_int, [C.POINTER(OmrImage), C.c_int32, C.c_int32, C.POINTER(OmrImageOwned)]),
    "omr_get_arithmetic_mean": (C.c_int, [f64p, C.c_size_t, f64p]),
    "omr_get_standard_deviation": (C.c_int, [f64p, C.c_size_t, f64p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libomrdeskew.so is not built (%s); run __graft_entry__.build() -- there is no "
                              "fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise OmrError(rc, lib().omr_last_error().decode("utf-8", "replace"))
    return rc
