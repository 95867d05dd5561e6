"""oics::projection (packages/lib/src/projection.rs) through the C ABI."""
import ctypes as C

import numpy as np

from ._lib import check, f64p, i32p, lib, u32p
from .transfer import _mat, as_image


def get_angle_with_projections(src_img, max_angle, step, resize_scale, threads):
    """projection.rs:17-23: (&TransformableMatrix, u16, f64, f64, usize) -> f64"""
    a, im = as_image(_mat(src_img))
    out = C.c_double()
    check(lib().omr_get_angle_with_projections(C.byref(im), int(max_angle), float(step), float(resize_scale),
                                               int(threads), C.byref(out)))
    return out.value


def find_target_angle(max_angle, step, thresh, threads):
    """app/src-tauri/src/test.rs:83-178"""
    a, im = as_image(_mat(thresh))
    out = C.c_double()
    check(lib().omr_find_target_angle(int(max_angle), float(step), C.byref(im), int(threads), C.byref(out)))
    return out.value


def candidate_count(max_angle, step):
    n = C.c_int32()
    A = lib().omr_candidate_count(int(max_angle), float(step), C.byref(n))
    return n.value, A


def sweep_matrices(rows, cols, max_angle, step, scale=1.0):
    _, A = candidate_count(max_angle, step)
    M = np.zeros((max(A, 1), 6), np.float64)
    check(lib().omr_sweep_matrices(rows, cols, int(max_angle), float(step), float(scale), M.ctypes.data_as(f64p), A))
    return M[:A]


def projection_sweep(bin_img, fwd_M, want_proj=True):
    """The hot loop (projection.rs:47-65 / omr.rs:153-180) for explicit matrices, host buffers."""
    a, im = as_image(_mat(bin_img))
    M = np.ascontiguousarray(fwd_M, np.float64).reshape(-1, 6)
    A = M.shape[0]
    vp = np.zeros((A, im.cols), np.uint32) if want_proj else None
    hp = np.zeros((A, im.rows), np.uint32) if want_proj else None
    vs, hs = np.zeros(A), np.zeros(A)
    check(lib().omr_projection_sweep(C.byref(im), M.ctypes.data_as(f64p), A,
                                     vp.ctypes.data_as(u32p) if want_proj else None,
                                     hp.ctypes.data_as(u32p) if want_proj else None,
                                     vs.ctypes.data_as(f64p), hs.ctypes.data_as(f64p)))
    return vp, hp, vs, hs


def argmax_projection(v_sd, h_sd):
    v = np.ascontiguousarray(v_sd, np.float64)
    h = np.ascontiguousarray(h_sd, np.float64)
    idx = C.c_int32()
    check(lib().omr_argmax_projection(v.ctypes.data_as(f64p), h.ctypes.data_as(f64p), v.size, C.byref(idx)))
    return idx.value


class SweepPlan:
    """Resident form (omr_sweep_plan_*): tables and scratch live on the device across scans."""

    def __init__(self, rows, cols, max_angle=None, step=None, scale=1.0, matrices=None, device=0):
        self.handle = C.c_void_p()
        self.rows, self.cols = rows, cols
        if matrices is not None:
            M = np.ascontiguousarray(matrices, np.float64).reshape(-1, 6)
            check(lib().omr_sweep_plan_create(rows, cols, M.ctypes.data_as(f64p), M.shape[0], device,
                                              C.byref(self.handle)))
        else:
            check(lib().omr_sweep_plan_create_angles(rows, cols, int(max_angle), float(step), float(scale), device,
                                                     C.byref(self.handle)))
        self.A = lib().omr_sweep_plan_candidates(self.handle)

    def close(self):
        if self.handle:
            lib().omr_sweep_plan_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_kernel(self, which):
        check(lib().omr_sweep_plan_set_kernel(self.handle, which))

    def set_timing(self, on):
        check(lib().omr_sweep_plan_set_timing(self.handle, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float()
        check(lib().omr_sweep_plan_last_kernel_ms(self.handle, C.byref(ms)))
        return ms.value

    def info(self):
        """(candidates on the run-merging kernel, candidates on the gather kernels)"""
        r, g = C.c_int32(), C.c_int32()
        check(lib().omr_sweep_plan_info(self.handle, C.byref(r), C.byref(g)))
        return r.value, g.value

    def tables(self, a):
        ad, bd = np.zeros(self.cols, np.int32), np.zeros(self.cols, np.int32)
        X0, Y0 = np.zeros(self.rows, np.int32), np.zeros(self.rows, np.int32)
        check(lib().omr_sweep_plan_tables(self.handle, a, ad.ctypes.data_as(i32p), bd.ctypes.data_as(i32p),
                                          X0.ctypes.data_as(i32p), Y0.ctypes.data_as(i32p)))
        return ad, bd, X0, Y0

    def run(self, img, black_max=0, want_proj=True):
        a, im = as_image(_mat(img))
        vp = np.zeros((self.A, self.cols), np.uint32) if want_proj else None
        hp = np.zeros((self.A, self.rows), np.uint32) if want_proj else None
        vs, hs = np.zeros(self.A), np.zeros(self.A)
        best = C.c_int32()
        check(lib().omr_sweep_plan_run(self.handle, C.byref(im), black_max,
                                       vp.ctypes.data_as(u32p) if want_proj else None,
                                       hp.ctypes.data_as(u32p) if want_proj else None,
                                       vs.ctypes.data_as(f64p), hs.ctypes.data_as(f64p), C.byref(best)))
        return vp, hp, vs, hs, best.value

    def run_device(self, d_img_ptr, step_bytes, black_max, stream, d_vproj, d_hproj, d_v_sd, d_h_sd, d_best):
        check(lib().omr_sweep_plan_run_device(self.handle, d_img_ptr, step_bytes, black_max, stream, d_vproj, d_hproj,
                                              d_v_sd, d_h_sd, d_best))


class Batch:
    """omr_batch_*: n device-resident scans of one shape, round-robin over internal streams."""

    def __init__(self, rows, cols, max_angle, step, scale=1.0, device=0, n_streams=2):
        self.handle = C.c_void_p()
        check(lib().omr_batch_create(rows, cols, int(max_angle), float(step), float(scale), device, n_streams,
                                     C.byref(self.handle)))
        self.N, self.A = candidate_count(max_angle, step)
        self.step = step

    def close(self):
        if self.handle:
            lib().omr_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def run_device(self, d_scans, scan_stride, step_bytes, n, black_max, d_best, d_v_sd=None, d_h_sd=None):
        check(lib().omr_batch_run_device(self.handle, d_scans, scan_stride, step_bytes, n, black_max, d_best, d_v_sd,
                                         d_h_sd))

    def deskew_canvas(self):
        """(rows, cols) every output slot of deskew_device must hold: the largest CONTAIN canvas of the candidates"""
        r, c = C.c_int32(), C.c_int32()
        check(lib().omr_batch_deskew_canvas(self.handle, C.byref(r), C.byref(c)))
        return r.value, c.value

    def deskew_device(self, d_scans, scan_stride, step_bytes, n, black_max, interp, border, d_out, out_stride, out_step,
                      d_out_size=None, d_best=None):
        """omr.rs:339-452 for a batch: sweep -> arg-max -> CONTAIN warp by the detected angle, all on the device"""
        check(lib().omr_batch_deskew_device(self.handle, d_scans, scan_stride, step_bytes, n, black_max, interp, border,
                                            d_out, out_stride, out_step, d_out_size, d_best))

    def sync(self):
        check(lib().omr_batch_sync(self.handle))

    def set_group(self, scans_per_launch):
        """Scans carried by one launch of each kernel (amortises the launch-to-launch cost)."""
        check(lib().omr_batch_set_group(self.handle, int(scans_per_launch)))

    def set_lanes(self, max_scans_per_launch):
        """Scan-lane sweep: up to this many scans per launch, 64 scans per wavefront (0 = back to the run-merging path)."""
        check(lib().omr_batch_set_lanes(self.handle, int(max_scans_per_launch)))

    def lanes_keep(self, on=True):
        """Inspection: launches leave their row counts in place for lanes_projections()."""
        check(lib().omr_batch_lanes_keep(self.handle, 1 if on else 0))

    def lanes_program_bytes(self):
        b, t, n = C.c_int64(), C.c_int32(), C.c_int32()
        check(lib().omr_batch_lanes_info(self.handle, C.byref(b), C.byref(t), C.byref(n)))
        return b.value

    def lanes_check_programs(self):
        """(dwords of programs, dwords that differ from the host generator's): the device-built plan against the
        reference implementation (tests, inspection)."""
        n, d = C.c_int64(), C.c_int64()
        check(lib().omr_batch_lanes_check_programs(self.handle, C.byref(n), C.byref(d)))
        return n.value, d.value

    def lanes_projections(self, scan, a, rows, cols, scratch_set=0):
        """(vproj, hproj) of one scan / candidate as the last scan-lane launch left them (tests, inspection)."""
        vp, hp = np.zeros(cols, np.uint32), np.zeros(rows, np.uint32)
        check(lib().omr_batch_lanes_projections(self.handle, scratch_set, scan, a, vp.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                hp.ctypes.data_as(C.POINTER(C.c_uint32))))
        return vp, hp

    def info(self):
        r, g = C.c_int32(), C.c_int32()
        check(lib().omr_batch_info(self.handle, C.byref(r), C.byref(g)))
        return r.value, g.value

    def set_timing(self, on):
        check(lib().omr_batch_set_timing(self.handle, 1 if on else 0))

    def kernel_ms(self):
        s, n = C.c_double(), C.c_int32()
        check(lib().omr_batch_kernel_ms(self.handle, C.byref(s), C.byref(n)))
        return s.value, n.value


class HostBatch:
    """omr_host_batch: repeated batches that start in host memory (plan, pinned ring and device stages made once)."""

    def __init__(self, rows, cols, max_angle, step, max_scans, n_devices=0):
        self.handle = C.c_void_p()
        self.rows, self.cols = rows, cols
        _, self.A = candidate_count(max_angle, step)
        check(lib().omr_host_batch_create(rows, cols, int(max_angle), float(step), n_devices, int(max_scans),
                                          C.byref(self.handle)))

    def info(self):
        nd, spl, sl = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().omr_host_batch_info(self.handle, C.byref(nd), C.byref(spl), C.byref(sl)))
        return nd.value, spl.value, bool(sl.value)

    PAGEABLE, PINNED, PACKED = 0, 1, 2  # omr_host_batch_run's transfer modes (include/omrdeskew.h)

    def set_launch(self, scans_per_launch):
        """scans per sweep launch (scan-lane contexts: multiples of 64; default 64)"""
        check(lib().omr_host_batch_set_launch(self.handle, int(scans_per_launch)))

    def run(self, scans, pinned=False, want_sd=False, packed=False):
        """scans: list of 2-D u8 arrays (pixels == 0 black).  pinned=True: the arrays live in page-locked memory;
        packed=True: the context's copier threads pack them to 1 bit per pixel and 1/8 of the bytes are uploaded."""
        from ._lib import OmrImage
        keep, arr = [], (OmrImage * len(scans))()
        for i, s in enumerate(scans):
            a, im = as_image(_mat(s))
            keep.append(a)
            arr[i] = im
        n = len(scans)
        best = np.zeros(n, np.int32)
        ang = np.zeros(n, np.float64)
        vs = np.zeros((n, self.A)) if want_sd else None
        hs = np.zeros((n, self.A)) if want_sd else None
        check(lib().omr_host_batch_run(self.handle, arr, n, 2 if packed else 1 if pinned else 0, best.ctypes.data_as(i32p),
                                       ang.ctypes.data_as(f64p), vs.ctypes.data_as(f64p) if want_sd else None,
                                       hs.ctypes.data_as(f64p) if want_sd else None))
        return best, ang, vs, hs

    def close(self):
        if self.handle:
            lib().omr_host_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def sweep_batch(scans, max_angle, step, n_devices=0, want_sd=False):
    """omr_sweep_batch: host images, scan i -> device i % n_devices, host-side gather."""
    from ._lib import OmrImage
    keep, arr = [], (OmrImage * len(scans))()
    for i, s in enumerate(scans):
        a, im = as_image(_mat(s))
        keep.append(a)
        arr[i] = im
    n = len(scans)
    _, A = candidate_count(max_angle, step)
    best = np.zeros(n, np.int32)
    ang = np.zeros(n, np.float64)
    vs = np.zeros((n, A)) if want_sd else None
    hs = np.zeros((n, A)) if want_sd else None
    check(lib().omr_sweep_batch(arr, n, int(max_angle), float(step), n_devices, best.ctypes.data_as(i32p),
                                ang.ctypes.data_as(f64p), vs.ctypes.data_as(f64p) if want_sd else None,
                                hs.ctypes.data_as(f64p) if want_sd else None))
    return best, ang, vs, hs
