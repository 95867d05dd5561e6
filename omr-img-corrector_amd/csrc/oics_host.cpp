// oics_host.cpp -- host side of crate `oics` above the device engine: the drivers and helpers
// with the reference's names, argument meaning and error behaviour, exported through the C ABI.
// (The reference's host language is Rust; there is no rustc in this image, so the host layer is
// C++ -- INTEGRATION.md shows the Rust `extern "C"` shim that binds it.)
//
//   oics::projection::get_angle_with_projections   packages/lib/src/projection.rs:17-194
//   find_target_angle                              packages/app/src-tauri/src/test.rs:83-178
//   oics::omr::get_result_from_projection          packages/lib/src/omr.rs:52-229
//   oics::transfer::*  / oics::calculate::*        packages/lib/src/transfer.rs, calculate.rs
//
// Every image operation runs on the GPU; the host only sequences launches, builds the 2x3
// matrices / resize tap tables (a few hundred doubles) and applies the arg-max policy to the A
// scores.  No CPU fallback exists: without a HIP device every entry point returns -217.
#include <float.h>
#include <math.h>
#include <string.h>

#include <list>
#include <memory>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/omrdeskew.h"
#include "engine.hpp"

using namespace omr;

namespace {

int current_device(int *dev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(OMR_ERR_GPU, "no usable HIP device (there is no CPU fallback)");
    OMR_HIP(hipGetDevice(dev));
    return OMR_OK;
}

int check_image(const omr_image *im, bool need_c1)
{
    if (!im || !im->data) return fail(OMR_ERR_BADARG, "null image");
    if (im->rows <= 0 || im->cols <= 0) return fail(OMR_ERR_ASSERT, "empty image");
    if (im->rows >= 32767 || im->cols >= 32767) return fail(OMR_ERR_ASSERT, "image dimension >= SHRT_MAX");
    if (im->channels < 1 || im->channels > 4) return fail(OMR_ERR_ASSERT, "unsupported channel count %d", im->channels);
    if (need_c1 && im->channels != 1) return fail(OMR_ERR_ASSERT, "expected a 1-channel image, got %d", im->channels);
    if (im->step_bytes < (int64_t)im->cols * im->channels) return fail(OMR_ERR_BADARG, "step_bytes too small");
    return OMR_OK;
}

// device image (tightly packed rows)
struct DevImage {
    DevBuf buf;
    int rows = 0, cols = 0, cn = 1;
    int64_t step() const { return (int64_t)cols * cn; }
    uint8_t *ptr() const { return buf.as<uint8_t>(); }
    int alloc(int r, int c, int ch)
    {
        rows = r;
        cols = c;
        cn = ch;
        OMR_HIP(buf.alloc((size_t)r * c * ch));
        return OMR_OK;
    }
    int upload(const omr_image *im, hipStream_t s)
    {
        int rc = alloc(im->rows, im->cols, im->channels);
        if (rc) return rc;
        // packed rows travel as one linear copy: the 2-D path degrades to row-by-row DMA for widths
        // that are not a multiple of 4 bytes (26 ms instead of 2 ms for a 2677-wide CONTAIN canvas)
        if (im->step_bytes == step())
            OMR_HIP(hipMemcpyAsync(buf.p, im->data, (size_t)step() * rows, hipMemcpyHostToDevice, s));
        else
            OMR_HIP(hipMemcpy2DAsync(buf.p, (size_t)step(), im->data, (size_t)im->step_bytes, (size_t)step(),
                                     (size_t)rows, hipMemcpyHostToDevice, s));
        return OMR_OK;
    }
    int download(uint8_t *dst, int64_t dstep, hipStream_t s) const
    {
        // through the calling thread's pinned staging buffer (engine.cpp: a direct copy into pageable memory is
        // an order of magnitude slower for image-sized results)
        return staged_d2h_2d(dst, (size_t)dstep, buf.p, (size_t)step(), (size_t)rows, s);
    }
};

struct Stream {
    hipStream_t s = nullptr;
    CallSlot *slot = nullptr;         // leased for the call: stream + pinned staging (engine.cpp)
    std::unique_ptr<PoolScope> pool;  // the call's device buffers come from / return to the block cache
    ~Stream()
    {
        pool.reset();  // drains the stream for the buffers it returns
        return_call_slot(slot);
    }
    int create()
    {
        int rc = lease_call_slot(&slot);
        if (rc) return rc;
        s = slot->stream;
        pool.reset(new PoolScope(s));
        return OMR_OK;
    }
};

// ---- plan cache: the app calls the drivers over and over with one parameter set --------------
struct PlanKey {
    int device, rows, cols;
    uint16_t max_angle;
    double step, scale;
    bool operator==(const PlanKey &o) const
    {
        return device == o.device && rows == o.rows && cols == o.cols && max_angle == o.max_angle && step == o.step &&
               scale == o.scale;
    }
};
std::mutex g_cache_mu;
std::list<std::pair<PlanKey, std::shared_ptr<omr_sweep_plan>>> g_cache;

int get_plan(const PlanKey &k, std::shared_ptr<omr_sweep_plan> *out)
{
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
            if (it->first == k) {
                *out = it->second;
                g_cache.splice(g_cache.begin(), g_cache, it);
                return OMR_OK;
            }
    }
    omr_sweep_plan *raw = nullptr;
    int rc = omr_sweep_plan_create_angles(k.rows, k.cols, k.max_angle, k.step, k.scale, k.device, &raw);
    if (rc) return rc;
    std::shared_ptr<omr_sweep_plan> sp(raw);
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.emplace_front(k, sp);
    while (g_cache.size() > 8) g_cache.pop_back();
    *out = sp;
    return OMR_OK;
}

// Sweep a device-resident 1-channel image and fetch the A scores.
int sweep_scores(const DevImage &img, int black_max, uint16_t max_angle, double step, double scale, int device,
                 hipStream_t s, std::vector<double> *v_sd, std::vector<double> *h_sd, int *N_out)
{
    int N, A = candidate_count(max_angle, step, &N);
    if (N_out) *N_out = N;
    v_sd->assign((size_t)(A > 0 ? A : 0), 0.0);
    h_sd->assign((size_t)(A > 0 ? A : 0), 0.0);
    if (A <= 0) return OMR_OK;
    std::shared_ptr<omr_sweep_plan> plan;
    int rc = get_plan(PlanKey{device, img.rows, img.cols, max_angle, step, scale}, &plan);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(plan->mu);
    rc = enqueue_sweep(plan->tables, plan->scratch, plan->kernel_sel, img.ptr(), img.step(), black_max, s, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    OMR_HIP(hipMemcpyAsync(v_sd->data(), plan->scratch.vsd.p, sizeof(double) * (size_t)A, hipMemcpyDeviceToHost, s));
    OMR_HIP(hipMemcpyAsync(h_sd->data(), plan->scratch.hsd.p, sizeof(double) * (size_t)A, hipMemcpyDeviceToHost, s));
    OMR_HIP(hipStreamSynchronize(s));
    return OMR_OK;
}

inline int cv_floor(double v)
{
    int i = (int)v;
    return i - (i > v);
}
inline int cv_ceil(double v)
{
    int i = (int)v;
    return i + (i < v);
}

// OpenCV computeResizeAreaTab (resize.cpp), grouped per destination index (CSR offsets).
void area_tab(int ssize, int dsize, int cn, double scale, std::vector<AreaTap> *tab, std::vector<int32_t> *ofs)
{
    tab->clear();
    ofs->assign((size_t)dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        (*ofs)[dx] = (int32_t)tab->size();
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) tab->push_back(AreaTap{(sx1 - 1) * cn, dx * cn, (float)((sx1 - fsx1) / cellWidth)});
        for (int sx = sx1; sx < sx2; sx++) tab->push_back(AreaTap{sx * cn, dx * cn, (float)(1.0 / cellWidth)});
        if (fsx2 - sx2 > 1e-3) {
            double m = fsx2 - sx2 < 1. ? fsx2 - sx2 : 1.;
            m = m < cellWidth ? m : cellWidth;
            tab->push_back(AreaTap{sx2 * cn, dx * cn, (float)(m / cellWidth)});
        }
    }
    (*ofs)[dsize] = (int32_t)tab->size();
}

// resize(src, dsize, interp) on the device for the two flags the reference passes (transfer.rs:66-91
// scale_self: INTER_LINEAR when enlarging, INTER_AREA otherwise; transfer.rs:128-145 resize_self and
// omr.rs:114-126: INTER_AREA whatever the direction).  OpenCV 4.6.0 resize.cpp dispatch:
//   same size                         -> copy
//   INTER_AREA, both axes shrink      -> resizeAreaFast_ (integer factors) / resizeArea_ (tap tables)
//   INTER_AREA, an axis enlarges      -> the bilinear kernel with area-mode coefficients (quirk B7)
//   INTER_LINEAR                      -> the bilinear kernel (exact 2x shrink is re-routed to INTER_AREA)
int resize_ptr(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst, int64_t dstep,
               int drows, int dcols, int interp, hipStream_t s)
{
    if (drows <= 0 || dcols <= 0) return fail(OMR_ERR_ASSERT, "resize to an empty size");
    if (interp != OMR_INTER_AREA && interp != OMR_INTER_LINEAR)
        return fail(OMR_ERR_NOTIMPL, "resize interpolation flag %d is not implemented", interp);
    if (drows == srows && dcols == scols) {
        OMR_HIP(hipMemcpy2DAsync(d_dst, (size_t)dstep, d_src, (size_t)sstep, (size_t)scols * cn, (size_t)srows,
                                 hipMemcpyDeviceToDevice, s));
        return OMR_OK;
    }
    double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    bool is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (interp == OMR_INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interp = OMR_INTER_AREA;
    if (!(interp == OMR_INTER_AREA && scale_x >= 1 && scale_y >= 1)) {
        OMR_HIP(launch_resize_linear(d_src, sstep, srows, scols, cn, d_dst, dstep, drows, dcols,
                                     interp == OMR_INTER_AREA, s));
        return OMR_OK;
    }
    if (is_area_fast) {
        OMR_HIP(launch_resize_area_int_fast(d_src, sstep, srows, scols, cn, d_dst, dstep, drows, dcols, iscale_x,
                                            iscale_y, s));
        return OMR_OK;
    }
    std::vector<AreaTap> xt, yt;
    std::vector<int32_t> xo, yo;
    area_tab(scols, dcols, cn, scale_x, &xt, &xo);
    area_tab(srows, drows, 1, scale_y, &yt, &yo);
    DevBuf dxt, dxo, dyt, dyo;
    OMR_HIP(dxt.alloc(sizeof(AreaTap) * xt.size()));
    OMR_HIP(dxo.alloc(sizeof(int32_t) * xo.size()));
    OMR_HIP(dyt.alloc(sizeof(AreaTap) * yt.size()));
    OMR_HIP(dyo.alloc(sizeof(int32_t) * yo.size()));
    OMR_HIP(hipMemcpyAsync(dxt.p, xt.data(), sizeof(AreaTap) * xt.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(hipMemcpyAsync(dxo.p, xo.data(), sizeof(int32_t) * xo.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(hipMemcpyAsync(dyt.p, yt.data(), sizeof(AreaTap) * yt.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(hipMemcpyAsync(dyo.p, yo.data(), sizeof(int32_t) * yo.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(launch_resize_area_general(d_src, sstep, cn, d_dst, dstep, drows, dcols, dxt.as<AreaTap>(),
                                       dxo.as<int32_t>(), dyt.as<AreaTap>(), dyo.as<int32_t>(), s));
    OMR_HIP(hipStreamSynchronize(s));  // the tap tables are freed on return
    return OMR_OK;
}

int resize_dev(const DevImage &src, int drows, int dcols, int interp, DevImage *dst, hipStream_t s)
{
    if (drows <= 0 || dcols <= 0) return fail(OMR_ERR_ASSERT, "resize to an empty size");
    int rc = dst->alloc(drows, dcols, src.cn);
    if (rc) return rc;
    return resize_ptr(src.ptr(), src.step(), src.rows, src.cols, src.cn, dst->ptr(), dst->step(), drows, dcols, interp, s);
}
int resize_area(const DevImage &src, int drows, int dcols, DevImage *dst, hipStream_t s)
{
    return resize_dev(src, drows, dcols, OMR_INTER_AREA, dst, s);
}

// transfer.rs:66-91 scale_self: size truncated with `as i32`, INTER_LINEAR when enlarging
int scale_dev(const DevImage &src, double scale, DevImage *dst, hipStream_t s)
{
    const int dc = (int)((double)src.cols * scale), dr = (int)((double)src.rows * scale);
    return resize_dev(src, dr, dc, scale > 1.0 ? OMR_INTER_LINEAR : OMR_INTER_AREA, dst, s);
}

int give_owned(const DevImage &img, omr_image_owned *dst, hipStream_t s)
{
    dst->rows = img.rows;
    dst->cols = img.cols;
    dst->channels = img.cn;
    dst->step_bytes = img.step();
    dst->data = (uint8_t *)malloc((size_t)img.rows * dst->step_bytes);
    if (!dst->data) return fail(OMR_ERR_NOMEM, "out of host memory");
    int rc = img.download(dst->data, dst->step_bytes, s);
    if (rc) omr_image_free(dst);
    return rc;
}

int to_gray(const DevImage &src, DevImage *gray, hipStream_t s)
{
    int rc = gray->alloc(src.rows, src.cols, 1);
    if (rc) return rc;
    if (src.cn == 1) {
        // the reference's cvtColor would raise on a 1-channel Mat; accepted here as a convenience
        OMR_HIP(hipMemcpyAsync(gray->buf.p, src.buf.p, (size_t)src.rows * src.cols, hipMemcpyDeviceToDevice, s));
        return OMR_OK;
    }
    if (src.cn != 3 && src.cn != 4) return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels, got %d", src.cn);
    OMR_HIP(launch_rgb2gray_fast(src.ptr(), src.step(), src.rows, src.cols, src.cn, gray->ptr(), gray->step(), s));
    return OMR_OK;
}

// projection.rs:125-190 on the host copy of the scores (lowest index on exact ties, quirk B5).
int argmax_path1(const double *v, const double *h, int n)
{
    double vmax = v[0], hmax = h[0];
    int vcount = 1, hcount = 1, vfirst = 0, hfirst = 0;
    for (int i = 0; i < n; i++) {
        if (v[i] > vmax) vmax = v[i], vcount = 1, vfirst = i;
        else if (v[i] == vmax) vcount++;
        if (h[i] > hmax) hmax = h[i], hcount = 1, hfirst = i;
        else if (h[i] == hmax) hcount++;
    }
    if (vcount == 1 && hcount == 1 && vfirst == hfirst) return vfirst;
    double sdp = 0.0;
    int result = -1;
    for (int i = 0; i < n; i++)
        if (v[i] == vmax || h[i] == hmax) {
            double cur = v[i] * v[i] + h[i] * h[i];
            if (sdp < cur) sdp = cur, result = i;
        }
    return result < 0 ? n / 2 : result;
}

}  // namespace

namespace omr {
int result_from_projection_device(const uint8_t *d_src, int rows, int cols, int cn, uint16_t max_angle, double step,
                                  int32_t max_w, int32_t max_h, hipStream_t s, double *angle, int32_t *status,
                                  double *candidates, int32_t cand_cap, int32_t *cand_len);
}

extern "C" {

void omr_image_free(omr_image_owned *img)
{
    if (img && img->data) {
        free(img->data);
        img->data = nullptr;
    }
}

int omr_argmax_projection(const double *v_sd, const double *h_sd, int32_t n, int32_t *index_out)
{
    if (!v_sd || !h_sd || !index_out || n <= 0) return fail(OMR_ERR_BADARG, "bad arguments");
    *index_out = argmax_path1(v_sd, h_sd, n);
    return OMR_OK;
}

// omr.rs:147-221
int omr_select_projection_result(const double *v_sd, const double *h_sd, int32_t n, int32_t N, double step,
                                 double *angle, int32_t *status, double *candidates, int32_t cand_cap,
                                 int32_t *cand_len)
{
    if (!v_sd || !h_sd || !angle || !status || n < 0) return fail(OMR_ERR_BADARG, "bad arguments");
    double max_h = 0.0, max_v = 0.0;
    unsigned hc = 1, vc = 1;
    std::vector<double> cand;
    for (int i = 0; i < n; i++) {
        double ang = (double)(i - N) * step;
        double hs = h_sd[i];
        if (max_h < hs) {
            max_h = hs;
            max_v = v_sd[i];
            hc = 1;
            vc = 1;
            cand.assign(1, ang);
        } else if (max_h == hs) {
            hc += 1;
            double vs = v_sd[i];
            if (max_v < vs) {
                vc = 1;
                max_v = vs;
                cand.assign(1, ang);
            } else if (max_v == vs) {
                vc += 1;
                cand.push_back(ang);
            }
        }
    }
    if (cand_len) *cand_len = (int32_t)cand.size();
    if (candidates)
        for (size_t i = 0; i < cand.size() && (int32_t)i < cand_cap; i++) candidates[i] = cand[i];
    if (cand.empty()) {  // omr.rs:211 would panic on [0] (quirk B6)
        *status = OMR_STATUS_NOT_A_RESULT;
        *angle = 0.0;
    } else if (hc == 1 && vc == 1) {
        *status = OMR_STATUS_BELIEVED;
        *angle = cand[0];
    } else if (cand.size() == 1) {
        *status = OMR_STATUS_NEED_CHECK;
        *angle = cand[0];
    } else {
        *status = OMR_STATUS_NOT_A_RESULT;
        *angle = 0.0;
    }
    return OMR_OK;
}

// projection.rs:17-194
int omr_get_angle_with_projections(const omr_image *src, uint16_t max_angle, double step, double resize_scale,
                                   size_t threads_hint, double *angle_out)
{
    (void)threads_hint;  // projection.rs:69-122 is buggy (:94) and unused by every caller
    int rc = check_image(src, false);
    if (rc) return rc;
    if (!angle_out) return fail(OMR_ERR_BADARG, "null angle_out");
    if (src->channels == 2) return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels");
    int N, A = candidate_count(max_angle, step, &N);
    if (A <= 0) return fail(OMR_ERR_BADARG, "empty candidate range (the reference indexes [0] and panics)");
    int dev;
    if ((rc = current_device(&dev))) return rc;
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in, scaled, gray;
    if ((rc = in.upload(src, st.s))) return rc;
    const DevImage *cur = &in;
    if (resize_scale != 1.0) {  // :24-27 scale_self (transfer.rs:66-91)
        if ((rc = scale_dev(in, resize_scale, &scaled, st.s))) return rc;
        cur = &scaled;
    }
    if ((rc = to_gray(*cur, &gray, st.s))) return rc;  // :30
    std::vector<double> vs, hs;
    // :31 threshold(127,255,BINARY) is fused into the bit-pack (black iff gray <= 127)
    if ((rc = sweep_scores(gray, 127, max_angle, step, 1.0, dev, st.s, &vs, &hs, nullptr))) return rc;
    int idx = argmax_path1(vs.data(), hs.data(), A);
    *angle_out = ((double)idx - (double)N) * step;  // :189-190
    return OMR_OK;
}

// app/src-tauri/src/test.rs:83-178 (same driver on an already binarised image)
int omr_find_target_angle(uint16_t max_angle, double step, const omr_image *thresh, size_t threads_hint,
                          double *angle_out)
{
    (void)threads_hint;
    int rc = check_image(thresh, true);
    if (rc) return rc;
    if (!angle_out) return fail(OMR_ERR_BADARG, "null angle_out");
    int N, A = candidate_count(max_angle, step, &N);
    if (A <= 0) return fail(OMR_ERR_BADARG, "empty candidate range");
    int dev;
    if ((rc = current_device(&dev))) return rc;
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in;
    if ((rc = in.upload(thresh, st.s))) return rc;
    std::vector<double> vs, hs;
    if ((rc = sweep_scores(in, 0, max_angle, step, 1.0, dev, st.s, &vs, &hs, nullptr))) return rc;
    int idx = argmax_path1(vs.data(), hs.data(), A);
    *angle_out = ((double)idx - (double)N) * step;
    return OMR_OK;
}

// omr.rs:52-229
int omr_get_result_from_projection(const omr_image *src, uint16_t max_angle, double step, int32_t max_w,
                                   int32_t max_h, double *angle, int32_t *status, double *candidates,
                                   int32_t cand_cap, int32_t *cand_len)
{
    int rc = check_image(src, false);
    if (rc) return rc;
    if (!angle || !status) return fail(OMR_ERR_BADARG, "null output");
    if (src->channels == 2) return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels");
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in;
    if ((rc = in.upload(src, st.s))) return rc;
    return omr::result_from_projection_device(in.ptr(), in.rows, in.cols, in.cn, max_angle, step, max_w, max_h, st.s, angle,
                                              status, candidates, cand_cap, cand_len);
}

// ---- per-image helpers ------------------------------------------------------------------------

int omr_threshold_binary(const omr_image *gray, uint8_t *dst, int64_t dst_step)
{
    int rc = check_image(gray, true);
    if (rc) return rc;
    if (!dst || dst_step < gray->cols) return fail(OMR_ERR_BADARG, "bad destination");
    int dev;
    if ((rc = current_device(&dev))) return rc;
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in, out;
    if ((rc = in.upload(gray, st.s))) return rc;
    if ((rc = out.alloc(in.rows, in.cols, 1))) return rc;
    OMR_HIP(launch_threshold_fast(in.ptr(), in.step(), in.rows, in.cols, out.ptr(), out.step(), 127, 255, st.s));
    return out.download(dst, dst_step, st.s);
}

int omr_rgb_to_gray(const omr_image *src, uint8_t *dst, int64_t dst_step)
{
    int rc = check_image(src, false);
    if (rc) return rc;
    if (src->channels != 3 && src->channels != 4) return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels");
    if (!dst || dst_step < src->cols) return fail(OMR_ERR_BADARG, "bad destination");
    int dev;
    if ((rc = current_device(&dev))) return rc;
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in, out;
    if ((rc = in.upload(src, st.s))) return rc;
    if ((rc = to_gray(in, &out, st.s))) return rc;
    return out.download(dst, dst_step, st.s);
}

// transfer.rs:459-523: forward matrix and canvas of rotate_mat
static int rotate_geometry_impl(int rows, int cols, double angle_deg, double scale, int clip, double M[6], int *drows,
                           int *dcols)
{
    if (clip == OMR_CLIP_DEFAULT) {  // :472-486
        *drows = rows;
        *dcols = cols;
        rotation_matrix_2d((float)cols / 2.0f, (float)rows / 2.0f, angle_deg, scale, M);
    } else if (clip == OMR_CLIP_CONTAIN) {  // :487-519
        const double CV_PI_ = 3.1415926535897932384626433832795;
        double sn = fabs(sin(angle_deg * CV_PI_ / 180.0)), cs = fabs(cos(angle_deg * CV_PI_ / 180.0));
        double rotated_width = ceil((double)rows * sn + (double)cols * cs);
        double rotated_height = ceil((double)cols * sn + (double)rows * cs);
        *dcols = (int)rotated_width;
        *drows = (int)rotated_height;
        rotation_matrix_2d((float)ceil(rotated_width / 2.0), (float)ceil(rotated_height / 2.0), angle_deg, scale, M);
        M[2] += ceil((rotated_width - (double)cols) / 2.0);
        M[5] += ceil((rotated_height - (double)rows) / 2.0);
    } else {
        return fail(OMR_ERR_BADARG, "unknown clip strategy %d", clip);
    }
    if (*drows <= 0 || *dcols <= 0 || *drows >= 32767 || *dcols >= 32767) return fail(OMR_ERR_ASSERT, "bad canvas size");
    return OMR_OK;
}

// launch the warp of rotate_mat on device buffers (1 / 3 channels: LDS-staged tiles; otherwise generic)
static int rotate_launch(const uint8_t *d_src, int64_t sstep, int rows, int cols, int cn, const double M[6], int interp,
                         const uint8_t border_value[4], uint8_t *d_dst, int64_t dstep, int drows, int dcols,
                         hipStream_t s, DevBuf *keep)
{
    double Minv[6];
    invert_affine(M, Minv);
    uint32_t border = (uint32_t)border_value[0] | ((uint32_t)border_value[1] << 8) | ((uint32_t)border_value[2] << 16) |
                      ((uint32_t)border_value[3] << 24);
    if (cn == 1 || cn == 3) {
        hipError_t e = launch_warp_fast(d_src, sstep, rows, cols, cn, d_dst, dstep, drows, dcols, Minv, interp, border, s);
        if (e == hipSuccess) return OMR_OK;
        return fail_gpu("launch_warp_fast", e);
    }
    OMR_HIP(keep->alloc(sizeof Minv));
    OMR_HIP(hipMemcpyAsync(keep->p, Minv, sizeof Minv, hipMemcpyHostToDevice, s));
    OMR_HIP(hipStreamSynchronize(s));  // Minv is a stack buffer
    if (interp == OMR_INTER_NEAREST)
        OMR_HIP(launch_warp_nn(d_src, sstep, rows, cols, cn, d_dst, dstep, drows, dcols, keep->as<double>(), border, s));
    else
        OMR_HIP(launch_warp_linear(d_src, sstep, rows, cols, cn, d_dst, dstep, drows, dcols, keep->as<double>(), border, s));
    return OMR_OK;
}

int omr_rotate_size(int32_t rows, int32_t cols, double angle_deg, int32_t clip, int32_t *dst_rows, int32_t *dst_cols)
{
    if (!dst_rows || !dst_cols || rows <= 0 || cols <= 0) return fail(OMR_ERR_BADARG, "bad arguments");
    double M[6];
    int dr, dc;
    int rc = rotate_geometry_impl(rows, cols, angle_deg, 1.0, clip, M, &dr, &dc);
    if (rc) return rc;
    *dst_rows = dr;
    *dst_cols = dc;
    return OMR_OK;
}

int omr_rotate_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols, int32_t channels,
                      double angle_deg, double scale, int32_t interp, const uint8_t border_value[4], int32_t clip,
                      uint8_t *d_dst, int64_t dst_step, int32_t dst_rows, int32_t dst_cols, void *stream)
{
    if (!d_src || !d_dst || !border_value) return fail(OMR_ERR_BADARG, "null pointer");
    if (rows <= 0 || cols <= 0 || rows >= 32767 || cols >= 32767 || channels < 1 || channels > 4)
        return fail(OMR_ERR_ASSERT, "bad image shape");
    if (interp != OMR_INTER_NEAREST && interp != OMR_INTER_LINEAR)
        return fail(OMR_ERR_NOTIMPL, "interpolation flag %d is not implemented", interp);
    double M[6];
    int dr, dc;
    int rc = rotate_geometry_impl(rows, cols, angle_deg, scale, clip, M, &dr, &dc);
    if (rc) return rc;
    if (dr != dst_rows || dc != dst_cols) return fail(OMR_ERR_ASSERT, "destination must be %dx%d", dc, dr);
    if (src_step < (int64_t)cols * channels || dst_step < (int64_t)dc * channels) return fail(OMR_ERR_BADARG, "step too small");
    DevBuf keep;
    rc = rotate_launch(d_src, src_step, rows, cols, channels, M, interp, border_value, d_dst, dst_step, dr, dc,
                       (hipStream_t)stream, &keep);
    if (!rc && keep.p) OMR_HIP(hipStreamSynchronize((hipStream_t)stream));  // generic path: matrix buffer is freed
    return rc;
}

int omr_rotate(const omr_image *src, double angle_deg, double scale, int32_t interp, const uint8_t border_value[4],
               int32_t clip, omr_image_owned *dst)
{
    int rc = check_image(src, false);
    if (rc) return rc;
    if (!dst || !border_value) return fail(OMR_ERR_BADARG, "null output");
    if (interp != OMR_INTER_NEAREST && interp != OMR_INTER_LINEAR)
        return fail(OMR_ERR_NOTIMPL, "interpolation flag %d is not implemented", interp);
    double M[6];
    int drows, dcols;
    if ((rc = rotate_geometry_impl(src->rows, src->cols, angle_deg, scale, clip, M, &drows, &dcols))) return rc;
    int dev;
    if ((rc = current_device(&dev))) return rc;
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in, out;
    DevBuf keep;
    if ((rc = in.upload(src, st.s))) return rc;
    if ((rc = out.alloc(drows, dcols, src->channels))) return rc;
    if ((rc = rotate_launch(in.ptr(), in.step(), in.rows, in.cols, in.cn, M, interp, border_value, out.ptr(), out.step(),
                            drows, dcols, st.s, &keep)))
        return rc;
    dst->rows = drows;
    dst->cols = dcols;
    dst->channels = src->channels;
    dst->step_bytes = (int64_t)dcols * src->channels;
    dst->data = (uint8_t *)malloc((size_t)drows * dst->step_bytes);
    if (!dst->data) return fail(OMR_ERR_NOMEM, "out of host memory");
    rc = out.download(dst->data, dst->step_bytes, st.s);
    if (rc) omr_image_free(dst);
    return rc;
}

// ---- device-resident stages --------------------------------------------------------------------
static int check_dev_image(const void *s, const void *d, int rows, int cols, int64_t sstep, int64_t dstep, int scn,
                           int dcn)
{
    if (!s || !d) return fail(OMR_ERR_BADARG, "null device pointer");
    if (rows <= 0 || cols <= 0 || rows >= 32767 || cols >= 32767) return fail(OMR_ERR_ASSERT, "bad image shape");
    if (sstep < (int64_t)cols * scn || dstep < (int64_t)cols * dcn) return fail(OMR_ERR_BADARG, "step too small");
    return OMR_OK;
}

int omr_rgb_to_gray_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols, int32_t channels,
                           uint8_t *d_dst, int64_t dst_step, void *stream)
{
    if (channels != 3 && channels != 4) return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels");
    int rc = check_dev_image(d_src, d_dst, rows, cols, src_step, dst_step, channels, 1);
    if (rc) return rc;
    OMR_HIP(launch_rgb2gray_fast(d_src, src_step, rows, cols, channels, d_dst, dst_step, (hipStream_t)stream));
    return OMR_OK;
}

int omr_erode3_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols, uint8_t *d_dst,
                      int64_t dst_step, void *stream)
{
    int rc = check_dev_image(d_src, d_dst, rows, cols, src_step, dst_step, 1, 1);
    if (rc) return rc;
    if (d_src == d_dst) return fail(OMR_ERR_BADARG, "erode cannot run in place");
    OMR_HIP(launch_erode3x_cross(d_src, src_step, rows, cols, d_dst, dst_step, (hipStream_t)stream));
    return OMR_OK;
}

int omr_resize_area_device(const uint8_t *d_src, int64_t src_step, int32_t src_rows, int32_t src_cols,
                           int32_t channels, uint8_t *d_dst, int64_t dst_step, int32_t dst_rows, int32_t dst_cols,
                           void *stream)
{
    if (!d_src || !d_dst) return fail(OMR_ERR_BADARG, "null device pointer");
    if (src_rows <= 0 || src_cols <= 0 || dst_rows <= 0 || dst_cols <= 0 || channels < 1 || channels > 4)
        return fail(OMR_ERR_ASSERT, "bad image shape");
    if (src_step < (int64_t)src_cols * channels || dst_step < (int64_t)dst_cols * channels)
        return fail(OMR_ERR_BADARG, "step too small");
    return resize_ptr(d_src, src_step, src_rows, src_cols, channels, d_dst, dst_step, dst_rows, dst_cols, OMR_INTER_AREA,
                      (hipStream_t)stream);
}

// ---- transfer.rs:66-145 on host images: scale_self / shrink_to / resize_self ------------------------
static int resize_host(const omr_image *src, int drows, int dcols, int interp, omr_image_owned *dst)
{
    int rc = check_image(src, false);
    if (rc) return rc;
    if (!dst) return fail(OMR_ERR_BADARG, "null output");
    if (drows <= 0 || dcols <= 0) return fail(OMR_ERR_ASSERT, "resize to an empty size (the reference's resize raises)");
    if (drows >= 32767 || dcols >= 32767) return fail(OMR_ERR_ASSERT, "image dimension >= SHRT_MAX");
    int dev;
    if ((rc = current_device(&dev))) return rc;
    Stream st;
    if ((rc = st.create())) return rc;
    DevImage in, out;
    if ((rc = in.upload(src, st.s))) return rc;
    if ((rc = resize_dev(in, drows, dcols, interp, &out, st.s))) return rc;
    return give_owned(out, dst, st.s);
}

int omr_scale(const omr_image *src, double scale, omr_image_owned *dst)
{
    if (!src) return fail(OMR_ERR_BADARG, "null image");
    if (!(scale > 0.0)) return fail(OMR_ERR_BADARG, "scale must be positive");
    // scale == 1.0 returns the image unchanged (transfer.rs:67-69): a copy here, outputs are always owned
    const int dc = scale == 1.0 ? src->cols : (int)((double)src->cols * scale);
    const int dr = scale == 1.0 ? src->rows : (int)((double)src->rows * scale);
    return resize_host(src, dr, dc, scale > 1.0 ? OMR_INTER_LINEAR : OMR_INTER_AREA, dst);
}

int omr_shrink_to(const omr_image *src, int32_t max_width, int32_t max_height, omr_image_owned *dst)
{
    if (!src) return fail(OMR_ERR_BADARG, "null image");
    if (src->cols <= 0 || src->rows <= 0) return fail(OMR_ERR_ASSERT, "empty image");
    const double ws = max_width <= 0 ? 1.0 : (double)max_width / (double)src->cols;   // transfer.rs:105-114
    const double hs = max_height <= 0 ? 1.0 : (double)max_height / (double)src->rows;
    const double t = ws < hs ? ws : hs;
    return omr_scale(src, t >= 1.0 ? 1.0 : t, dst);  // :121-125: never enlarges
}

int omr_resize(const omr_image *src, int32_t width, int32_t height, omr_image_owned *dst)
{
    return resize_host(src, height, width, OMR_INTER_AREA, dst);  // transfer.rs:128-145
}

int omr_threshold_binary_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols, uint8_t *d_dst,
                                int64_t dst_step, void *stream)
{
    int rc = check_dev_image(d_src, d_dst, rows, cols, src_step, dst_step, 1, 1);
    if (rc) return rc;
    OMR_HIP(launch_threshold_fast(d_src, src_step, rows, cols, d_dst, dst_step, 127, 255, (hipStream_t)stream));
    return OMR_OK;
}

// Identity "sweep" = plain projections of the image itself: X = (x*1024 + 512) >> 10 = x.
static int identity_projections(const omr_image *bin, std::vector<uint32_t> *vp, std::vector<uint32_t> *hp, double *v_sd,
                                double *h_sd)
{
    int rc = check_image(bin, true);
    if (rc) return rc;
    const double I[6] = {1, 0, 0, 0, 1, 0};
    vp->assign((size_t)bin->cols, 0);
    hp->assign((size_t)bin->rows, 0);
    double vs, hs;
    rc = omr_projection_sweep(bin, I, 1, vp->data(), hp->data(), &vs, &hs);
    if (v_sd) *v_sd = vs;
    if (h_sd) *h_sd = hs;
    return rc;
}

int omr_get_vertical_projection(const omr_image *bin, double *out_cols)
{
    if (!out_cols) return fail(OMR_ERR_BADARG, "null output");
    std::vector<uint32_t> vp, hp;
    int rc = identity_projections(bin, &vp, &hp, nullptr, nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < vp.size(); i++) out_cols[i] = (double)vp[i];
    return OMR_OK;
}

int omr_get_horizontal_projection(const omr_image *bin, double *out_rows)
{
    if (!out_rows) return fail(OMR_ERR_BADARG, "null output");
    std::vector<uint32_t> vp, hp;
    int rc = identity_projections(bin, &vp, &hp, nullptr, nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < hp.size(); i++) out_rows[i] = (double)hp[i];
    return OMR_OK;
}

int omr_get_mat_projection_data(const omr_image *bin, double *h_rows, double *v_cols)
{
    if (!h_rows || !v_cols) return fail(OMR_ERR_BADARG, "null output");
    std::vector<uint32_t> vp, hp;
    int rc = identity_projections(bin, &vp, &hp, nullptr, nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < hp.size(); i++) h_rows[i] = (double)hp[i];
    for (size_t i = 0; i < vp.size(); i++) v_cols[i] = (double)vp[i];
    return OMR_OK;
}

int omr_get_projection_standard_deviations(const omr_image *bin, double *v_sd, double *h_sd)
{
    if (!v_sd || !h_sd) return fail(OMR_ERR_BADARG, "null output");
    std::vector<uint32_t> vp, hp;
    return identity_projections(bin, &vp, &hp, v_sd, h_sd);
}

// calculate.rs:2-10 -- a sequential f64 chain over a host vector: nothing to parallelise.
int omr_get_arithmetic_mean(const double *v, size_t n, double *out)
{
    if (!v || !out || n == 0) return fail(OMR_ERR_BADARG, "empty vector (the reference indexes [0] and panics)");
    double sum = v[0];
    for (size_t i = 1; i < n; i++) sum = sum + v[i];
    *out = sum / (double)n;
    return OMR_OK;
}

// calculate.rs:13-23
int omr_get_standard_deviation(const double *v, size_t n, double *out)
{
    double mean;
    int rc = omr_get_arithmetic_mean(v, n, &mean);
    if (rc) return rc;
    double d = v[0] - mean;
    double sum = d * d;
    for (size_t i = 1; i < n; i++) {
        d = v[i] - mean;
        sum = sum + d * d;
    }
    *out = sqrt(sum / (double)n);
    return OMR_OK;
}

// Host-buffer batch over the visible devices: an omr_host_batch (oics_hostbatch.cpp) made for this one call -- one per SHAPE:
// the reference corrects one file per call, any size (app/src-tauri/src/task.rs:19-38; its own dataset holds two shapes,
// 1240x1150 and ~1237x1300), so a batch may mix shapes.  The scans are bucketed by (rows, cols), every bucket is swept with a
// context of its own (plan, ring, stages), and the results land at the scans' original positions.
int omr_sweep_batch(const omr_image *scans, int32_t n, uint16_t max_angle, double step, int32_t n_devices,
                    int32_t *best_idx, double *best_angle, double *v_sd_opt, double *h_sd_opt)
{
    if (!scans || n < 0 || !best_idx) return fail(OMR_ERR_BADARG, "bad batch arguments");
    if (n == 0) return OMR_OK;
    int N = 0;
    const int A = candidate_count(max_angle, step, &N);
    if (A <= 0) return fail(OMR_ERR_BADARG, "empty candidate range");
    std::vector<std::pair<int, int>> shapes;  // in order of first appearance
    std::vector<std::vector<int>> members;
    for (int i = 0; i < n; i++) {
        int rc = check_image(&scans[i], true);
        if (rc) return rc;
        const std::pair<int, int> sh(scans[i].rows, scans[i].cols);
        size_t k = 0;
        while (k < shapes.size() && shapes[k] != sh) k++;
        if (k == shapes.size()) {
            shapes.push_back(sh);
            members.emplace_back();
        }
        members[k].push_back(i);
    }
    const bool one = shapes.size() == 1;
    for (size_t k = 0; k < shapes.size(); k++) {
        const std::vector<int> &idx = members[k];
        const int m = (int)idx.size();
        omr_host_batch *hb = nullptr;
        int rc = omr_host_batch_create(shapes[k].first, shapes[k].second, max_angle, step, n_devices, m, &hb);
        if (rc) return rc;
        if (one) {
            rc = omr_host_batch_run(hb, scans, n, OMR_HOST_PACKED, best_idx, best_angle, v_sd_opt, h_sd_opt);
        } else {
            std::vector<omr_image> sub((size_t)m);
            std::vector<int32_t> b((size_t)m);
            std::vector<double> ang((size_t)m), vs(v_sd_opt ? (size_t)m * A : 0), hs(h_sd_opt ? (size_t)m * A : 0);
            for (int j = 0; j < m; j++) sub[(size_t)j] = scans[idx[(size_t)j]];
            rc = omr_host_batch_run(hb, sub.data(), m, OMR_HOST_PACKED, b.data(), ang.data(), v_sd_opt ? vs.data() : nullptr,
                                    h_sd_opt ? hs.data() : nullptr);
            for (int j = 0; j < m && rc == OMR_OK; j++) {
                const int i = idx[(size_t)j];
                best_idx[i] = b[(size_t)j];
                if (best_angle) best_angle[i] = ang[(size_t)j];
                if (v_sd_opt) memcpy(v_sd_opt + (size_t)i * A, vs.data() + (size_t)j * A, sizeof(double) * (size_t)A);
                if (h_sd_opt) memcpy(h_sd_opt + (size_t)i * A, hs.data() + (size_t)j * A, sizeof(double) * (size_t)A);
            }
        }
        omr_host_batch_destroy(hb);
        if (rc) return rc;
    }
    return OMR_OK;
}

}  // extern "C"

// omr.rs:52-229 on a packed device-resident image (the host-image driver and correct_default both end here, so
// a sheet that is already on the device is not uploaded again)
namespace omr {
int result_from_projection_device(const uint8_t *d_src, int rows, int cols, int cn, uint16_t max_angle, double step,
                                  int32_t max_w, int32_t max_h, hipStream_t s, double *angle, int32_t *status,
                                  double *candidates, int32_t cand_cap, int32_t *cand_len)
{
    // :60-82
    const double width_scale = max_w <= 0 ? 1.0 : (double)max_w / (double)cols;
    const double height_scale = max_h <= 0 ? 1.0 : (double)max_h / (double)rows;
    const double scale = width_scale < height_scale ? width_scale : height_scale;
    int dev, rc;
    if ((rc = current_device(&dev))) return rc;
    DevImage gray, e1, scaled;
    if ((rc = gray.alloc(rows, cols, 1))) return rc;
    if (cn == 1) {  // the reference's cvtColor would raise on a 1-channel Mat; accepted here as a convenience
        OMR_HIP(hipMemcpyAsync(gray.buf.p, d_src, (size_t)rows * cols, hipMemcpyDeviceToDevice, s));
    } else if (cn == 3 || cn == 4) {
        OMR_HIP(launch_rgb2gray_fast(d_src, (int64_t)cols * cn, rows, cols, cn, gray.ptr(), gray.step(), s));  // :88-92
    } else {
        return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels, got %d", cn);
    }
    // :98-112 erode(3x3 cross, iterations = 3)
    if ((rc = e1.alloc(rows, cols, 1))) return rc;
    OMR_HIP(launch_erode3x_cross(gray.ptr(), gray.step(), rows, cols, e1.ptr(), e1.step(), s));
    // :114-126
    const int dc = (int)((double)cols * scale), dr = (int)((double)rows * scale);
    if ((rc = resize_area(e1, dr, dc, &scaled, s))) return rc;
    // :129-139 threshold fused into the pack; :153-208 sweep with matrix scale = resize scale (quirk B4)
    std::vector<double> vs, hs;
    int N = 0;
    if ((rc = sweep_scores(scaled, 127, max_angle, step, scale, dev, s, &vs, &hs, &N))) return rc;
    return omr_select_projection_result(vs.data(), hs.data(), (int32_t)vs.size(), N, step, angle, status, candidates,
                                        cand_cap, cand_len);
}

// rotate_mat (transfer.rs:459-523) of a packed device-resident image into a new host image
int rotate_device_to_host(const uint8_t *d_src, int rows, int cols, int cn, double angle_deg, double scale, int interp,
                          const uint8_t border_value[4], int clip, hipStream_t s, omr_image_owned *dst)
{
    double M[6];
    int drows, dcols, rc;
    if ((rc = rotate_geometry_impl(rows, cols, angle_deg, scale, clip, M, &drows, &dcols))) return rc;
    DevImage out;
    DevBuf keep;
    if ((rc = out.alloc(drows, dcols, cn))) return rc;
    if ((rc = rotate_launch(d_src, (int64_t)cols * cn, rows, cols, cn, M, interp, border_value, out.ptr(), out.step(), drows,
                            dcols, s, &keep)))
        return rc;
    return give_owned(out, dst, s);
}
int rotate_geometry(int rows, int cols, double angle_deg, double scale, int clip, double M[6], int *drows, int *dcols)
{
    return rotate_geometry_impl(rows, cols, angle_deg, scale, clip, M, drows, dcols);
}
}  // namespace omr
