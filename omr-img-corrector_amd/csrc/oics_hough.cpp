// oics_hough.cpp -- host side of the Hough-line deskew path and of
// correct_default, exported through the C ABI with the reference's names and argument meaning:
//
//   oics::hough::get_angle_with_hough            packages/lib/src/hough.rs:17-100
//   oics::omr::get_result_from_edges_detection   packages/lib/src/omr.rs:231-302
//   oics::omr::correct_default                   packages/lib/src/omr.rs:339-448 (minus imread / imwrite)
//   imgproc::canny / imgproc::hough_lines_p      call sites hough.rs:27-43, omr.rs:236-253
//
// The image work (Canny, point list, progressive probabilistic Hough, the O(n^2) vote of large line
// sets) runs on the GPU (hough.hip).  The host builds the 180-entry trigonometric / walk tables
// (double cos / sin -> float, exactly hough.cpp's expressions), turns the segments into angles with
// libm atan2f / fmodf (what Rust's f32::atan2 and % call) and applies the reference's selection
// rules.  No CPU fallback: without a HIP device every entry point returns -217.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/omrdeskew.h"
#include "engine.hpp"
#include "hough.hpp"
#include "hough_host.hpp"

using namespace omr;
using namespace omr::hh;

namespace omr {
namespace hh {

int HStream::create()
{
    int rc = lease_call_slot(&slot);  // a pooled stream, returned when the call ends
    if (rc) return rc;
    s = slot->stream;
    pool.reset(new PoolScope(s));
    return OMR_OK;
}
HStream::~HStream()
{
    pool.reset();
    return_call_slot(slot);
}

int have_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(OMR_ERR_GPU, "no usable HIP device (there is no CPU fallback)");
    return OMR_OK;
}

int check_img(const omr_image *im)
{
    if (!im || !im->data) return fail(OMR_ERR_BADARG, "null image");
    if (im->rows <= 0 || im->cols <= 0) return fail(OMR_ERR_ASSERT, "empty image");
    if (im->rows >= 32767 || im->cols >= 32767) return fail(OMR_ERR_ASSERT, "image dimension >= SHRT_MAX");
    if (im->channels != 1 && im->channels != 3 && im->channels != 4)
        return fail(OMR_ERR_ASSERT, "Canny / HoughLinesP take 1, 3 or 4 channels, got %d", im->channels);
    if (im->step_bytes < (int64_t)im->cols * im->channels) return fail(OMR_ERR_BADARG, "step_bytes too small");
    return OMR_OK;
}

// omr_hough_set_scans_in_flight(): 0 = the default (see ppht_device)
std::atomic<int> g_scans_in_flight{0};

inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_round(float v) { return (int)lrintf(v); }

// Canny on n device-resident scans of one shape -> d_map holds the edges (0 / 255), packed.
int canny_device(const uint8_t *d_src, int64_t scan_stride, int64_t step, int rows, int cols, int cn, int n, double low_t,
                 double high_t, uint8_t *d_map, int *d_flag, hipStream_t s, int32_t *d_rowcnt)
{
    if (low_t > high_t) std::swap(low_t, high_t);
    const int low = (int)floor(low_t), high = (int)floor(high_t);
    OMR_HIP(launch_canny_nms(d_src, scan_stride, step, rows, cols, cn, n, low, high, d_map, s));
    for (int pass = 0; pass < 100000; pass++) {
        int flag = 0;
        OMR_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), s));
        OMR_HIP(launch_canny_hysteresis(d_map, rows, cols, n, d_flag, s));
        OMR_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        OMR_HIP(hipStreamSynchronize(s));
        if (!flag) break;
    }
    OMR_HIP(launch_edges_rowcount(d_map, rows, cols, n, 1, d_rowcnt, s));
    return OMR_OK;
}

// HoughLinesP on n device-resident edge images (d_edges: packed, kept; d_rowcnt filled).
int ppht_device(uint8_t *d_edges, int32_t *d_rowcnt, int rows, int cols, int n, const HoughParams &hp, hipStream_t s,
                std::vector<std::vector<int32_t>> *lines_out)
{
    const float rho = (float)hp.rho, theta = (float)hp.theta, irho = 1.0f / rho;
    if (!(rho > 0) || !(theta > 0)) return fail(OMR_ERR_BADARG, "rho and theta must be positive");
    const int numangle = cv_round(3.1415926535897932384626433832795 / theta);
    if (numangle <= 0 || numangle > OMR_PPHT_MAX_ANGLES)
        return fail(OMR_ERR_NOTIMPL, "HoughLinesP: %d accumulator angles (theta too small; the reference uses pi/180)",
                    numangle);
    std::vector<float> ttab((size_t)numangle * 2);
    std::vector<PphtWalk> walk((size_t)numangle);
    for (int k = 0; k < numangle; k++) {
        ttab[2 * k] = (float)(cos((double)k * theta) * irho);
        ttab[2 * k + 1] = (float)(sin((double)k * theta) * irho);
        const float a = -ttab[2 * k + 1], b = ttab[2 * k];
        PphtWalk w{};
        if (fabsf(a) > fabsf(b)) {
            w.xflag = 1;
            w.dx0 = a > 0 ? 1 : -1;
            w.dy0 = cv_round(b * (float)(1 << 16) / fabsf(a));
        } else {
            w.xflag = 0;
            w.dy0 = b > 0 ? 1 : -1;
            w.dx0 = cv_round(a * (float)(1 << 16) / fabsf(b));
        }
        walk[k] = w;
    }
    // OpenCV's accumulator has (cols + rows) * 2 + 1 bins per angle; a pixel of a cols x rows image only reaches
    // rho = cvRound(x cos + y sin) in [lo_k, hi_k], so row k keeps just that range (+-2 for the float rounding)
    std::vector<int32_t> row_base((size_t)numangle);
    int64_t accum_stride = 0;
    for (int k = 0; k < numangle; k++) {
        const double c = ttab[2 * k], sn = ttab[2 * k + 1];
        const double lo = (cols - 1) * std::min(c, 0.0) + (rows - 1) * std::min(sn, 0.0);
        const double hi = (cols - 1) * std::max(c, 0.0) + (rows - 1) * std::max(sn, 0.0);
        const int64_t rmin = (int64_t)floor(lo) - 2, rmax = (int64_t)ceil(hi) + 2;
        row_base[k] = (int32_t)(accum_stride - rmin);
        accum_stride += rmax - rmin + 1;
    }
    accum_stride += 64;  // scratch bins of the lanes that hold no angle
    accum_stride += accum_stride & 1;
    const bool acc_u16 = rows + cols <= OMR_PPHT_U16_MAX_EXTENT;  // hough.hip: a bin's count is bounded by the diagonal
    const size_t bin_bytes = acc_u16 ? sizeof(uint16_t) : sizeof(int32_t);
    DevBuf rowoff, total, scanoff, nz, order, d_ttab, d_walk, d_rowbase, accum, lines, nlines, tiled, queue;
    OMR_HIP(rowoff.alloc(sizeof(int32_t) * (size_t)n * rows));
    OMR_HIP(total.alloc(sizeof(int32_t) * (size_t)n));
    OMR_HIP(launch_edges_rowscan(d_rowcnt, rows, n, rowoff.as<int32_t>(), total.as<int32_t>(), s));
    std::vector<int32_t> counts((size_t)n);
    OMR_HIP(hipMemcpyAsync(counts.data(), total.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    OMR_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> off((size_t)n);
    int64_t sum = 0;
    int maxc = 0;
    for (int i = 0; i < n; i++) {
        off[i] = sum;
        sum += counts[i];
        maxc = std::max(maxc, counts[i]);
    }
    const int cap = std::max(1, std::min(maxc, 1 << 16));
    OMR_HIP(scanoff.alloc(sizeof(int64_t) * (size_t)n));
    OMR_HIP(nz.alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(sum, 1)));
    OMR_HIP(order.alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(sum, 1)));
    OMR_HIP(hipMemcpyAsync(scanoff.p, off.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, s));
    OMR_HIP(tiled.alloc((size_t)n * (size_t)ppht_mask_bytes(rows, cols)));
    OMR_HIP(hipMemsetAsync(tiled.p, 0, tiled.bytes, s));
    OMR_HIP(launch_edges_compact(d_edges, rows, cols, n, rowoff.as<int32_t>(), scanoff.as<int64_t>(), nz.as<uint32_t>(),
                                 tiled.as<uint8_t>(), s));
    OMR_HIP(d_ttab.alloc(sizeof(float) * ttab.size()));
    OMR_HIP(d_walk.alloc(sizeof(PphtWalk) * walk.size()));
    OMR_HIP(hipMemcpyAsync(d_ttab.p, ttab.data(), sizeof(float) * ttab.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(hipMemcpyAsync(d_walk.p, walk.data(), sizeof(PphtWalk) * walk.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(d_rowbase.alloc(sizeof(int32_t) * row_base.size()));
    OMR_HIP(hipMemcpyAsync(d_rowbase.p, row_base.data(), sizeof(int32_t) * row_base.size(), hipMemcpyHostToDevice, s));
    OMR_HIP(accum.alloc(bin_bytes * (size_t)n * (size_t)accum_stride));
    OMR_HIP(hipMemsetAsync(accum.p, acc_u16 ? 0x80 : 0, accum.bytes, s));
    OMR_HIP(lines.alloc(sizeof(int32_t) * 4 * (size_t)n * cap));
    OMR_HIP(nlines.alloc(sizeof(int32_t) * (size_t)n));
    PphtArgs a{};
    a.mask = tiled.as<uint8_t>();
    a.width = cols;
    a.height = rows;
    a.nz = nz.as<uint32_t>();
    a.order = order.as<uint32_t>();
    a.scan_off = scanoff.as<int64_t>();
    a.count = total.as<int32_t>();
    a.accum = accum.p;
    a.accum_stride = accum_stride;
    a.acc_u16 = acc_u16 ? 1 : 0;
    a.row_base = d_rowbase.as<int32_t>();
    a.numangle = numangle;
    a.ttab = d_ttab.as<float>();
    a.walk = d_walk.as<PphtWalk>();
    a.threshold = hp.threshold;
    a.line_length = cv_round(hp.min_line_length);
    a.line_gap = cv_round(hp.max_line_gap);
    a.lines = lines.as<int32_t>();
    a.cap = cap;
    a.n_lines = nlines.as<int32_t>();
    OMR_HIP(queue.alloc(sizeof(int32_t)));
    OMR_HIP(hipMemsetAsync(queue.p, 0, sizeof(int32_t), s));
    a.n_scans = n;
    a.queue = queue.as<int32_t>();
    OMR_HIP(launch_ppht(a, g_scans_in_flight.load(std::memory_order_relaxed), s));
    std::vector<int32_t> nl((size_t)n);
    OMR_HIP(hipMemcpyAsync(nl.data(), nlines.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    OMR_HIP(hipStreamSynchronize(s));
    lines_out->assign((size_t)n, {});
    for (int i = 0; i < n; i++) {
        if (nl[i] > cap) return fail(OMR_ERR_NOMEM, "HoughLinesP: %d segments exceed the buffer of %d", nl[i], cap);
        (*lines_out)[i].resize((size_t)nl[i] * 4);
        if (nl[i])
            OMR_HIP(hipMemcpyAsync((*lines_out)[i].data(), lines.as<int32_t>() + (size_t)i * cap * 4,
                                   sizeof(int32_t) * 4 * (size_t)nl[i], hipMemcpyDeviceToHost, s));
    }
    OMR_HIP(hipStreamSynchronize(s));
    return OMR_OK;
}

// hough.rs:50-68 / omr.rs:257-267
void line_angles(const std::vector<int32_t> &l, std::vector<float> *ang)
{
    const float pi32 = 3.14159274101257324f;  // std::f32::consts::PI
    const size_t n = l.size() / 4;
    ang->resize(n);
    for (size_t i = 0; i < n; i++) {
        const float x1 = (float)l[4 * i], y1 = (float)l[4 * i + 1], x2 = (float)l[4 * i + 2], y2 = (float)l[4 * i + 3];
        float angle = atan2f(y2 - y1, x2 - x1) * 180.0f / pi32;
        (*ang)[i] = fmodf(angle, 45.0f);
    }
}

// counts[i] = #{j : |a_i - a_j| < 0.1}: on the host for small sets, on the GPU otherwise
int vote_counts(const std::vector<float> &ang, bool as_f64, hipStream_t s, std::vector<int32_t> *counts)
{
    const int n = (int)ang.size();
    counts->assign((size_t)n, 0);
    if (n <= 2048) {
        for (int i = 0; i < n; i++) {
            int c = 0;
            if (as_f64) {
                for (int j = 0; j < n; j++) c += fabs((double)ang[i] - (double)ang[j]) < 0.1;
            } else {
                for (int j = 0; j < n; j++) c += fabsf(ang[i] - ang[j]) < 0.1f;
            }
            (*counts)[i] = c;
        }
        return OMR_OK;
    }
    DevBuf da, dc;
    OMR_HIP(da.alloc(sizeof(float) * (size_t)n));
    OMR_HIP(dc.alloc(sizeof(int32_t) * (size_t)n));
    OMR_HIP(hipMemcpyAsync(da.p, ang.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
    OMR_HIP(launch_angle_votes(da.as<float>(), n, as_f64 ? 1 : 0, dc.as<int32_t>(), s));
    OMR_HIP(hipMemcpyAsync(counts->data(), dc.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    OMR_HIP(hipStreamSynchronize(s));
    return OMR_OK;
}

// hough.rs:70-92: strict ">" keeps the first maximum
int select_hough_rs(const std::vector<float> &ang, const std::vector<int32_t> &cnt, double *angle)
{
    if (ang.empty()) return fail(OMR_ERR_ASSERT, "no line segment found (the reference panics on angles[0], hough.rs:74)");
    float target = ang[0];
    int best = 0;
    for (size_t i = 0; i < ang.size(); i++)
        if (cnt[i] > best) {
            target = ang[i];
            best = cnt[i];
        }
    *angle = (double)target;
    return OMR_OK;
}

// omr.rs:268-301
int select_omr_rs(const std::vector<float> &ang, const std::vector<int32_t> &cnt, double *angle, int32_t *status,
                  double *candidates, int32_t cand_cap, int32_t *cand_len)
{
    if (ang.empty()) return fail(OMR_ERR_ASSERT, "no line segment found (the reference panics on angles[0], omr.rs:272)");
    double target = (double)ang[0];
    int best = 0, nc = 0;
    for (size_t i = 0; i < ang.size(); i++) {
        if (cnt[i] > best) {
            target = (double)ang[i];
            best = cnt[i];
            if (candidates && cand_cap > 0) candidates[0] = target;
            nc = 1;
        } else if (cnt[i] == best) {
            if (candidates && nc < cand_cap) candidates[nc] = (double)ang[i];
            nc++;
        }
    }
    *angle = target;
    if (cand_len) *cand_len = nc;
    if (status) *status = nc == 0 ? OMR_STATUS_NOT_A_RESULT : (nc == 1 ? OMR_STATUS_BELIEVED : OMR_STATUS_NEED_CHECK);
    return OMR_OK;
}

// upload a host image into a packed device buffer
int upload(const omr_image *im, DevBuf *buf, hipStream_t s)
{
    const size_t row = (size_t)im->cols * im->channels;
    OMR_HIP(buf->alloc(row * (size_t)im->rows));
    if ((size_t)im->step_bytes == row)  // packed: one linear copy (the 2-D path is slow for odd widths)
        OMR_HIP(hipMemcpyAsync(buf->p, im->data, row * (size_t)im->rows, hipMemcpyHostToDevice, s));
    else
        OMR_HIP(hipMemcpy2DAsync(buf->p, row, im->data, (size_t)im->step_bytes, row, (size_t)im->rows, hipMemcpyHostToDevice, s));
    return OMR_OK;
}

// Canny + HoughLinesP of n device-resident scans -> per-scan segments
int edges_lines_device(const uint8_t *d_src, int64_t scan_stride, int64_t step, int rows, int cols, int cn, int n,
                       const HoughParams &hp, hipStream_t s, std::vector<std::vector<int32_t>> *lines)
{
    DevBuf map, flag, rowcnt;
    OMR_HIP(map.alloc((size_t)n * rows * cols));
    OMR_HIP(flag.alloc(sizeof(int)));
    OMR_HIP(rowcnt.alloc(sizeof(int32_t) * (size_t)n * rows));
    int rc = canny_device(d_src, scan_stride, step, rows, cols, cn, n, hp.low, hp.high, map.as<uint8_t>(), flag.as<int>(), s,
                          rowcnt.as<int32_t>());
    if (rc) return rc;
    return ppht_device(map.as<uint8_t>(), rowcnt.as<int32_t>(), rows, cols, n, hp, s, lines);
}

}  // namespace hh
}  // namespace omr

extern "C" {

int32_t omr_hough_set_scans_in_flight(int32_t scans)
{
    return (int32_t)omr::hh::g_scans_in_flight.exchange(scans < 0 ? 0 : scans);
}

int omr_canny(const omr_image *src, double low_thresh, double high_thresh, omr_image_owned *edges)
{
    clear_error();
    int rc = check_img(src);
    if (rc) return rc;
    if (!edges) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    DevBuf in, map, flag, rowcnt;
    if ((rc = upload(src, &in, st.s))) return rc;
    OMR_HIP(map.alloc((size_t)src->rows * src->cols));
    OMR_HIP(flag.alloc(sizeof(int)));
    OMR_HIP(rowcnt.alloc(sizeof(int32_t) * (size_t)src->rows));
    if ((rc = canny_device(in.as<uint8_t>(), 0, (int64_t)src->cols * src->channels, src->rows, src->cols, src->channels, 1,
                           low_thresh, high_thresh, map.as<uint8_t>(), flag.as<int>(), st.s, rowcnt.as<int32_t>())))
        return rc;
    edges->rows = src->rows;
    edges->cols = src->cols;
    edges->channels = 1;
    edges->step_bytes = src->cols;
    edges->data = (uint8_t *)malloc((size_t)src->rows * src->cols);
    if (!edges->data) return fail(OMR_ERR_NOMEM, "out of host memory");
    rc = staged_d2h(edges->data, map.p, (size_t)src->rows * src->cols, st.s);
    if (rc) omr_image_free(edges);
    return rc;
}

int omr_hough_lines_p(const omr_image *edges, double rho, double theta, int32_t threshold, double min_line_length,
                      double max_line_gap, int32_t *lines, int32_t cap, int32_t *n_lines)
{
    clear_error();
    int rc = check_img(edges);
    if (rc) return rc;
    if (edges->channels != 1) return fail(OMR_ERR_ASSERT, "HoughLinesP takes an 8-bit single-channel image");
    if (!n_lines || (cap > 0 && !lines)) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    DevBuf img, rowcnt;
    if ((rc = upload(edges, &img, st.s))) return rc;
    OMR_HIP(rowcnt.alloc(sizeof(int32_t) * (size_t)edges->rows));
    OMR_HIP(launch_edges_rowcount(img.as<uint8_t>(), edges->rows, edges->cols, 1, 0, rowcnt.as<int32_t>(), st.s));
    HoughParams hp;
    hp.rho = rho;
    hp.theta = theta;
    hp.threshold = threshold;
    hp.min_line_length = min_line_length;
    hp.max_line_gap = max_line_gap;
    std::vector<std::vector<int32_t>> out;
    if ((rc = ppht_device(img.as<uint8_t>(), rowcnt.as<int32_t>(), edges->rows, edges->cols, 1, hp, st.s, &out))) return rc;
    const int n = (int)(out[0].size() / 4);
    *n_lines = n;
    if (lines && cap > 0) memcpy(lines, out[0].data(), sizeof(int32_t) * 4 * (size_t)std::min(n, (int)cap));
    return OMR_OK;
}

int omr_get_angle_with_hough(const omr_image *gray, double min_line_length, double max_line_gap, double *angle_out)
{
    clear_error();
    int rc = check_img(gray);
    if (rc) return rc;
    if (!angle_out) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    DevBuf in;
    if ((rc = upload(gray, &in, st.s))) return rc;
    HoughParams hp;
    hp.min_line_length = min_line_length;
    hp.max_line_gap = max_line_gap;
    std::vector<std::vector<int32_t>> lines;
    if ((rc = edges_lines_device(in.as<uint8_t>(), 0, (int64_t)gray->cols * gray->channels, gray->rows, gray->cols,
                                 gray->channels, 1, hp, st.s, &lines)))
        return rc;
    std::vector<float> ang;
    std::vector<int32_t> cnt;
    line_angles(lines[0], &ang);
    if ((rc = vote_counts(ang, false, st.s, &cnt))) return rc;
    return select_hough_rs(ang, cnt, angle_out);
}

int omr_get_result_from_edges_detection(const omr_image *src, double edges_min_line_length, double edges_max_line_gap,
                                        double *angle, int32_t *status, double *candidates, int32_t cand_cap,
                                        int32_t *cand_len)
{
    clear_error();
    int rc = check_img(src);
    if (rc) return rc;
    if (!angle) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    DevBuf in;
    if ((rc = upload(src, &in, st.s))) return rc;
    HoughParams hp;
    hp.min_line_length = edges_min_line_length;
    hp.max_line_gap = edges_max_line_gap;
    std::vector<std::vector<int32_t>> lines;
    if ((rc = edges_lines_device(in.as<uint8_t>(), 0, (int64_t)src->cols * src->channels, src->rows, src->cols,
                                 src->channels, 1, hp, st.s, &lines)))
        return rc;
    std::vector<float> ang;
    std::vector<int32_t> cnt;
    line_angles(lines[0], &ang);
    if ((rc = vote_counts(ang, true, st.s, &cnt))) return rc;
    return select_omr_rs(ang, cnt, angle, status, candidates, cand_cap, cand_len);
}

int omr_edges_detection_batch_device(const uint8_t *d_scans, int32_t n, int64_t scan_stride_bytes, int32_t rows,
                                     int32_t cols, int32_t channels, int64_t step_bytes, double min_line_length,
                                     double max_line_gap, double *angles, int32_t *status, int32_t *n_lines, void *stream)
{
    clear_error();
    if (!d_scans || !angles || n <= 0) return fail(OMR_ERR_BADARG, "null pointer or empty batch");
    if (rows <= 0 || cols <= 0 || rows >= 32767 || cols >= 32767) return fail(OMR_ERR_ASSERT, "bad image shape");
    if (channels != 1 && channels != 3 && channels != 4) return fail(OMR_ERR_ASSERT, "1, 3 or 4 channels");
    if (step_bytes < (int64_t)cols * channels) return fail(OMR_ERR_BADARG, "step_bytes too small");
    int rc = have_device();
    if (rc) return rc;
    HoughParams hp;
    hp.min_line_length = min_line_length;
    hp.max_line_gap = max_line_gap;
    std::vector<std::vector<int32_t>> lines;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = edges_lines_device(d_scans, scan_stride_bytes, step_bytes, rows, cols, channels, n, hp, s, &lines))) return rc;
    // angles on host threads (libm atan2f, as the reference), one vote launch for the whole batch
    std::vector<std::vector<float>> ang((size_t)n);
    {
        const int nt = (int)std::max(1u, std::min<unsigned>((unsigned)n, std::min(std::thread::hardware_concurrency(), 32u)));
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&, t]() {
                for (int i = t; i < n; i += nt) line_angles(lines[i], &ang[i]);
            });
        for (auto &th : pool) th.join();
    }
    std::vector<int64_t> off((size_t)n + 1, 0);
    int max_n = 0;
    for (int i = 0; i < n; i++) {
        off[i + 1] = off[i] + (int64_t)ang[i].size();
        max_n = std::max(max_n, (int)ang[i].size());
    }
    std::vector<int32_t> cnt_all((size_t)off[n]);
    if (off[n] > 0) {
        std::vector<float> flat((size_t)off[n]);
        for (int i = 0; i < n; i++) std::copy(ang[i].begin(), ang[i].end(), flat.begin() + off[i]);
        DevBuf da, dc, doff;
        OMR_HIP(da.alloc(sizeof(float) * flat.size()));
        OMR_HIP(dc.alloc(sizeof(int32_t) * flat.size()));
        OMR_HIP(doff.alloc(sizeof(int64_t) * off.size()));
        OMR_HIP(hipMemcpyAsync(da.p, flat.data(), sizeof(float) * flat.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipMemcpyAsync(doff.p, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(launch_angle_votes_batch(da.as<float>(), doff.as<int64_t>(), n, max_n, 1, dc.as<int32_t>(), s));
        OMR_HIP(hipMemcpyAsync(cnt_all.data(), dc.p, sizeof(int32_t) * flat.size(), hipMemcpyDeviceToHost, s));
        OMR_HIP(hipStreamSynchronize(s));
    }
    for (int i = 0; i < n; i++) {
        if (n_lines) n_lines[i] = (int32_t)ang[i].size();
        if (ang[i].empty()) {  // the reference would panic (quirk B11): report "not a result"
            angles[i] = 0.0;
            if (status) status[i] = OMR_STATUS_NOT_A_RESULT;
            continue;
        }
        std::vector<int32_t> cnt(cnt_all.begin() + off[i], cnt_all.begin() + off[i + 1]);
        int32_t st = 0, nc = 0;
        if ((rc = select_omr_rs(ang[i], cnt, &angles[i], &st, nullptr, 0, &nc))) return rc;
        if (status) status[i] = st;
    }
    return OMR_OK;
}

// omr.rs:351-399
void omr_correct_default_decision(double proj_angle, int32_t proj_status, const double *proj_candidates, int32_t n_cand,
                                  double edges_angle, double *rotate_angle, int32_t *need_check)
{
    if (proj_status == OMR_STATUS_BELIEVED) {
        *rotate_angle = proj_angle;
        *need_check = 0;
    } else if (proj_status == OMR_STATUS_NEED_CHECK) {
        const bool far = fabs(proj_angle - edges_angle) >= 0.1;
        *rotate_angle = far ? edges_angle : proj_angle;
        *need_check = far ? 1 : 0;
    } else {
        int bi = -1;
        for (int i = 0; i < n_cand; i++)  // min_by keeps the first minimum
            if (bi < 0 || fabs(proj_candidates[i] - edges_angle) < fabs(proj_candidates[bi] - edges_angle)) bi = i;
        if (bi >= 0 && fabs(proj_candidates[bi] - edges_angle) < 0.05) {
            *rotate_angle = proj_candidates[bi];
            *need_check = 0;
        } else {
            *rotate_angle = edges_angle;
            *need_check = 1;
        }
    }
}

int omr_correct_default(const omr_image *src, uint16_t projection_max_angle, double projection_angle_step,
                        int32_t projection_max_width, int32_t projection_max_height, double hough_min_line_length,
                        double hough_max_line_gap, double *rotate_angle, int32_t *need_check, omr_image_owned *rotated)
{
    clear_error();
    if (!rotate_angle || !need_check) return fail(OMR_ERR_BADARG, "null output");
    int rc = check_img(src);
    if (rc) return rc;
    double pa = 0;
    int32_t pst = 0, pn = 0;
    int32_t n_half = 0;
    const int n_cand = omr_candidate_count(projection_max_angle, projection_angle_step, &n_half);  // candidates <= sweep angles
    if (n_cand <= 0) return fail(OMR_ERR_BADARG, "empty candidate range");
    std::vector<double> pc((size_t)n_cand + 1);
    if (src->channels == 2) return fail(OMR_ERR_ASSERT, "RGB2GRAY needs 3 or 4 channels");
    if ((rc = have_device())) return rc;
    // the sheet goes to the device once; projection, the Hough fallback and the final warp all read that copy
    HStream st;
    if ((rc = st.create())) return rc;
    DevBuf in;
    if ((rc = upload(src, &in, st.s))) return rc;
    const uint8_t *d_src = in.as<uint8_t>();
    rc = omr::result_from_projection_device(d_src, src->rows, src->cols, src->channels, projection_max_angle,
                                            projection_angle_step, projection_max_width, projection_max_height, st.s, &pa,
                                            &pst, pc.data(), (int32_t)pc.size(), &pn);
    if (rc) return rc;
    if (pst == OMR_STATUS_BELIEVED) {
        *rotate_angle = pa;
        *need_check = 0;
    } else {  // omr.rs:361-402
        HoughParams hp;
        hp.min_line_length = hough_min_line_length;
        hp.max_line_gap = hough_max_line_gap;
        std::vector<std::vector<int32_t>> lines;
        if ((rc = edges_lines_device(d_src, 0, (int64_t)src->cols * src->channels, src->rows, src->cols, src->channels, 1, hp,
                                     st.s, &lines)))
            return rc;
        std::vector<float> ang;
        std::vector<int32_t> cnt;
        line_angles(lines[0], &ang);
        if ((rc = vote_counts(ang, true, st.s, &cnt))) return rc;
        double ea = 0;
        int32_t est = 0, en = 0;
        if ((rc = select_omr_rs(ang, cnt, &ea, &est, nullptr, 0, &en))) return rc;
        omr_correct_default_decision(pa, pst, pc.data(), std::min<int32_t>(pn, (int32_t)pc.size()), ea, rotate_angle,
                                     need_check);
    }
    if (rotated) {  // omr.rs:404-445: CONTAIN canvas, nearest neighbour, white border, scale 1
        const uint8_t white[4] = {255, 255, 255, 0};
        rc = omr::rotate_device_to_host(d_src, src->rows, src->cols, src->channels, *rotate_angle, 1.0, OMR_INTER_NEAREST, white,
                                        OMR_CLIP_CONTAIN, st.s, rotated);
    }
    return rc;
}

}  // extern "C"

#ifdef OMR_RUNS_DEBUG
extern "C" int omr_debug_ppht_stamps(unsigned long long *out12, int reset)
{
    OMR_HIP(omr::debug_ppht_stamps(out12, reset != 0));
    return OMR_OK;
}
#endif
