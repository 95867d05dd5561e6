/*
 * oracle_hough.c -- CPU restatement of the reference's Hough-line deskew path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  "parity unpinned": the reference holds no golden
 * vectors for this path and the arithmetic lives in the un-vendored OpenCV 4.6.0
 * (modules/imgproc/src/canny.cpp, hough.cpp, deriv.cpp; core/src/rand.cpp, RNG in
 * core/include/opencv2/core.hpp / operations.hpp), restated here from the published algorithm:
 *
 *   Canny(src 8UC1|8UC3, 50, 150, apertureSize 3, L2gradient false)
 *     call sites packages/lib/src/hough.rs:27, packages/lib/src/omr.rs:239, :323-330
 *   HoughLinesP(edges, rho 1, theta pi/180, threshold 0, minLineLength, maxLineGap)
 *     call sites packages/lib/src/hough.rs:31-43, packages/lib/src/omr.rs:245-253
 *   line angles + "% 45" + the O(n^2) +-0.1 degree vote
 *     packages/lib/src/hough.rs:50-92 (f32, first maximum) and
 *     packages/lib/src/omr.rs:257-301 (f64 vector, candidates + status)
 *   the decision of correct_default, packages/lib/src/omr.rs:351-399
 *
 * Built -ffp-contract=off; atan2f / fmodf / cos / sin come from the host libm, like the
 * reference's (Rust f32::atan2 -> libm atan2f).
 */
#include "oracle_hough.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Sobel 3x3, CV_16S, BORDER_REPLICATE (deriv.cpp; what Canny asks for) ----------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_sobel3_16s(const uint8_t *src, int rows, int cols, int cn, int64_t sstep, int16_t *dx, int16_t *dy)
{
    for (int y = 0; y < rows; y++) {
        const uint8_t *r0 = src + (int64_t)clampi(y - 1, 0, rows - 1) * sstep;
        const uint8_t *r1 = src + (int64_t)y * sstep;
        const uint8_t *r2 = src + (int64_t)clampi(y + 1, 0, rows - 1) * sstep;
        for (int x = 0; x < cols; x++) {
            const int xl = clampi(x - 1, 0, cols - 1) * cn, xc = x * cn, xr = clampi(x + 1, 0, cols - 1) * cn;
            for (int c = 0; c < cn; c++) {
                const int a00 = r0[xl + c], a01 = r0[xc + c], a02 = r0[xr + c];
                const int a10 = r1[xl + c], a12 = r1[xr + c];
                const int a20 = r2[xl + c], a21 = r2[xc + c], a22 = r2[xr + c];
                const int64_t o = ((int64_t)y * cols + x) * cn + c;
                dx[o] = (int16_t)((a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20));
                dy[o] = (int16_t)((a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02));
            }
        }
    }
}

/* ---- Canny (canny.cpp), L1 gradient, aperture 3 -------------------------------------------- */
int orc_canny(const uint8_t *src, int rows, int cols, int cn, int64_t sstep, double low_thresh, double high_thresh,
              uint8_t *dst, int64_t dstep)
{
    if (!src || !dst || rows <= 0 || cols <= 0 || (cn != 1 && cn != 3 && cn != 4)) return -215;
    if (low_thresh > high_thresh) {
        const double t = low_thresh;
        low_thresh = high_thresh;
        high_thresh = t;
    }
    const int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
    const int64_t n = (int64_t)rows * cols;
    int16_t *dx = (int16_t *)malloc(sizeof(int16_t) * (size_t)n * cn);
    int16_t *dy = (int16_t *)malloc(sizeof(int16_t) * (size_t)n * cn);
    /* magnitude with a zero frame: mag[(y + 1) * (cols + 2) + x + 1] */
    const int mstep = cols + 2;
    int32_t *mag = (int32_t *)calloc((size_t)(rows + 2) * mstep, sizeof(int32_t));
    int16_t *gx = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    int16_t *gy = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    /* map with a frame of 1 ("cannot be an edge"): 0 candidate, 1 no edge, 2 edge */
    uint8_t *map = (uint8_t *)malloc((size_t)(rows + 2) * mstep);
    uint8_t **stack = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)(n > 0 ? n : 1));
    if (!dx || !dy || !mag || !gx || !gy || !map || !stack) {
        free(dx), free(dy), free(mag), free(gx), free(gy), free(map), free(stack);
        return -4;
    }
    orc_sobel3_16s(src, rows, cols, cn, sstep, dx, dy);
    /* per pixel: the channel with the largest |dx| + |dy| (first one on ties) */
    for (int64_t i = 0; i < n; i++) {
        int best = 0, bm = abs(dx[i * cn]) + abs(dy[i * cn]);
        for (int c = 1; c < cn; c++) {
            const int m = abs(dx[i * cn + c]) + abs(dy[i * cn + c]);
            if (m > bm) bm = m, best = c;
        }
        gx[i] = dx[i * cn + best];
        gy[i] = dy[i * cn + best];
        mag[((i / cols) + 1) * mstep + (i % cols) + 1] = bm;
    }
    memset(map, 1, (size_t)(rows + 2) * mstep);
    size_t sp = 0;
    const int TG22 = 13573; /* (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5) */
    for (int y = 0; y < rows; y++) {
        const int32_t *mp = mag + (int64_t)y * mstep + 1;       /* previous row */
        const int32_t *ma = mag + (int64_t)(y + 1) * mstep + 1; /* this row */
        const int32_t *mn = mag + (int64_t)(y + 2) * mstep + 1; /* next row */
        uint8_t *pm = map + (int64_t)(y + 1) * mstep + 1;
        for (int x = 0; x < cols; x++) {
            const int m = ma[x];
            int edge = 0;
            if (m > low) {
                const int xs = gx[(int64_t)y * cols + x], ys = gy[(int64_t)y * cols + x];
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * TG22;
                if (ay < tg22x) {
                    edge = m > ma[x - 1] && m >= ma[x + 1];
                } else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        edge = m > mp[x] && m >= mn[x];
                    } else {
                        /* gradient along a diagonal: same signs -> up-left / down-right */
                        const int s = (xs ^ ys) < 0 ? -1 : 1;
                        edge = m > mp[x - s] && m > mn[x + s];
                    }
                }
            }
            if (edge) {
                if (m > high) {
                    pm[x] = 2;
                    stack[sp++] = pm + x;
                } else {
                    pm[x] = 0;
                }
            } else {
                pm[x] = 1;
            }
        }
    }
    /* hysteresis: candidates 8-connected to an edge become edges */
    while (sp > 0) {
        uint8_t *m = stack[--sp];
        static const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
        static const int dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        for (int k = 0; k < 8; k++) {
            uint8_t *q = m + dys[k] * mstep + dxs[k];
            if (*q == 0) {
                *q = 2;
                stack[sp++] = q;
            }
        }
    }
    for (int y = 0; y < rows; y++) {
        const uint8_t *pm = map + (int64_t)(y + 1) * mstep + 1;
        uint8_t *d = dst + (int64_t)y * dstep;
        for (int x = 0; x < cols; x++) d[x] = (uint8_t)(-(pm[x] >> 1));
    }
    free(dx), free(dy), free(mag), free(gx), free(gy), free(map), free(stack);
    return 0;
}

/* ---- cv::RNG (multiply-with-carry), operations.hpp ------------------------------------------ */
static inline uint32_t rng_next(uint64_t *state)
{
    *state = (uint64_t)(uint32_t)*state * 4164903690u + (uint32_t)(*state >> 32);
    return (uint32_t)*state;
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

void orc_hough_trigtab(double theta_in, double rho_in, int *numangle_out, float *trigtab)
{
    const float theta = (float)theta_in, irho = 1.0f / (float)rho_in;
    const int numangle = cv_round_d(3.1415926535897932384626433832795 / theta);
    *numangle_out = numangle;
    if (!trigtab) return;
    for (int n = 0; n < numangle; n++) {
        trigtab[n * 2] = (float)(cos((double)n * theta) * irho);
        trigtab[n * 2 + 1] = (float)(sin((double)n * theta) * irho);
    }
}

/* statistics of the last orc_hough_lines_p call of this thread: points, points still set when
 * drawn (= line walks), points cleared by walks, accumulator decrements (sizing aid for the GPU build) */
static _Thread_local int64_t g_stats[4];
void orc_hough_last_stats(int64_t out[4]) { memcpy(out, g_stats, sizeof g_stats); }

/* hough.cpp HoughLinesProbabilistic.  lines: cap x 4 ints (x0, y0, x1, y1). */
int orc_hough_lines_p(const uint8_t *image, int height, int width, int64_t step, double rho_in, double theta_in,
                      int threshold, double min_line_length, double max_line_gap, int32_t *lines, int cap,
                      int *n_lines)
{
    if (!image || height <= 0 || width <= 0 || !n_lines) return -215;
    const float rho = (float)rho_in;
    const int lineLength = cv_round_d(min_line_length), lineGap = cv_round_d(max_line_gap);
    uint64_t rng = (uint64_t)-1;
    int numangle;
    orc_hough_trigtab(theta_in, rho_in, &numangle, NULL);
    const int numrho = cv_round_f((float)((width + height) * 2 + 1) / rho);
    float *ttab = (float *)malloc(sizeof(float) * 2 * (size_t)numangle);
    int32_t *accum = (int32_t *)calloc((size_t)numangle * numrho, sizeof(int32_t));
    uint8_t *mask = (uint8_t *)malloc((size_t)height * width);
    int32_t *nz = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)height * width);
    if (!ttab || !accum || !mask || !nz) {
        free(ttab), free(accum), free(mask), free(nz);
        return -4;
    }
    orc_hough_trigtab(theta_in, rho_in, &numangle, ttab);
    int count = 0;
    for (int y = 0; y < height; y++) {
        const uint8_t *d = image + (int64_t)y * step;
        for (int x = 0; x < width; x++) {
            if (d[x]) {
                mask[(int64_t)y * width + x] = 1;
                nz[2 * count] = x;
                nz[2 * count + 1] = y;
                count++;
            } else {
                mask[(int64_t)y * width + x] = 0;
            }
        }
    }
    int nl = 0;
    const int shift = 16;
    g_stats[0] = count, g_stats[1] = g_stats[2] = g_stats[3] = 0;
    for (; count > 0; count--) {
        const int idx = (int)(rng_next(&rng) % (uint32_t)count);
        int max_val = threshold - 1, max_n = 0;
        const int j = nz[2 * idx], i = nz[2 * idx + 1];
        int line_end[2][2] = {{0, 0}, {0, 0}}; /* [k] = (x, y) */
        nz[2 * idx] = nz[2 * (count - 1)];
        nz[2 * idx + 1] = nz[2 * (count - 1) + 1];
        if (!mask[(int64_t)i * width + j]) continue;
        for (int n = 0; n < numangle; n++) {
            int r = cv_round_f((float)j * ttab[n * 2] + (float)i * ttab[n * 2 + 1]);
            r += (numrho - 1) / 2;
            const int val = ++accum[(int64_t)n * numrho + r];
            if (max_val < val) {
                max_val = val;
                max_n = n;
            }
        }
        if (max_val < threshold) continue;
        g_stats[1]++;
        const float a = -ttab[max_n * 2 + 1], b = ttab[max_n * 2];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = cv_round_f(b * (float)(1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = cv_round_f(a * (float)(1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        for (int k = 0; k < 2; k++) {
            int gap = 0, x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) dx = -dx, dy = -dy;
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) {
                    j1 = x;
                    i1 = y >> shift;
                } else {
                    j1 = x >> shift;
                    i1 = y;
                }
                if (j1 < 0 || j1 >= width || i1 < 0 || i1 >= height) break;
                if (mask[(int64_t)i1 * width + j1]) {
                    gap = 0;
                    line_end[k][1] = i1;
                    line_end[k][0] = j1;
                } else if (++gap > lineGap) {
                    break;
                }
            }
        }
        const int good_line = abs(line_end[1][0] - line_end[0][0]) >= lineLength ||
                              abs(line_end[1][1] - line_end[0][1]) >= lineLength;
        for (int k = 0; k < 2; k++) {
            int x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) dx = -dx, dy = -dy;
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) {
                    j1 = x;
                    i1 = y >> shift;
                } else {
                    j1 = x >> shift;
                    i1 = y;
                }
                uint8_t *m = mask + (int64_t)i1 * width + j1;
                if (*m) {
                    g_stats[2]++;
                    if (good_line) {
                        g_stats[3]++;
                        for (int n = 0; n < numangle; n++) {
                            int r = cv_round_f((float)j1 * ttab[n * 2] + (float)i1 * ttab[n * 2 + 1]);
                            r += (numrho - 1) / 2;
                            accum[(int64_t)n * numrho + r]--;
                        }
                    }
                    *m = 0;
                }
                if (i1 == line_end[k][1] && j1 == line_end[k][0]) break;
            }
        }
        if (good_line) {
            if (lines && nl < cap) {
                lines[4 * nl] = line_end[0][0];
                lines[4 * nl + 1] = line_end[0][1];
                lines[4 * nl + 2] = line_end[1][0];
                lines[4 * nl + 3] = line_end[1][1];
            }
            nl++;
        }
    }
    *n_lines = nl;
    free(ttab), free(accum), free(mask), free(nz);
    return 0;
}

/* hough.rs:50-68 / omr.rs:257-267: angle of a segment in degrees, f32 arithmetic, "% 45.0". */
float orc_line_angle_f32(const int32_t l[4])
{
    const float x1 = (float)l[0], y1 = (float)l[1], x2 = (float)l[2], y2 = (float)l[3];
    const float pi32 = 3.14159274101257324f; /* std::f32::consts::PI */
    float angle = atan2f(y2 - y1, x2 - x1) * 180.0f / pi32;
    angle = fmodf(angle, 45.0f);
    return angle;
}

/* hough.rs:70-92: f32 vector, range 0.1f32, strict ">" keeps the first maximum. */
int orc_vote_hough_rs(const float *angles, int n, double *angle_out)
{
    if (n <= 0) return -215; /* angles[0] panics (quirk B11) */
    const float range = 0.1f;
    float target = angles[0];
    int best = 0;
    for (int i = 0; i < n; i++) {
        int count = 0;
        for (int j = 0; j < n; j++)
            if (fabsf(angles[i] - angles[j]) < range) count++;
        if (count > best) {
            target = angles[i];
            best = count;
        }
    }
    *angle_out = (double)target;
    return 0;
}

/* omr.rs:268-301: f64 vector (f32 angles widened), range 0.1 f64, candidates + status. */
int orc_vote_omr_rs(const float *angles32, int n, double *angle_out, int *status, double *candidates, int cand_cap,
                    int *cand_len)
{
    if (n <= 0) return -215; /* angles[0] panics (quirk B11) */
    const double range = 0.1;
    double target = (double)angles32[0];
    int best = 0, nc = 0;
    for (int i = 0; i < n; i++) {
        const double ai = (double)angles32[i];
        int count = 0;
        for (int j = 0; j < n; j++)
            if (fabs(ai - (double)angles32[j]) < range) count++;
        if (count > best) {
            target = ai;
            best = count;
            nc = 0;
            if (candidates && nc < cand_cap) candidates[nc] = target;
            nc = 1;
        } else if (count == best) {
            if (candidates && nc < cand_cap) candidates[nc] = ai;
            nc++;
        }
    }
    *angle_out = target;
    if (cand_len) *cand_len = nc;
    if (status) *status = nc == 0 ? 2 : (nc == 1 ? 0 : 1);
    return 0;
}

int orc_get_angle_with_hough(const uint8_t *gray, int rows, int cols, int cn, int64_t step, double min_line_length,
                             double max_line_gap, double *angle_out, int *n_lines_out)
{
    uint8_t *edges = (uint8_t *)malloc((size_t)rows * cols);
    if (!edges) return -4;
    int rc = orc_canny(gray, rows, cols, cn, step, 50.0, 150.0, edges, cols);
    int n = 0;
    if (rc == 0) rc = orc_hough_lines_p(edges, rows, cols, cols, 1.0, 3.14159265358979323846 / 180.0, 0, min_line_length,
                                        max_line_gap, NULL, 0, &n);
    int32_t *lines = NULL;
    float *ang = NULL;
    if (rc == 0 && n > 0) {
        lines = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)n);
        ang = (float *)malloc(sizeof(float) * (size_t)n);
        if (!lines || !ang) rc = -4;
        if (rc == 0) rc = orc_hough_lines_p(edges, rows, cols, cols, 1.0, 3.14159265358979323846 / 180.0, 0,
                                            min_line_length, max_line_gap, lines, n, &n);
        for (int i = 0; rc == 0 && i < n; i++) ang[i] = orc_line_angle_f32(lines + 4 * i);
    }
    if (n_lines_out) *n_lines_out = n;
    if (rc == 0) rc = orc_vote_hough_rs(ang, n, angle_out);
    free(edges), free(lines), free(ang);
    return rc;
}

int orc_get_result_from_edges_detection(const uint8_t *src, int rows, int cols, int cn, int64_t step,
                                        double min_line_length, double max_line_gap, double *angle, int *status,
                                        double *candidates, int cand_cap, int *cand_len, int *n_lines_out)
{
    uint8_t *edges = (uint8_t *)malloc((size_t)rows * cols);
    if (!edges) return -4;
    int rc = orc_canny(src, rows, cols, cn, step, 50.0, 150.0, edges, cols);
    int n = 0;
    if (rc == 0) rc = orc_hough_lines_p(edges, rows, cols, cols, 1.0, 3.14159265358979323846 / 180.0, 0, min_line_length,
                                        max_line_gap, NULL, 0, &n);
    int32_t *lines = NULL;
    float *ang = NULL;
    if (rc == 0 && n > 0) {
        lines = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)n);
        ang = (float *)malloc(sizeof(float) * (size_t)n);
        if (!lines || !ang) rc = -4;
        if (rc == 0) rc = orc_hough_lines_p(edges, rows, cols, cols, 1.0, 3.14159265358979323846 / 180.0, 0,
                                            min_line_length, max_line_gap, lines, n, &n);
        for (int i = 0; rc == 0 && i < n; i++) ang[i] = orc_line_angle_f32(lines + 4 * i);
    }
    if (n_lines_out) *n_lines_out = n;
    if (rc == 0) rc = orc_vote_omr_rs(ang, n, angle, status, candidates, cand_cap, cand_len);
    free(edges), free(lines), free(ang);
    return rc;
}

/* omr.rs:351-399: which angle correct_default rotates by, and whether the sheet needs a check. */
void orc_correct_default_decision(double proj_angle, int proj_status, const double *proj_candidates, int n_cand,
                                  double edges_angle, double *rotate_angle, int *need_check)
{
    if (proj_status == 0) { /* Believed */
        *rotate_angle = proj_angle;
        *need_check = 0;
    } else if (proj_status == 1) { /* NeedCheck */
        if (fabs(proj_angle - edges_angle) >= 0.1) {
            *rotate_angle = edges_angle;
            *need_check = 1;
        } else {
            *rotate_angle = proj_angle;
            *need_check = 0;
        }
    } else { /* NotAResult: the projection candidate nearest to the edges angle (first minimum) */
        if (n_cand <= 0) {
            *rotate_angle = edges_angle;
            *need_check = 1;
            return;
        }
        int bi = 0;
        for (int i = 1; i < n_cand; i++)
            if (fabs(proj_candidates[i] - edges_angle) < fabs(proj_candidates[bi] - edges_angle)) bi = i;
        if (fabs(proj_candidates[bi] - edges_angle) < 0.05) {
            *rotate_angle = proj_candidates[bi];
            *need_check = 0;
        } else {
            *rotate_angle = edges_angle;
            *need_check = 1;
        }
    }
}
