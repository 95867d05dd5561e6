/*
 * oracle.c -- CPU restatement of the reference's projection-std-dev deskew hot path.
 * TEST INFRASTRUCTURE ONLY; see oracle.h for the usage rule and the "parity unpinned" note.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: OpenCV's baseline x86 build and rustc emit no FMA contraction,
 * and several expressions below ((M1*y + M2)*1024, sum + d*d) change in the last bit if fused.
 *
 * Citations are file:line under /root/reference unless marked [OpenCV 4.6.0], which names the
 * upstream function whose published algorithm is restated (the source is not in the tree).
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_ASSERT (-215) /* cv::Error::StsAssert */
#define ORC_ERR_BADARG (-5)   /* cv::Error::StsBadArg */
#define ORC_ERR_NOMEM (-4)    /* cv::Error::StsNoMem */

#define ORC_PI 3.1415926535897932384626433832795 /* CV_PI */

/* cvRound(double) == saturate_cast<int>(double): lrint under the default rounding mode
 * (round-half-to-even).  [OpenCV 4.6.0 core/fast_math.hpp] */
static inline int orc_cv_round(double v) { return (int)lrint(v); }
static inline int orc_cv_roundf(float v) { return (int)lrintf(v); }
static inline int orc_cv_floor(double v)
{
    int i = (int)v;
    return i - (i > v);
}
static inline int orc_cv_ceil(double v)
{
    int i = (int)v;
    return i + (i < v);
}
static inline short orc_sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint8_t orc_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* ------------------------------------------------------------------------------------ */
/* [OpenCV 4.6.0 imgwarp.cpp getRotationMatrix2D_]; Appendix A.1 of SURVEY.md.          */
void orc_get_rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6])
{
    double angle = angle_deg * (ORC_PI / 180);
    double alpha = cos(angle) * scale;
    double beta = sin(angle) * scale;
    M[0] = alpha;
    M[1] = beta;
    M[2] = (1 - alpha) * (double)cx - beta * (double)cy;
    M[3] = -beta;
    M[4] = alpha;
    M[5] = beta * (double)cx + (1 - alpha) * (double)cy;
}

/* [OpenCV 4.6.0 imgwarp.cpp cv::warpAffine, !(flags & WARP_INVERSE_MAP) branch]. */
void orc_invert_affine(const double Min[6], double M[6])
{
    memcpy(M, Min, 6 * sizeof(double));
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11;
    M[1] *= -D;
    M[3] *= -D;
    M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1;
    M[5] = b2;
}

/* [OpenCV 4.6.0 imgwarp.cpp hal::warpAffine + WarpAffineInvoker]: AB_BITS = 10. */
void orc_warp_tables(const double M[6], int dcols, int drows, int round_delta,
                     int32_t *adelta, int32_t *bdelta, int32_t *X0, int32_t *Y0)
{
    const int AB_SCALE = 1 << 10;
    for (int x = 0; x < dcols; x++) {
        adelta[x] = orc_cv_round(M[0] * x * AB_SCALE);
        bdelta[x] = orc_cv_round(M[3] * x * AB_SCALE);
    }
    for (int y = 0; y < drows; y++) {
        X0[y] = orc_cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        Y0[y] = orc_cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    }
}

static int orc_check_dims(int srows, int scols, int drows, int dcols)
{
    /* [OpenCV 4.6.0 remap]: CV_Assert on SHRT_MAX for every dimension; also non-empty. */
    if (srows <= 0 || scols <= 0 || drows <= 0 || dcols <= 0) return ORC_ERR_ASSERT;
    if (srows >= 32767 || scols >= 32767 || drows >= 32767 || dcols >= 32767) return ORC_ERR_ASSERT;
    return ORC_OK;
}

/* [OpenCV 4.6.0 imgwarp.cpp WarpAffineInvoker (INTER_NEAREST) + remapNearest, BORDER_CONSTANT];
 * Appendix A.2.  Call sites transfer.rs:477-485, omr.rs:165-173,435-443. */
int orc_warp_affine_nn(const uint8_t *src, int srows, int scols, int cn, int64_t sstep,
                       uint8_t *dst, int drows, int dcols, int64_t dstep,
                       const double Mfwd[6], const uint8_t border[4])
{
    int rc = orc_check_dims(srows, scols, drows, dcols);
    if (rc) return rc;
    if (cn < 1 || cn > 4) return ORC_ERR_ASSERT;
    double M[6];
    orc_invert_affine(Mfwd, M);
    int32_t *tab = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * dcols + 2 * drows));
    if (!tab) return ORC_ERR_NOMEM;
    int32_t *adelta = tab, *bdelta = tab + dcols, *X0 = bdelta + dcols, *Y0 = X0 + drows;
    orc_warp_tables(M, dcols, drows, 512, adelta, bdelta, X0, Y0);
    for (int y = 0; y < drows; y++) {
        uint8_t *D = dst + (int64_t)y * dstep;
        for (int x = 0; x < dcols; x++) {
            int X = (X0[y] + adelta[x]) >> 10; /* arithmetic shift = floor */
            int Y = (Y0[y] + bdelta[x]) >> 10;
            int sx = orc_sat_short(X), sy = orc_sat_short(Y);
            if ((unsigned)sx < (unsigned)scols && (unsigned)sy < (unsigned)srows) {
                const uint8_t *S = src + (int64_t)sy * sstep + (int64_t)sx * cn;
                for (int k = 0; k < cn; k++) D[x * cn + k] = S[k];
            } else {
                for (int k = 0; k < cn; k++) D[x * cn + k] = border[k];
            }
        }
    }
    free(tab);
    return ORC_OK;
}

/* [OpenCV 4.6.0 imgwarp.cpp WarpAffineInvoker (INTER_LINEAR) + remapBilinear<FixedPtCast<int,uchar,15>>];
 * Appendix A.8.  INTER_BITS = 5, INTER_TAB_SIZE = 32, INTER_REMAP_COEF_BITS = 15.  For the
 * bilinear table every weight (a/32)(b/32)*32768 is an exact integer, so the table's
 * sum-to-32768 fix-up never fires. */
int orc_warp_affine_linear(const uint8_t *src, int srows, int scols, int cn, int64_t sstep,
                           uint8_t *dst, int drows, int dcols, int64_t dstep,
                           const double Mfwd[6], const uint8_t border[4])
{
    int rc = orc_check_dims(srows, scols, drows, dcols);
    if (rc) return rc;
    if (cn < 1 || cn > 4) return ORC_ERR_ASSERT;
    double M[6];
    orc_invert_affine(Mfwd, M);
    int32_t *tab = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * dcols + 2 * drows));
    if (!tab) return ORC_ERR_NOMEM;
    int32_t *adelta = tab, *bdelta = tab + dcols, *X0 = bdelta + dcols, *Y0 = X0 + drows;
    orc_warp_tables(M, dcols, drows, 16, adelta, bdelta, X0, Y0);
    const int width1 = scols - 1 > 0 ? scols - 1 : 0, height1 = srows - 1 > 0 ? srows - 1 : 0;
    for (int y = 0; y < drows; y++) {
        uint8_t *D = dst + (int64_t)y * dstep;
        for (int x = 0; x < dcols; x++) {
            int X = (X0[y] + adelta[x]) >> 5;
            int Y = (Y0[y] + bdelta[x]) >> 5;
            int sx = orc_sat_short(X >> 5), sy = orc_sat_short(Y >> 5);
            int fx = X & 31, fy = Y & 31;
            int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
            if ((unsigned)sx < (unsigned)width1 && (unsigned)sy < (unsigned)height1) {
                const uint8_t *S = src + (int64_t)sy * sstep + (int64_t)sx * cn;
                for (int k = 0; k < cn; k++) {
                    int v = S[k] * w[0] + S[cn + k] * w[1] + S[sstep + k] * w[2] + S[sstep + cn + k] * w[3];
                    D[x * cn + k] = orc_sat_u8((v + (1 << 14)) >> 15);
                }
            } else if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) {
                for (int k = 0; k < cn; k++) D[x * cn + k] = border[k];
            } else {
                int sx0 = sx, sx1 = sx + 1, sy0 = sy, sy1 = sy + 1;
                int in_x0 = sx0 >= 0 && sx0 < scols, in_x1 = sx1 >= 0 && sx1 < scols;
                int in_y0 = sy0 >= 0 && sy0 < srows, in_y1 = sy1 >= 0 && sy1 < srows;
                for (int k = 0; k < cn; k++) {
                    int v0 = in_x0 && in_y0 ? src[(int64_t)sy0 * sstep + sx0 * cn + k] : border[k];
                    int v1 = in_x1 && in_y0 ? src[(int64_t)sy0 * sstep + sx1 * cn + k] : border[k];
                    int v2 = in_x0 && in_y1 ? src[(int64_t)sy1 * sstep + sx0 * cn + k] : border[k];
                    int v3 = in_x1 && in_y1 ? src[(int64_t)sy1 * sstep + sx1 * cn + k] : border[k];
                    int v = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
                    D[x * cn + k] = orc_sat_u8((v + (1 << 14)) >> 15);
                }
            }
        }
    }
    free(tab);
    return ORC_OK;
}

/* [OpenCV 4.6.0 thresh.cpp, THRESH_BINARY 8U]: dst = src > thresh ? maxval : 0; Appendix A.3.
 * Call sites transfer.rs:294-301 (127, 255), omr.rs:129-139. */
void orc_threshold_binary(const uint8_t *src, int rows, int cols, int64_t sstep,
                          uint8_t *dst, int64_t dstep, int thresh, int maxval)
{
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            dst[(int64_t)y * dstep + x] = src[(int64_t)y * sstep + x] > thresh ? (uint8_t)maxval : 0;
}

/* [OpenCV 4.6.0 color_rgb.simd.hpp RGB2Gray<uchar>]: (c0*R2Y + c1*G2Y + c2*B2Y + (1<<14)) >> 15 with
 * R2Y=9798, G2Y=19235, B2Y=3735 applied in memory order for COLOR_RGB2GRAY; Appendix A.5.
 * Call sites transfer.rs:283-290, omr.rs:88-92 (on imread's BGR data: quirk B8 preserved). */
void orc_rgb2gray(const uint8_t *src, int rows, int cols, int cn, int64_t sstep,
                  uint8_t *dst, int64_t dstep)
{
    for (int y = 0; y < rows; y++) {
        const uint8_t *S = src + (int64_t)y * sstep;
        uint8_t *D = dst + (int64_t)y * dstep;
        for (int x = 0; x < cols; x++, S += cn)
            D[x] = (uint8_t)((S[0] * 9798 + S[1] * 19235 + S[2] * 3735 + (1 << 14)) >> 15);
    }
}

/* [OpenCV 4.6.0 morph: getStructuringElement(MORPH_ELLIPSE,3x3) = cross; erode with
 * BORDER_CONSTANT + morphologyDefaultBorderValue (= +inf for erode); a non-rectangular kernel
 * keeps `iterations` successive passes]; Appendix A.6.  Call site omr.rs:98-112. */
void orc_erode_cross3(const uint8_t *src, int rows, int cols, int64_t sstep,
                      uint8_t *dst, int64_t dstep, int iterations)
{
    uint8_t *a = (uint8_t *)malloc((size_t)rows * cols), *b = (uint8_t *)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; y++) memcpy(a + (size_t)y * cols, src + (int64_t)y * sstep, (size_t)cols);
    for (int it = 0; it < iterations; it++) {
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) {
                int m = a[(size_t)y * cols + x];
                if (y > 0 && a[(size_t)(y - 1) * cols + x] < m) m = a[(size_t)(y - 1) * cols + x];
                if (y + 1 < rows && a[(size_t)(y + 1) * cols + x] < m) m = a[(size_t)(y + 1) * cols + x];
                if (x > 0 && a[(size_t)y * cols + x - 1] < m) m = a[(size_t)y * cols + x - 1];
                if (x + 1 < cols && a[(size_t)y * cols + x + 1] < m) m = a[(size_t)y * cols + x + 1];
                b[(size_t)y * cols + x] = (uint8_t)m;
            }
        uint8_t *t = a;
        a = b;
        b = t;
    }
    for (int y = 0; y < rows; y++) memcpy(dst + (int64_t)y * dstep, a + (size_t)y * cols, (size_t)cols);
    free(a);
    free(b);
}

/* [OpenCV 4.6.0 resize.cpp computeResizeAreaTab]. */
typedef struct {
    int si, di;
    float alpha;
} orc_decimate_alpha;

static int orc_area_tab(int ssize, int dsize, int cn, double scale, orc_decimate_alpha *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = orc_cv_ceil(fsx1), sx2 = orc_cv_floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx * cn;
            tab[k].si = (sx1 - 1) * cn;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx * cn;
            tab[k].si = sx * cn;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            double m = fsx2 - sx2 < 1. ? fsx2 - sx2 : 1.;
            m = m < cellWidth ? m : cellWidth;
            tab[k].di = dx * cn;
            tab[k].si = sx2 * cn;
            tab[k++].alpha = (float)(m / cellWidth);
        }
    }
    return k;
}

/* [OpenCV 4.6.0 resize.cpp hal::resize -> resizeGeneric_ with HResizeLinear<uchar,int,short,2048> and
 * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>], 8U, cn channels.  Used for
 *   - INTER_LINEAR (area_mode = 0): scale_self with scale > 1, transfer.rs:66-91;
 *   - INTER_AREA when either axis enlarges (area_mode = 1): "true area interpolation is only implemented
 *     for scale_x >= 1 && scale_y >= 1, in other cases it is emulated using some variant of bilinear
 *     interpolation" -- path 2 has no clamp on its scale, omr.rs:60-82,114-126 (quirk B7).
 * Coefficients: float fx -> saturate_cast<short>(c * 2048) (11 bits, round half to even); horizontal pass
 * S[sx]*a0 + S[sx+cn]*a1 (S[sx]*2048 right of xmax); vertical pass
 * (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; source rows sy, sy+1 clipped into the image.
 * IPP is disabled upstream for 8U resize (IPP_DISABLE_RESIZE_8U), the SIMD forms compute the same integers. */
static inline short orc_sat_s16(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

int orc_resize_linear(const uint8_t *src, int srows, int scols, int cn, int64_t sstep, uint8_t *dst, int drows,
                      int dcols, int64_t dstep, int area_mode)
{
    if (srows <= 0 || scols <= 0 || drows <= 0 || dcols <= 0 || cn < 1 || cn > 4) return ORC_ERR_ASSERT;
    const double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dcols), *yofs = (int *)malloc(sizeof(int) * (size_t)drows);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dcols), *ibeta = (short *)malloc(sizeof(short) * 2 * (size_t)drows);
    int xmax = dcols;
    for (int dx = 0; dx < dcols; dx++) {
        float fx;
        int sx;
        if (!area_mode) {
            fx = (float)((dx + 0.5) * scale_x - 0.5);
            sx = orc_cv_floor(fx);
            fx -= sx;
        } else {
            sx = orc_cv_floor(dx * scale_x);
            fx = (float)((dx + 1) - (sx + 1) * inv_scale_x);
            fx = fx <= 0 ? 0.f : fx - orc_cv_floor(fx);
        }
        if (sx < 0) fx = 0, sx = 0; /* ksize2 - 1 == 0: xmin plays no role in the linear kernel */
        if (sx + 1 >= scols) {
            xmax = xmax < dx ? xmax : dx;
            if (sx >= scols - 1) fx = 0, sx = scols - 1;
        }
        xofs[dx] = sx * cn;
        ialpha[dx * 2] = orc_sat_s16(orc_cv_roundf((1.f - fx) * 2048));
        ialpha[dx * 2 + 1] = orc_sat_s16(orc_cv_roundf(fx * 2048));
    }
    for (int dy = 0; dy < drows; dy++) {
        float fy;
        int sy;
        if (!area_mode) {
            fy = (float)((dy + 0.5) * scale_y - 0.5);
            sy = orc_cv_floor(fy);
            fy -= sy;
        } else {
            sy = orc_cv_floor(dy * scale_y);
            fy = (float)((dy + 1) - (sy + 1) * inv_scale_y);
            fy = fy <= 0 ? 0.f : fy - orc_cv_floor(fy);
        }
        yofs[dy] = sy;
        ibeta[dy * 2] = orc_sat_s16(orc_cv_roundf((1.f - fy) * 2048));
        ibeta[dy * 2 + 1] = orc_sat_s16(orc_cv_roundf(fy * 2048));
    }
    for (int dy = 0; dy < drows; dy++) {
        int sy0 = yofs[dy], sy1 = sy0 + 1; /* clip(sy0 - ksize2 + 1 + k, 0, ssize.height), k = 0, 1 */
        sy0 = sy0 < 0 ? 0 : (sy0 < srows ? sy0 : srows - 1);
        sy1 = sy1 < 0 ? 0 : (sy1 < srows ? sy1 : srows - 1);
        const uint8_t *S0 = src + (int64_t)sy0 * sstep, *S1 = src + (int64_t)sy1 * sstep;
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *D = dst + (int64_t)dy * dstep;
        for (int dx = 0; dx < dcols; dx++)
            for (int k = 0; k < cn; k++) {
                const int sx = xofs[dx] + k;
                int h0, h1;
                if (dx < xmax) {
                    const int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
                    h0 = S0[sx] * a0 + S0[sx + cn] * a1;
                    h1 = S1[sx] * a0 + S1[sx + cn] * a1;
                } else {
                    h0 = S0[sx] * 2048;
                    h1 = S1[sx] * 2048;
                }
                D[dx * cn + k] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
    }
    free(xofs);
    free(yofs);
    free(ialpha);
    free(ibeta);
    return ORC_OK;
}

/* [OpenCV 4.6.0 resize.cpp hal::resize, INTER_AREA, 8U]: integer factors -> resizeAreaFast_
 * (2x2 special case (a+b+c+d+2)>>2, otherwise saturate_cast<uchar>(sum * (1.f/area)));
 * other shrink factors -> resizeArea_ with float tables; Appendix A.7.
 * Call sites transfer.rs:66-91 (scale_self), omr.rs:114-126. */
int orc_resize_area(const uint8_t *src, int srows, int scols, int cn, int64_t sstep,
                    uint8_t *dst, int drows, int dcols, int64_t dstep)
{
    if (srows <= 0 || scols <= 0 || drows <= 0 || dcols <= 0 || cn < 1 || cn > 4) return ORC_ERR_ASSERT;
    if (srows == drows && scols == dcols) { /* identical size: plain copy */
        for (int y = 0; y < srows; y++) memcpy(dst + (int64_t)y * dstep, src + (int64_t)y * sstep, (size_t)scols * cn);
        return ORC_OK;
    }
    double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = orc_cv_round(scale_x), iscale_y = orc_cv_round(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (!(scale_x >= 1 && scale_y >= 1)) /* either axis enlarges: the bilinear emulation (quirk B7) */
        return orc_resize_linear(src, srows, scols, cn, sstep, dst, drows, dcols, dstep, 1);

    if (is_area_fast) {
        int area = iscale_x * iscale_y;
        float scale = 1.f / (area);
        int dwidth1 = (scols / iscale_x) * cn;
        int dwidth = dcols * cn, swidth = scols * cn;
        for (int dy = 0; dy < drows; dy++) {
            uint8_t *D = dst + (int64_t)dy * dstep;
            int sy0 = dy * iscale_y;
            int w = sy0 + iscale_y <= srows ? dwidth1 : 0;
            if (sy0 >= srows) {
                for (int dx = 0; dx < dwidth; dx++) D[dx] = 0;
                continue;
            }
            int dx = 0;
            for (; dx < w; dx++) {
                int sx0 = iscale_x * (dx / cn) * cn + dx % cn; /* xofs[dx] */
                const uint8_t *S = src + (int64_t)sy0 * sstep + sx0;
                int sum = 0;
                for (int sy = 0; sy < iscale_y; sy++)
                    for (int sx = 0; sx < iscale_x; sx++) sum += S[(int64_t)sy * sstep + sx * cn];
                if (iscale_x == 2 && iscale_y == 2)
                    D[dx] = (uint8_t)((sum + 2) >> 2); /* ResizeAreaFastVec fast_mode */
                else
                    D[dx] = orc_sat_u8(orc_cv_roundf(sum * scale));
            }
            for (; dx < dwidth; dx++) {
                int sum = 0, count = 0, sx0 = iscale_x * (dx / cn) * cn + dx % cn;
                if (sx0 >= swidth) D[dx] = 0;
                for (int sy = 0; sy < iscale_y; sy++) {
                    if (sy0 + sy >= srows) break;
                    const uint8_t *S = src + (int64_t)(sy0 + sy) * sstep + sx0;
                    for (int sx = 0; sx < iscale_x * cn; sx += cn) {
                        if (sx0 + sx >= swidth) break;
                        sum += S[sx];
                        count++;
                    }
                }
                D[dx] = orc_sat_u8(orc_cv_roundf((float)sum / count));
            }
        }
        return ORC_OK;
    }

    /* general area resampling, WT = float */
    orc_decimate_alpha *xtab = (orc_decimate_alpha *)malloc(sizeof(orc_decimate_alpha) * (size_t)(scols + srows) * 2);
    if (!xtab) return ORC_ERR_NOMEM;
    orc_decimate_alpha *ytab = xtab + (size_t)scols * 2;
    int xtab_size = orc_area_tab(scols, dcols, cn, scale_x, xtab);
    int ytab_size = orc_area_tab(srows, drows, 1, scale_y, ytab);
    int dwidth = dcols * cn;
    float *buf = (float *)malloc(sizeof(float) * (size_t)dwidth * 2), *sum = buf + dwidth;
    for (int dx = 0; dx < dwidth; dx++) sum[dx] = 0.f;
    int prev_dy = ytab[0].di;
    for (int j = 0; j < ytab_size; j++) {
        float beta = ytab[j].alpha;
        int dy = ytab[j].di, sy = ytab[j].si;
        const uint8_t *S = src + (int64_t)sy * sstep;
        for (int dx = 0; dx < dwidth; dx++) buf[dx] = 0.f;
        for (int k = 0; k < xtab_size; k++) {
            float alpha = xtab[k].alpha;
            for (int c = 0; c < cn; c++) buf[xtab[k].di + c] += S[xtab[k].si + c] * alpha;
        }
        if (dy != prev_dy) {
            uint8_t *D = dst + (int64_t)prev_dy * dstep;
            for (int dx = 0; dx < dwidth; dx++) {
                D[dx] = orc_sat_u8(orc_cv_roundf(sum[dx]));
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dwidth; dx++) sum[dx] += beta * buf[dx];
        }
    }
    {
        uint8_t *D = dst + (int64_t)prev_dy * dstep;
        for (int dx = 0; dx < dwidth; dx++) D[dx] = orc_sat_u8(orc_cv_roundf(sum[dx]));
    }
    free(buf);
    free(xtab);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------ */
/* packages/lib/src/calculate.rs:2-10 */
double orc_arithmetic_mean(const double *v, size_t n)
{
    double sum = v[0];
    for (size_t i = 1; i < n; i++) sum = sum + v[i];
    return sum / (double)n;
}

/* packages/lib/src/calculate.rs:13-23.  powf(2.0) == x*x and powf(0.5) == sqrt (LLVM folds
 * pow(x,2) -> x*x and pow(x,0.5) -> sqrt for finite non-negative x); SURVEY.md A.4. */
double orc_standard_deviation(const double *v, size_t n)
{
    double mean = orc_arithmetic_mean(v, n);
    double d = v[0] - mean;
    double sum = d * d;
    for (size_t i = 1; i < n; i++) {
        d = v[i] - mean;
        sum = sum + d * d;
    }
    return sqrt(sum / (double)n);
}

/* packages/lib/src/transfer.rs:380-405 */
void orc_vertical_projection(const uint8_t *img, int rows, int cols, int64_t step, double *out)
{
    for (int x = 0; x < cols; x++) out[x] = 0.0;
    for (int y = 0; y < rows; y++) {
        const uint8_t *row = img + (int64_t)y * step;
        for (int x = 0; x < cols; x++)
            if (row[x] == 0) out[x] += 1.0;
    }
}

/* packages/lib/src/transfer.rs:305-333 */
void orc_horizontal_projection(const uint8_t *img, int rows, int cols, int64_t step, double *out)
{
    for (int y = 0; y < rows; y++) {
        const uint8_t *row = img + (int64_t)y * step;
        int sum = 0;
        for (int x = 0; x < cols; x++)
            if (row[x] == 0) sum += 1;
        out[y] = (double)sum;
    }
}

/* packages/lib/src/omr.rs:8-39 */
void orc_mat_projection_data(const uint8_t *img, int rows, int cols, int64_t step, double *h, double *v)
{
    for (int x = 0; x < cols; x++) v[x] = 0.0;
    for (int y = 0; y < rows; y++) {
        const uint8_t *row = img + (int64_t)y * step;
        int row_black_sum = 0;
        for (int x = 0; x < cols; x++)
            if (row[x] == 0) {
                row_black_sum += 1;
                v[x] += 1.0;
            }
        h[y] = (double)row_black_sum;
    }
}

/* packages/lib/src/transfer.rs:527-536: order (vertical, horizontal) */
void orc_projection_standard_deviations(const uint8_t *img, int rows, int cols, int64_t step,
                                        double *v_sd, double *h_sd)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)cols);
    double *h = (double *)malloc(sizeof(double) * (size_t)rows);
    orc_vertical_projection(img, rows, cols, step, v);
    *v_sd = orc_standard_deviation(v, (size_t)cols);
    orc_horizontal_projection(img, rows, cols, step, h);
    *h_sd = orc_standard_deviation(h, (size_t)rows);
    free(v);
    free(h);
}

/* packages/lib/src/transfer.rs:487-498: CONTAIN canvas size */
void orc_rotate_mat_size(int rows, int cols, double angle_deg, int clip, int *drows, int *dcols)
{
    if (clip == 0) {
        *drows = rows;
        *dcols = cols;
        return;
    }
    double s = fabs(sin(angle_deg * ORC_PI / 180.0)), c = fabs(cos(angle_deg * ORC_PI / 180.0));
    double rotated_width = ceil((double)rows * s + (double)cols * c);
    double rotated_height = ceil((double)cols * s + (double)rows * c);
    *dcols = (int)rotated_width;
    *drows = (int)rotated_height;
}

/* packages/lib/src/transfer.rs:459-523 (DEFAULT :472-486, CONTAIN :487-519); the CONTAIN
 * geometry is repeated in omr.rs:408-445. */
int orc_rotate_mat(const uint8_t *src, int rows, int cols, int cn, int64_t sstep,
                   double angle_deg, double scale, int interp, const uint8_t border[4], int clip,
                   uint8_t *dst, int drows, int dcols, int64_t dstep)
{
    double M[6];
    if (clip == 0) {
        if (drows != rows || dcols != cols) return ORC_ERR_ASSERT;
        float cx = (float)cols / 2.0f, cy = (float)rows / 2.0f;
        orc_get_rotation_matrix_2d(cx, cy, angle_deg, scale, M);
    } else {
        double s = fabs(sin(angle_deg * ORC_PI / 180.0)), c = fabs(cos(angle_deg * ORC_PI / 180.0));
        double rotated_width = ceil((double)rows * s + (double)cols * c);
        double rotated_height = ceil((double)cols * s + (double)rows * c);
        if (dcols != (int)rotated_width || drows != (int)rotated_height) return ORC_ERR_ASSERT;
        float cx = (float)ceil(rotated_width / 2.0), cy = (float)ceil(rotated_height / 2.0);
        orc_get_rotation_matrix_2d(cx, cy, angle_deg, scale, M);
        M[2] += ceil((rotated_width - (double)cols) / 2.0);
        M[5] += ceil((rotated_height - (double)rows) / 2.0);
    }
    if (interp == 0)
        return orc_warp_affine_nn(src, rows, cols, cn, sstep, dst, drows, dcols, dstep, M, border);
    return orc_warp_affine_linear(src, rows, cols, cn, sstep, dst, drows, dcols, dstep, M, border);
}

/* projection.rs:36-38 / omr.rs:140-145: `(max_angle as f64 / step) as u16` (Rust float->int
 * casts truncate toward zero and saturate; NaN -> 0). */
int orc_candidate_count(uint16_t max_angle, double step, int *N_out)
{
    double q = (double)max_angle / step;
    int N;
    if (!(q == q)) N = 0;
    else if (q <= 0) N = 0;
    else if (q >= 65535.0) N = 65535;
    else N = (int)q;
    if (N_out) *N_out = N;
    return 2 * N;
}

static void orc_sweep_one(const uint8_t *bin, int rows, int cols, int64_t step, const double M[6],
                          uint8_t *rot, uint8_t *clone, double *vbuf, double *hbuf,
                          uint32_t *vproj, uint32_t *hproj, double *v_sd, double *h_sd)
{
    static const uint8_t white[4] = {255, 255, 255, 0};
    /* transfer.rs:477-485 warp into a fresh Mat, :522 from_matrix -> Mat::clone */
    orc_warp_affine_nn(bin, rows, cols, 1, step, rot, rows, cols, cols, M, white);
    memcpy(clone, rot, (size_t)rows * cols);
    /* transfer.rs:527-536 */
    orc_vertical_projection(clone, rows, cols, cols, vbuf);
    *v_sd = orc_standard_deviation(vbuf, (size_t)cols);
    orc_horizontal_projection(clone, rows, cols, cols, hbuf);
    *h_sd = orc_standard_deviation(hbuf, (size_t)rows);
    if (vproj)
        for (int x = 0; x < cols; x++) vproj[x] = (uint32_t)vbuf[x];
    if (hproj)
        for (int y = 0; y < rows; y++) hproj[y] = (uint32_t)hbuf[y];
}

int orc_sweep_matrices(const uint8_t *bin, int rows, int cols, int64_t step,
                       const double *fwd_M, int A, int threads,
                       uint32_t *vproj, uint32_t *hproj, double *v_sd, double *h_sd)
{
    int rc = orc_check_dims(rows, cols, rows, cols);
    if (rc) return rc;
    if (A < 0) return ORC_ERR_BADARG;
    int nt = threads > 1 ? threads : 1;
    int fail = 0;
#pragma omp parallel num_threads(nt) if (nt > 1)
    {
        uint8_t *rot = (uint8_t *)malloc((size_t)rows * cols), *clone = (uint8_t *)malloc((size_t)rows * cols);
        double *vbuf = (double *)malloc(sizeof(double) * (size_t)cols), *hbuf = (double *)malloc(sizeof(double) * (size_t)rows);
        if (!rot || !clone || !vbuf || !hbuf) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int a = 0; a < A; a++)
                orc_sweep_one(bin, rows, cols, step, fwd_M + 6 * (size_t)a, rot, clone, vbuf, hbuf,
                              vproj ? vproj + (size_t)a * cols : NULL, hproj ? hproj + (size_t)a * rows : NULL,
                              &v_sd[a], &h_sd[a]);
        }
        free(rot);
        free(clone);
        free(vbuf);
        free(hbuf);
    }
    return fail ? ORC_ERR_NOMEM : ORC_OK;
}

/* projection.rs:47-65 (hot loop; matrix from transfer.rs:473-475). */
int orc_sweep(const uint8_t *bin, int rows, int cols, int64_t step,
              uint16_t max_angle, double angle_step, double matrix_scale, int threads,
              uint32_t *vproj, uint32_t *hproj, double *v_sd, double *h_sd)
{
    int N, A = orc_candidate_count(max_angle, angle_step, &N);
    double *Ms = (double *)malloc(sizeof(double) * 6 * (size_t)(A > 0 ? A : 1));
    if (!Ms) return ORC_ERR_NOMEM;
    float cx = (float)cols / 2.0f, cy = (float)rows / 2.0f; /* transfer.rs:474 */
    for (int i = 0; i < A; i++) {
        int deg = i - N;
        orc_get_rotation_matrix_2d(cx, cy, (double)deg * angle_step, matrix_scale, Ms + 6 * (size_t)i);
    }
    int rc = orc_sweep_matrices(bin, rows, cols, step, Ms, A, threads, vproj, hproj, v_sd, h_sd);
    free(Ms);
    return rc;
}

/* projection.rs:125-190.  The two "possibles" lists start as [0] and the scan re-visits
 * index 0, so index 0 appears twice while it holds the maximum (faithfully kept: it only
 * affects the len()==1 test at :153-156).  The final pick iterates a HashMap (random order,
 * :171-177) with strict `<` from 0.0, so any candidate whose v^2+h^2 equals the maximum over
 * the candidates can be returned; none (all zero) -> len/2 (:183-186). */
size_t orc_argmax_path1(const double *v, const double *h, size_t n, uint8_t *accept)
{
    size_t *vl = (size_t *)malloc(sizeof(size_t) * (n + 1)), *hl = (size_t *)malloc(sizeof(size_t) * (n + 1));
    size_t vn = 0, hn = 0;
    double vmax = v[0], hmax = h[0];
    vl[vn++] = 0;
    for (size_t i = 0; i < n; i++) {
        if (v[i] > vmax) {
            vmax = v[i];
            vn = 0;
            vl[vn++] = i;
        } else if (v[i] == vmax) {
            vl[vn++] = i;
        }
    }
    hl[hn++] = 0;
    for (size_t i = 0; i < n; i++) {
        if (h[i] > hmax) {
            hmax = h[i];
            hn = 0;
            hl[hn++] = i;
        } else if (h[i] == hmax) {
            hl[hn++] = i;
        }
    }
    size_t result;
    if (accept) memset(accept, 0, n);
    if (vn == 1 && hn == 1 && vl[0] == hl[0]) {
        result = vl[0];
        if (accept) accept[result] = 1;
    } else {
        uint8_t *cand = (uint8_t *)calloc(n, 1);
        for (size_t k = 0; k < vn; k++) cand[vl[k]] = 1;
        for (size_t k = 0; k < hn; k++) cand[hl[k]] = 1;
        double best = 0.0;
        int found = 0;
        for (size_t i = 0; i < n; i++)
            if (cand[i]) {
                double cur = v[i] * v[i] + h[i] * h[i];
                if (best < cur) best = cur, found = 1;
            }
        if (!found) {
            result = n / 2;
            if (accept) accept[result] = 1;
        } else {
            result = (size_t)-1;
            for (size_t i = 0; i < n; i++)
                if (cand[i] && v[i] * v[i] + h[i] * h[i] == best) {
                    if (result == (size_t)-1) result = i;
                    if (accept) accept[i] = 1;
                }
        }
        free(cand);
    }
    free(vl);
    free(hl);
    return result;
}

/* omr.rs:147-221 */
double orc_select_path2(const double *v_sd, const double *h_sd, size_t n, int N, double angle_step,
                        int *status, double *candidates, int *cand_len)
{
    double max_h = 0.0, max_v = 0.0;
    unsigned hc = 1, vc = 1;
    int len = 0;
    for (size_t i = 0; i < n; i++) {
        double ang = (double)((int)i - N) * angle_step;
        double hs = h_sd[i];
        if (max_h < hs) {
            max_h = hs;
            max_v = v_sd[i];
            hc = 1;
            vc = 1;
            len = 0;
            candidates[len++] = ang;
        } else if (max_h == hs) {
            hc += 1;
            double vs = v_sd[i];
            if (max_v < vs) {
                vc = 1;
                max_v = vs;
                len = 0;
                candidates[len++] = ang;
            } else if (max_v == vs) {
                vc += 1;
                candidates[len++] = ang;
            }
        }
    }
    *cand_len = len;
    if (len == 0) { /* omr.rs:211 would index [0] and panic (quirk B6): report NotAResult */
        *status = 2;
        return 0.0;
    }
    if (hc == 1 && vc == 1) {
        *status = 0;
        return candidates[0];
    } else if (len == 1) {
        *status = 1;
        return candidates[0];
    }
    *status = 2;
    return 0.0;
}

/* packages/lib/src/projection.rs:17-194, threads <= 1 branch. */
int orc_get_angle_with_projections(const uint8_t *src, int rows, int cols, int cn, int64_t step,
                                   uint16_t max_angle, double angle_step, double resize_scale,
                                   double *angle_out, size_t *index_out)
{
    if (rows <= 0 || cols <= 0 || (cn != 1 && cn != 3 && cn != 4)) return ORC_ERR_ASSERT;
    int rc = ORC_OK;
    /* :24-27 clone + scale_self (transfer.rs:66-91) */
    int srows = rows, scols = cols;
    uint8_t *scaled = NULL;
    const uint8_t *cur = src;
    int64_t cur_step = step;
    if (resize_scale != 1.0) {
        scols = (int)((double)cols * resize_scale);
        srows = (int)((double)rows * resize_scale);
        if (srows <= 0 || scols <= 0) return ORC_ERR_ASSERT;
        scaled = (uint8_t *)malloc((size_t)srows * scols * cn);
        if (resize_scale > 1.0) /* transfer.rs:82-86: INTER_LINEAR when enlarging, INTER_AREA otherwise */
            rc = orc_resize_linear(src, rows, cols, cn, step, scaled, srows, scols, (int64_t)scols * cn, 0);
        else
            rc = orc_resize_area(src, rows, cols, cn, step, scaled, srows, scols, (int64_t)scols * cn);
        if (rc) {
            free(scaled);
            return rc;
        }
        cur = scaled;
        cur_step = (int64_t)scols * cn;
    }
    /* :29-32 gray + threshold */
    uint8_t *gray = (uint8_t *)malloc((size_t)srows * scols), *bin = (uint8_t *)malloc((size_t)srows * scols);
    if (cn == 1)
        for (int y = 0; y < srows; y++) memcpy(gray + (size_t)y * scols, cur + (int64_t)y * cur_step, (size_t)scols);
    else
        orc_rgb2gray(cur, srows, scols, cn, cur_step, gray, scols);
    orc_threshold_binary(gray, srows, scols, scols, bin, scols, 127, 255);
    /* :36-65 */
    int N, A = orc_candidate_count(max_angle, angle_step, &N);
    if (A <= 0) {
        free(scaled);
        free(gray);
        free(bin);
        return ORC_ERR_BADARG; /* the reference indexes vertical_vec[0] and panics */
    }
    double *vs = (double *)malloc(sizeof(double) * (size_t)A), *hs = (double *)malloc(sizeof(double) * (size_t)A);
    rc = orc_sweep(bin, srows, scols, scols, max_angle, angle_step, 1.0, 1, NULL, NULL, vs, hs);
    if (!rc) {
        size_t idx = orc_argmax_path1(vs, hs, (size_t)A, NULL);
        if (index_out) *index_out = idx;
        *angle_out = ((double)idx - (double)N) * angle_step; /* :189-190 */
    }
    free(vs);
    free(hs);
    free(scaled);
    free(gray);
    free(bin);
    return rc;
}

/* packages/lib/src/omr.rs:52-229 */
int orc_get_result_from_projection(const uint8_t *src, int rows, int cols, int cn, int64_t step,
                                   uint16_t max_angle, double angle_step, int max_w, int max_h,
                                   double *angle, int *status, double *candidates, int cand_cap,
                                   int *cand_len)
{
    if (rows <= 0 || cols <= 0 || (cn != 1 && cn != 3 && cn != 4)) return ORC_ERR_ASSERT;
    /* :60-82 */
    double width_scale = max_w <= 0 ? 1.0 : (double)max_w / (double)cols;
    double height_scale = max_h <= 0 ? 1.0 : (double)max_h / (double)rows;
    double scale = width_scale < height_scale ? width_scale : height_scale;
    /* :88-92 gray, :98-112 erode */
    uint8_t *gray = (uint8_t *)malloc((size_t)rows * cols), *er = (uint8_t *)malloc((size_t)rows * cols);
    if (cn == 1)
        for (int y = 0; y < rows; y++) memcpy(gray + (size_t)y * cols, src + (int64_t)y * step, (size_t)cols);
    else
        orc_rgb2gray(src, rows, cols, cn, step, gray, cols);
    orc_erode_cross3(gray, rows, cols, cols, er, cols, 3);
    /* :114-126 resize to ((w*scale) as i32, (h*scale) as i32) */
    int scols = (int)((double)cols * scale), srows = (int)((double)rows * scale);
    if (scols <= 0 || srows <= 0) {
        free(gray);
        free(er);
        return ORC_ERR_ASSERT;
    }
    uint8_t *scaled = (uint8_t *)malloc((size_t)srows * scols), *bin = (uint8_t *)malloc((size_t)srows * scols);
    int rc = orc_resize_area(er, rows, cols, 1, cols, scaled, srows, scols, scols);
    if (!rc) {
        orc_threshold_binary(scaled, srows, scols, scols, bin, scols, 127, 255); /* :129-139 */
        int N, A = orc_candidate_count(max_angle, angle_step, &N);
        double *vs = (double *)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1)), *hs = (double *)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1));
        double *cand = (double *)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1));
        /* :153-208: rotation matrix scale = projection_resize_scale (quirk B4) */
        rc = orc_sweep(bin, srows, scols, scols, max_angle, angle_step, scale, 1, NULL, NULL, vs, hs);
        if (!rc) {
            int len = 0;
            *angle = orc_select_path2(vs, hs, (size_t)A, N, angle_step, status, cand, &len);
            *cand_len = len;
            for (int i = 0; i < len && i < cand_cap; i++) candidates[i] = cand[i];
        }
        free(vs);
        free(hs);
        free(cand);
    }
    free(gray);
    free(er);
    free(scaled);
    free(bin);
    return rc;
}
