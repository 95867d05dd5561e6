// stages.hip -- tuned single-channel stage kernels either side of the sweep:
// the front end of omr.rs:87-139 (gray, erode x3, INTER_AREA shrink) and the final deskew warp of
// transfer.rs:459-523 / omr.rs:408-445.  All are HBM-bound byte work: the fast forms move 4 pixels
// per lane (dword loads / stores), stage reuse through LDS and keep OpenCV 4.6.0's integer
// arithmetic bit for bit (kernels.hip holds the generic any-channel-count forms).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace omr {

__device__ __forceinline__ uint8_t st_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// ------------------------------------------------------------------------------------------
// cvtColor(COLOR_RGB2GRAY), 3 channels, 4 pixels per lane: three dword loads, one dword store.
__global__ __launch_bounds__(256) void rgb2gray3_x4_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                           int cols, uint8_t *__restrict__ dst, int64_t dstep)
{
    const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int x = q * 4;
    if (x >= cols) return;
    const uint8_t *S = src + (int64_t)y * sstep + (int64_t)x * 3;
    uint8_t *D = dst + (int64_t)y * dstep + x;
    if (x + 4 <= cols) {
        const uint32_t a = *(const uint32_t *)(S), b = *(const uint32_t *)(S + 4), c = *(const uint32_t *)(S + 8);
        // bytes: a = p0.c0 p0.c1 p0.c2 p1.c0 | b = p1.c1 p1.c2 p2.c0 p2.c1 | c = p2.c2 p3.c0 p3.c1 p3.c2
        const uint32_t g0 = ((a & 255) * 9798 + ((a >> 8) & 255) * 19235 + ((a >> 16) & 255) * 3735 + (1 << 14)) >> 15;
        const uint32_t g1 = ((a >> 24) * 9798 + (b & 255) * 19235 + ((b >> 8) & 255) * 3735 + (1 << 14)) >> 15;
        const uint32_t g2 = (((b >> 16) & 255) * 9798 + (b >> 24) * 19235 + (c & 255) * 3735 + (1 << 14)) >> 15;
        const uint32_t g3 = (((c >> 8) & 255) * 9798 + ((c >> 16) & 255) * 19235 + (c >> 24) * 3735 + (1 << 14)) >> 15;
        *(uint32_t *)D = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (int j = 0; x + j < cols; j++)
            D[j] = (uint8_t)((S[3 * j] * 9798 + S[3 * j + 1] * 19235 + S[3 * j + 2] * 3735 + (1 << 14)) >> 15);
    }
}

hipError_t launch_rgb2gray_fast(const uint8_t *d_src, int64_t sstep, int rows, int cols, int cn, uint8_t *d_dst,
                                int64_t dstep, hipStream_t s)
{
    if (cn == 3 && (sstep & 3) == 0 && (dstep & 3) == 0 && ((uintptr_t)d_src & 3) == 0 && ((uintptr_t)d_dst & 3) == 0) {
        hipLaunchKernelGGL(rgb2gray3_x4_kernel, dim3((cols + 1023) / 1024, rows), dim3(256), 0, s, d_src, sstep, rows,
                           cols, d_dst, dstep);
        return hipGetLastError();
    }
    return launch_rgb2gray(d_src, sstep, rows, cols, cn, d_dst, dstep, s);
}

// ------------------------------------------------------------------------------------------
// erode(3x3 cross, iterations = 3, border = +inf), the three passes fused in LDS: a 64 x 16 output
// tile with a halo of 3; positions outside the image stay +inf (255) in every pass, exactly as
// three separate cv::erode calls would treat them.
#define ER_TW 64
#define ER_TH 16
#define ER_W (ER_TW + 6)
#define ER_H (ER_TH + 6)

__global__ __launch_bounds__(256) void erode3x_cross_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                            int cols, uint8_t *__restrict__ dst, int64_t dstep)
{
    __shared__ uint8_t t0[ER_H][ER_W + 2], t1[ER_H][ER_W + 2];
    const int x0 = blockIdx.x * ER_TW - 3, y0 = blockIdx.y * ER_TH - 3;
    for (int i = threadIdx.x; i < ER_W * ER_H; i += 256) {
        const int ly = i / ER_W, lx = i - ly * ER_W;
        const int gx = x0 + lx, gy = y0 + ly;
        const bool in = (unsigned)gx < (unsigned)cols && (unsigned)gy < (unsigned)rows;
        t0[ly][lx] = in ? src[(int64_t)gy * sstep + gx] : 255;
    }
    __syncthreads();
    // pass p shrinks the valid region by one on every side
    for (int pass = 1; pass <= 3; pass++) {
        uint8_t(*a)[ER_W + 2] = (pass & 1) ? t0 : t1;
        uint8_t(*b)[ER_W + 2] = (pass & 1) ? t1 : t0;
        const int w = ER_W - 2 * pass, h = ER_H - 2 * pass;
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int ly = pass + i / w, lx = pass + i % w;
            const int gx = x0 + lx, gy = y0 + ly;
            int m = 255;
            if ((unsigned)gx < (unsigned)cols && (unsigned)gy < (unsigned)rows)
                m = min(min((int)a[ly][lx], min((int)a[ly - 1][lx], (int)a[ly + 1][lx])),
                        min((int)a[ly][lx - 1], (int)a[ly][lx + 1]));
            b[ly][lx] = (uint8_t)m;
        }
        __syncthreads();
    }
    // after three passes the result sits in t1 (passes 1 and 3 write t1)
    for (int i = threadIdx.x; i < ER_TW * ER_TH; i += 256) {
        const int ly = i / ER_TW, lx = i - ly * ER_TW;
        const int gx = x0 + 3 + lx, gy = y0 + 3 + ly;
        if (gx < cols && gy < rows) dst[(int64_t)gy * dstep + gx] = t1[ly + 3][lx + 3];
    }
}

hipError_t launch_erode3x_cross(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                                hipStream_t s)
{
    hipLaunchKernelGGL(erode3x_cross_kernel, dim3((cols + ER_TW - 1) / ER_TW, (rows + ER_TH - 1) / ER_TH), dim3(256), 0,
                       s, d_src, sstep, rows, cols, d_dst, dstep);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// resize INTER_AREA, integer factor k (OpenCV resizeAreaFast_), 1 channel, full blocks only
// (k divides both sizes): coalesced row loads, k*k sum per output pixel from LDS.
#define RA_OW 64
#define RA_OH 4
#define RA_MAXK 8

__global__ __launch_bounds__(256) void resize_area_int_c1_kernel(const uint8_t *__restrict__ src, int64_t sstep,
                                                                 uint8_t *__restrict__ dst, int64_t dstep, int drows,
                                                                 int dcols, int k)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[RA_OH * RA_MAXK][RA_OW * RA_MAXK + 4];
    const int ox0 = blockIdx.x * RA_OW, oy0 = blockIdx.y * RA_OH;
    const int iw = min(RA_OW, dcols - ox0) * k, ih = min(RA_OH, drows - oy0) * k;
    const uint8_t *S = src + (int64_t)oy0 * k * sstep + (int64_t)ox0 * k;
    if ((((uintptr_t)S | (uintptr_t)sstep) & 3) == 0) {  // dword loads: a quarter of the load instructions
        const int iw4 = iw >> 2;
        for (int i = threadIdx.x; i < iw4 * ih; i += 256) {
            const int ly = i / iw4, lq = i - ly * iw4;
            *(uint32_t *)&tile[ly][lq * 4] = *(const uint32_t *)(S + (int64_t)ly * sstep + lq * 4);
        }
        for (int i = threadIdx.x; i < (iw & 3) * ih; i += 256) {  // last, partial block of a row
            const int ly = i / (iw & 3), lx = (iw & ~3) + i % (iw & 3);
            tile[ly][lx] = S[(int64_t)ly * sstep + lx];
        }
    } else {
        for (int i = threadIdx.x; i < iw * ih; i += 256) {
            const int ly = i / iw, lx = i - ly * iw;
            tile[ly][lx] = S[(int64_t)ly * sstep + lx];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & (RA_OW - 1), ly = threadIdx.x / RA_OW;
    const int ox = ox0 + lx, oy = oy0 + ly;
    if (ox < dcols && oy < drows) {
        int sum = 0;
        for (int yy = 0; yy < k; yy++)
            for (int xx = 0; xx < k; xx++) sum += tile[ly * k + yy][lx * k + xx];
        uint8_t out;
        if (k == 2) out = (uint8_t)((sum + 2) >> 2);
        else out = st_sat_u8((int)rintf((float)sum * (1.f / (float)(k * k))));
        dst[(int64_t)oy * dstep + ox] = out;
    }
}

hipError_t launch_resize_area_int_fast(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn,
                                       uint8_t *d_dst, int64_t dstep, int drows, int dcols, int kx, int ky,
                                       hipStream_t s)
{
    if (cn == 1 && kx == ky && kx >= 2 && kx <= RA_MAXK && drows * ky == srows && dcols * kx == scols) {
        hipLaunchKernelGGL(resize_area_int_c1_kernel, dim3((dcols + RA_OW - 1) / RA_OW, (drows + RA_OH - 1) / RA_OH),
                           dim3(256), 0, s, d_src, sstep, d_dst, dstep, drows, dcols, kx);
        return hipGetLastError();
    }
    return launch_resize_area_int(d_src, sstep, srows, scols, cn, d_dst, dstep, drows, dcols, kx, ky, s);
}

// ------------------------------------------------------------------------------------------
// warpAffine on a 1-channel image, 4 destination pixels per lane (one dword store).  The fixed-
// point tables are evaluated in place with the same f64 expressions (file built -ffp-contract=off).
struct WarpM {
    double m[6];
};

template <bool LINEAR>
__global__ __launch_bounds__(256) void warp_c1_x4_kernel(const uint8_t *__restrict__ src, int64_t sstep, int srows,
                                                         int scols, uint8_t *__restrict__ dst, int64_t dstep, int drows,
                                                         int dcols, const WarpM W, int border)
{
    const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int x0 = q * 4;
    if (x0 >= dcols) return;
    const double *M = W.m;
    const int rd = LINEAR ? 16 : 512;
    const int X0 = (int)rint((M[1] * (double)y + M[2]) * 1024.0) + rd;
    const int Y0 = (int)rint((M[4] * (double)y + M[5]) * 1024.0) + rd;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        const int adelta = (int)rint(M[0] * (double)x * 1024.0);
        const int bdelta = (int)rint(M[3] * (double)x * 1024.0);
        int v;
        if (!LINEAR) {
            int X = (X0 + adelta) >> 10, Y = (Y0 + bdelta) >> 10;
            X = max(-32768, min(32767, X));
            Y = max(-32768, min(32767, Y));
            v = ((unsigned)X < (unsigned)scols && (unsigned)Y < (unsigned)srows) ? src[(int64_t)Y * sstep + X] : border;
        } else {
            const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
            const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
            const int fx = X & 31, fy = Y & 31;
            if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) {
                v = border;
            } else {
                const bool in_x0 = sx >= 0 && sx < scols, in_x1 = sx + 1 >= 0 && sx + 1 < scols;
                const bool in_y0 = sy >= 0 && sy < srows, in_y1 = sy + 1 >= 0 && sy + 1 < srows;
                const int v0 = in_x0 && in_y0 ? src[(int64_t)sy * sstep + sx] : border;
                const int v1 = in_x1 && in_y0 ? src[(int64_t)sy * sstep + sx + 1] : border;
                const int v2 = in_x0 && in_y1 ? src[(int64_t)(sy + 1) * sstep + sx] : border;
                const int v3 = in_x1 && in_y1 ? src[(int64_t)(sy + 1) * sstep + sx + 1] : border;
                const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32,
                          w3 = fy * fx * 32;
                v = st_sat_u8((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
            }
        }
        out |= (uint32_t)v << (8 * j);
    }
    uint8_t *D = dst + (int64_t)y * dstep + x0;
    if (x0 + 4 <= dcols) {
        *(uint32_t *)D = out;
    } else {
        for (int j = 0; x0 + j < dcols; j++) D[j] = (uint8_t)(out >> (8 * j));
    }
}

hipError_t launch_warp_c1_fast(const uint8_t *d_src, int64_t sstep, int srows, int scols, uint8_t *d_dst, int64_t dstep,
                               int drows, int dcols, const double Minv[6], int interp, int border, hipStream_t s)
{
    WarpM W;
    for (int i = 0; i < 6; i++) W.m[i] = Minv[i];
    if ((dstep & 3) != 0 || ((uintptr_t)d_dst & 3) != 0) return hipErrorInvalidValue;  // caller falls back
    dim3 grid((dcols + 1023) / 1024, drows);
    if (interp == 0)
        hipLaunchKernelGGL((warp_c1_x4_kernel<false>), grid, dim3(256), 0, s, d_src, sstep, srows, scols, d_dst, dstep,
                           drows, dcols, W, border);
    else
        hipLaunchKernelGGL((warp_c1_x4_kernel<true>), grid, dim3(256), 0, s, d_src, sstep, srows, scols, d_dst, dstep,
                           drows, dcols, W, border);
    return hipGetLastError();
}

// threshold(thresh, maxval, BINARY), 16 pixels per lane
__global__ __launch_bounds__(256) void threshold_x16_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                            int cols, uint8_t *__restrict__ dst, int64_t dstep,
                                                            int thresh, int maxval)
{
    const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int x = q * 16;
    if (x >= cols) return;
    const uint8_t *S = src + (int64_t)y * sstep + x;
    uint8_t *D = dst + (int64_t)y * dstep + x;
    if (x + 16 <= cols) {
        const uint4 v = *(const uint4 *)S;
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t r = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) r |= ((int)((in[k] >> (8 * j)) & 255) > thresh ? (uint32_t)maxval : 0u) << (8 * j);
            o[k] = r;
        }
        *(uint4 *)D = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        for (int j = 0; x + j < cols; j++) D[j] = (int)S[j] > thresh ? (uint8_t)maxval : 0;
    }
}

hipError_t launch_threshold_fast(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                                 int thresh, int maxval, hipStream_t s)
{
    if ((sstep & 15) == 0 && (dstep & 15) == 0 && ((uintptr_t)d_src & 15) == 0 && ((uintptr_t)d_dst & 15) == 0) {
        hipLaunchKernelGGL(threshold_x16_kernel, dim3((cols + 4095) / 4096, rows), dim3(256), 0, s, d_src, sstep, rows,
                           cols, d_dst, dstep, thresh, maxval);
        return hipGetLastError();
    }
    return launch_threshold(d_src, sstep, rows, cols, d_dst, dstep, thresh, maxval, s);
}

// resize(INTER_LINEAR) and INTER_AREA's bilinear emulation when an axis enlarges (OpenCV resizeGeneric_ with
// HResizeLinear<uchar,int,short,2048> / VResizeLinear<uchar,int,short,FixedPtCast<22>>): scale_self with
// scale > 1 (transfer.rs:66-91) and path 2's unclamped scale (omr.rs:60-82,114-126, quirk B7).  Every thread
// rebuilds its two coefficient pairs with the expressions of resize.cpp (double products, float fractions,
// saturate_cast<short>(c * 2048) with round-half-even; the file is built -ffp-contract=off): 11-bit
// horizontal taps on the two source rows, then (((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2) >> 2.
// sx is monotone in dx, so "dx >= xmax" (the columns that copy S[sx] * 2048) is just sx + 1 >= scols.
__device__ __forceinline__ void linear_coef(int d, double scale, double inv_scale, int ssize, bool area_mode, int &s0,
                                            int &c0, int &c1, bool &edge)
{
    float f;
    int sx;
    if (!area_mode) {
        f = (float)(((double)d + 0.5) * scale - 0.5);
        sx = (int)floorf(f);
        f -= (float)sx;
    } else {
        sx = (int)floor((double)d * scale);
        f = (float)((double)(d + 1) - (double)(sx + 1) * inv_scale);
        f = f <= 0.f ? 0.f : f - floorf(f);
    }
    s0 = sx;
    edge = false;
    if (ssize > 0) {  // horizontal axis only: the vertical axis keeps sy and clips the ROWS instead
        if (sx < 0) f = 0.f, sx = 0;
        if (sx + 1 >= ssize) {
            edge = true;
            if (sx >= ssize - 1) f = 0.f, sx = ssize - 1;
        }
        s0 = sx;
    }
    c0 = max(-32768, min(32767, (int)rintf((1.f - f) * 2048.f)));
    c1 = max(-32768, min(32767, (int)rintf(f * 2048.f)));
}

__global__ __launch_bounds__(256) void resize_linear_kernel(const uint8_t *__restrict__ src, int64_t sstep, int srows,
                                                            int scols, int cn, uint8_t *__restrict__ dst, int64_t dstep,
                                                            int drows, int dcols, double scale_x, double inv_scale_x,
                                                            double scale_y, double inv_scale_y, int area_mode)
{
    const int dxb = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dxb >= dcols * cn) return;
    const int dx = dxb / cn, c = dxb - dx * cn;
    int sx, a0, a1, sy, b0, b1;
    bool edge, unused;
    linear_coef(dx, scale_x, inv_scale_x, scols, area_mode != 0, sx, a0, a1, edge);
    linear_coef(dy, scale_y, inv_scale_y, 0, area_mode != 0, sy, b0, b1, unused);
    const int sy0 = max(0, min(srows - 1, sy)), sy1 = max(0, min(srows - 1, sy + 1));
    const uint8_t *S0 = src + (int64_t)sy0 * sstep + (int64_t)sx * cn + c;
    const uint8_t *S1 = src + (int64_t)sy1 * sstep + (int64_t)sx * cn + c;
    int h0, h1;
    if (!edge) {
        h0 = (int)S0[0] * a0 + (int)S0[cn] * a1;
        h1 = (int)S1[0] * a0 + (int)S1[cn] * a1;
    } else {
        h0 = (int)S0[0] * 2048;
        h1 = (int)S1[0] * 2048;
    }
    dst[(int64_t)dy * dstep + dxb] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
}

hipError_t launch_resize_linear(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                                int64_t dstep, int drows, int dcols, bool area_mode, hipStream_t s)
{
    const double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((dcols * cn + 255) / 256, drows), dim3(256), 0, s, d_src, sstep, srows,
                       scols, cn, d_dst, dstep, drows, dcols, scale_x, inv_scale_x, scale_y, inv_scale_y,
                       area_mode ? 1 : 0);
    return hipGetLastError();
}

}  // namespace omr
