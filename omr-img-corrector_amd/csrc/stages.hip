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

// The same filter at 4 pixels per lane.  Three passes of the 5-point minimum with a +inf border are one
// minimum over the L1 ball of radius 3 (in-image points only):
//     out(y, x) = min over dy in [-3, 3] of  h_{3 - |dy|}(y + dy, x),   h_r(y, x) = min over |dx| <= r of in(y, x + dx).
// Tile = 256 x 64 output pixels; its source rows (+3 halo rows, +1 halo dword either side, 255 outside the
// image) go to LDS with every load in flight at once.  A lane then owns one dword column (4 pixels) of a
// 16-row strip and walks down its 22 source rows: the eight byte pairs (p_d, p_{d+2}), d = -3..4, come out of
// the three dwords around the column with v_perm_b32 as zero-extended u16 pairs, so every minimum handles
// two pixels (v_pk_min_u16): 12 for h_1..h_3 of the even and the odd pixels, 12 for the seven running
// column minima that a source row feeds (row t closes output row t - 3).
#define E4_TW 64  // dword columns per tile
#define E4_SH 16  // output rows per strip
#define E4_NS 4   // strips per tile
#define E4_LW (E4_TW + 2)
#define E4_LH (E4_SH * E4_NS + 6)

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(256) void erode3x_cross_x4_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                               int cols, uint8_t *__restrict__ dst, int64_t dstep)
{
    __shared__ uint32_t tile[E4_LH][E4_LW + 1];
    const int qx0 = blockIdx.x * E4_TW - 1;          // first dword column of the tile (halo)
    const int y0 = blockIdx.y * (E4_SH * E4_NS) - 3;  // first source row of the tile (halo)
    for (int i = threadIdx.x; i < E4_LH * E4_LW; i += 256) {
        const int ly = i / E4_LW, lq = i - ly * E4_LW;
        const int gy = y0 + ly, gx = (qx0 + lq) * 4;
        uint32_t v = 0xffffffffu;
        if ((unsigned)gy < (unsigned)rows && gx >= 0 && gx < cols) {
            const uint8_t *S = src + (int64_t)gy * sstep + gx;
            if (gx + 4 <= cols) {
                v = *(const uint32_t *)S;
            } else {  // the row ends inside this dword: pixels past the end stay +inf
                for (int j = 0; gx + j < cols; j++) v = (v & ~(255u << (8 * j))) | ((uint32_t)S[j] << (8 * j));
            }
        }
        tile[ly][lq] = v;
    }
    __syncthreads();
    const int lq = threadIdx.x & (E4_TW - 1), strip = threadIdx.x / E4_TW;
    const int gx = (qx0 + 1 + lq) * 4;
    const int oy0 = blockIdx.y * (E4_SH * E4_NS) + strip * E4_SH;
    if (gx >= cols || oy0 >= rows) return;
    uint32_t aE[6], aO[6];  // running minima of output rows t - 2 .. t + 3 (t = source row being read)
#pragma unroll
    for (int i = 0; i < 6; i++) aE[i] = aO[i] = 0x00ff00ffu;
#pragma unroll
    for (int t = 0; t < E4_SH + 6; t++) {  // source row oy0 - 3 + t
        const uint32_t *T = &tile[strip * E4_SH + t][lq];
        const uint32_t L = T[0], C = T[1], R = T[2];
        const uint32_t Pm3 = __builtin_amdgcn_perm(L, L, 0x0c030c01u), Pm2 = __builtin_amdgcn_perm(C, L, 0x0c040c02u);
        const uint32_t Pm1 = __builtin_amdgcn_perm(C, L, 0x0c050c03u), P0 = C & 0x00ff00ffu;
        const uint32_t P1 = __builtin_amdgcn_perm(C, C, 0x0c030c01u), P2 = __builtin_amdgcn_perm(R, C, 0x0c040c02u);
        const uint32_t P3 = __builtin_amdgcn_perm(R, C, 0x0c050c03u), P4 = R & 0x00ff00ffu;
        const uint32_t h1E = pk_min(pk_min(Pm1, P0), P1), h1O = pk_min(pk_min(P0, P1), P2);
        const uint32_t h2E = pk_min(pk_min(h1E, Pm2), P2), h2O = pk_min(pk_min(h1O, Pm1), P3);
        const uint32_t h3E = pk_min(pk_min(h2E, Pm3), P3), h3O = pk_min(pk_min(h2O, Pm2), P4);
        // source row t feeds output rows t-3 (h0, closes it), t-2 (h1), t-1 (h2), t (h3), t+1 (h2), t+2 (h1), t+3 (h0)
        const uint32_t oE = pk_min(aE[0], P0), oO = pk_min(aO[0], P1);
        aE[0] = pk_min(aE[1], h1E), aO[0] = pk_min(aO[1], h1O);
        aE[1] = pk_min(aE[2], h2E), aO[1] = pk_min(aO[2], h2O);
        aE[2] = pk_min(aE[3], h3E), aO[2] = pk_min(aO[3], h3O);
        aE[3] = pk_min(aE[4], h2E), aO[3] = pk_min(aO[4], h2O);
        aE[4] = pk_min(aE[5], h1E), aO[4] = pk_min(aO[5], h1O);
        aE[5] = P0, aO[5] = P1;
        if (t >= 6) {
            const int oy = oy0 + t - 6;
            if (oy < rows) {
                const uint32_t out = oE | (oO << 8);
                uint8_t *D = dst + (int64_t)oy * dstep + gx;
                if (gx + 4 <= cols && (((uintptr_t)D) & 3) == 0) {
                    *(uint32_t *)D = out;
                } else {
                    for (int j = 0; gx + j < cols && j < 4; j++) D[j] = (uint8_t)(out >> (8 * j));
                }
            }
        }
    }
}

hipError_t launch_erode3x_cross(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                                hipStream_t s)
{
    if ((sstep & 3) == 0 && ((uintptr_t)d_src & 3) == 0) {
        hipLaunchKernelGGL(erode3x_cross_x4_kernel,
                           dim3((cols + 4 * E4_TW - 1) / (4 * E4_TW), (rows + E4_SH * E4_NS - 1) / (E4_SH * E4_NS)), dim3(256),
                           0, s, d_src, sstep, rows, cols, d_dst, dstep);
        return hipGetLastError();
    }
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

// The same reduction with the column sums taken first: a lane adds the k source rows of one dword column
// straight from global memory (k coalesced dword loads in flight, 4 pixels each, byte pairs widened to
// u16 with two masks and added with v_pk_add_u16), parks four u16 column sums in LDS, and after the
// barrier a lane per output pixel adds k neighbouring column sums -- 2k LDS accesses per output pixel
// instead of k*k byte reads, and no staging of the raw tile.
#define RB_OW 64
#define RB_OH 4
#define RB_MAXK 16

__global__ __launch_bounds__(256) void resize_area_int_colsum_kernel(const uint8_t *__restrict__ src, int64_t sstep,
                                                                     uint8_t *__restrict__ dst, int64_t dstep, int drows,
                                                                     int dcols, int k)
{
    __shared__ uint16_t colsum[RB_OH][RB_OW * RB_MAXK + 8];
    const int ox0 = blockIdx.x * RB_OW, oy0 = blockIdx.y * RB_OH;
    const int ow = min(RB_OW, dcols - ox0), oh = min(RB_OH, drows - oy0);
    const int iw = ow * k, nq = (iw + 3) >> 2;  // source pixels / dword columns of the tile (iw may end inside a dword)
    const uint8_t *S = src + (int64_t)oy0 * k * sstep + (int64_t)ox0 * k;  // 4-byte aligned: ox0 * k is a multiple of 64
    for (int i = threadIdx.x; i < nq * oh; i += 256) {
        const int ly = i / nq, q = i - ly * nq;
        const uint8_t *P = S + (int64_t)ly * k * sstep + q * 4;
        uint32_t e = 0, o = 0;  // (px0, px2) and (px1, px3) as u16 pairs
        for (int yy = 0; yy < k; yy++) {
            const uint32_t v = *(const uint32_t *)(P + (int64_t)yy * sstep);  // may read past iw inside the row pitch: unused
            e += v & 0x00ff00ffu;
            o += (v >> 8) & 0x00ff00ffu;
        }
        uint16_t *C = &colsum[ly][q * 4];
        C[0] = (uint16_t)e;
        C[1] = (uint16_t)o;
        C[2] = (uint16_t)(e >> 16);
        C[3] = (uint16_t)(o >> 16);
    }
    __syncthreads();
    const int lx = threadIdx.x & (RB_OW - 1), ly = threadIdx.x / RB_OW;
    if (lx < ow && ly < oh) {
        int sum = 0;
        for (int xx = 0; xx < k; xx++) sum += colsum[ly][lx * k + xx];
        uint8_t out;
        if (k == 2) out = (uint8_t)((sum + 2) >> 2);
        else out = st_sat_u8((int)rintf((float)sum * (1.f / (float)(k * k))));
        dst[(int64_t)(oy0 + ly) * dstep + ox0 + lx] = out;
    }
}

hipError_t launch_resize_area_int_fast(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn,
                                       uint8_t *d_dst, int64_t dstep, int drows, int dcols, int kx, int ky,
                                       hipStream_t s)
{
    const bool full = cn == 1 && kx == ky && kx >= 2 && drows * ky == srows && dcols * kx == scols;
    // the dword loads of the last column may run up to 3 bytes past the last source pixel of a row: the row pitch
    // must cover them (always true for a pitch that is a multiple of 4)
    if (full && kx <= RB_MAXK && (sstep & 3) == 0 && ((uintptr_t)d_src & 3) == 0) {
        hipLaunchKernelGGL(resize_area_int_colsum_kernel, dim3((dcols + RB_OW - 1) / RB_OW, (drows + RB_OH - 1) / RB_OH),
                           dim3(256), 0, s, d_src, sstep, d_dst, dstep, drows, dcols, kx);
        return hipGetLastError();
    }
    if (full && kx <= RA_MAXK) {
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

// The same warp with the source staged in LDS.  A workgroup owns a 64 x 16 destination tile; the affine map
// takes it to a parallelogram whose bounding box (found from the four corner samples, one pixel of slack for
// the rounding of the fixed-point tables, one more for the bilinear taps) is copied to LDS with row-contiguous
// dword loads, border value outside the image -- so a tap is one LDS byte read with no bounds test, instead of
// a byte gather through L1 along a slanted line (about 45 cache lines per wave).  A tile whose box does not
// fit (strong magnification) takes its taps from global memory as before.
#define WL_TW 64
#define WL_TH 16
#define WL_LDS 16384

template <bool LINEAR>
__device__ __forceinline__ int warp_tap_global(const uint8_t *__restrict__ src, int64_t sstep, int srows, int scols,
                                               int cn, int Xf, int Yf, int border)
{
    if (!LINEAR) {
        const int X = max(-32768, min(32767, Xf >> 10)), Y = max(-32768, min(32767, Yf >> 10));
        return ((unsigned)X < (unsigned)scols && (unsigned)Y < (unsigned)srows) ? src[(int64_t)Y * sstep + (int64_t)X * cn] : border;
    }
    const int X = Xf >> 5, Y = Yf >> 5;
    const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
    const int fx = X & 31, fy = Y & 31;
    if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) return border;
    const bool in_x0 = sx >= 0 && sx < scols, in_x1 = sx + 1 >= 0 && sx + 1 < scols;
    const bool in_y0 = sy >= 0 && sy < srows, in_y1 = sy + 1 >= 0 && sy + 1 < srows;
    const int v0 = in_x0 && in_y0 ? src[(int64_t)sy * sstep + (int64_t)sx * cn] : border;
    const int v1 = in_x1 && in_y0 ? src[(int64_t)sy * sstep + (int64_t)(sx + 1) * cn] : border;
    const int v2 = in_x0 && in_y1 ? src[(int64_t)(sy + 1) * sstep + (int64_t)sx * cn] : border;
    const int v3 = in_x1 && in_y1 ? src[(int64_t)(sy + 1) * sstep + (int64_t)(sx + 1) * cn] : border;
    const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    return st_sat_u8((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
}

template <int CN, bool LINEAR>
__global__ __launch_bounds__(256) void warp_lds_kernel(const uint8_t *__restrict__ src, int64_t sstep, int srows,
                                                       int scols, uint8_t *__restrict__ dst, int64_t dstep, int drows,
                                                       int dcols, const WarpM W, uint32_t border_rgba)
{
    __shared__ __attribute__((aligned(16))) uint8_t box[WL_LDS];
    const double *M = W.m;
    const int rd = LINEAR ? 16 : 512;
    const int tx0 = blockIdx.x * WL_TW, ty0 = blockIdx.y * WL_TH;
    const int tx1 = min(dcols, tx0 + WL_TW) - 1, ty1 = min(drows, ty0 + WL_TH) - 1;
    // fixed-point source coordinates (OpenCV's tables) of the tile's corner samples: wave-uniform.  X0(y) and
    // adelta(x) are both monotone, so the four corners bound every sample of the tile.
    auto FX = [&](int x, int y) { return (int)rint((M[1] * (double)y + M[2]) * 1024.0) + rd + (int)rint(M[0] * (double)x * 1024.0); };
    auto FY = [&](int x, int y) { return (int)rint((M[4] * (double)y + M[5]) * 1024.0) + rd + (int)rint(M[3] * (double)x * 1024.0); };
    const int cx[4] = {FX(tx0, ty0) >> 10, FX(tx1, ty0) >> 10, FX(tx0, ty1) >> 10, FX(tx1, ty1) >> 10};
    const int cy[4] = {FY(tx0, ty0) >> 10, FY(tx1, ty0) >> 10, FY(tx0, ty1) >> 10, FY(tx1, ty1) >> 10};
    const int bx0 = min(min(cx[0], cx[1]), min(cx[2], cx[3])) - 1;
    const int bx1 = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 1 + (LINEAR ? 1 : 0);
    const int by0 = min(min(cy[0], cy[1]), min(cy[2], cy[3])) - 1;
    const int by1 = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 1 + (LINEAR ? 1 : 0);
    // the box in BYTES of a source row, widened to whole dwords of the row
    const int bb0 = (bx0 * CN) & ~3, bb1 = ((bx1 + 1) * CN + 3) & ~3;  // [bb0, bb1)
    const int bwb = bb1 - bb0, bh = by1 - by0 + 1;
    const bool staged = bwb > 0 && bh > 0 && (int64_t)bwb * bh <= WL_LDS && bx0 > -30000 && bx1 < 30000 &&
                        by0 > -30000 && by1 < 30000;
    const int rowb = scols * CN;  // bytes of a source row that hold pixels
    if (staged) {
        const int bq = bwb >> 2;
        const bool aligned = ((sstep | (int64_t)(uintptr_t)src) & 3) == 0;
        for (int i = threadIdx.x; i < bq * bh; i += 256) {
            const int ly = i / bq, lq = i - ly * bq;
            const int gy = by0 + ly, gb = bb0 + lq * 4;
            uint32_t v;
            if (aligned && (unsigned)gy < (unsigned)srows && gb >= 0 && gb + 4 <= rowb) {
                v = *(const uint32_t *)(src + (int64_t)gy * sstep + gb);
            } else {
                v = 0;
                for (int j = 0; j < 4; j++) {
                    const int b = gb + j;  // byte b of row gy: pixel b / CN, channel b % CN (floor semantics for b < 0)
                    const int ch = ((b % CN) + CN) % CN;
                    uint32_t px = (border_rgba >> (8 * ch)) & 255u;
                    if ((unsigned)gy < (unsigned)srows && b >= 0 && b < rowb) px = src[(int64_t)gy * sstep + b];
                    v |= px << (8 * j);
                }
            }
            *(uint32_t *)&box[ly * bwb + lq * 4] = v;
        }
    }
    __syncthreads();
    const int lx = (threadIdx.x & 15) * 4, ly = threadIdx.x >> 4;
    const int x0 = tx0 + lx, y = ty0 + ly;
    if (x0 >= dcols || y >= drows) return;
    const int X0 = (int)rint((M[1] * (double)y + M[2]) * 1024.0) + rd;
    const int Y0 = (int)rint((M[4] * (double)y + M[5]) * 1024.0) + rd;
    uint8_t o[4 * CN];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        const int Xf = X0 + (int)rint(M[0] * (double)x * 1024.0), Yf = Y0 + (int)rint(M[3] * (double)x * 1024.0);
        if (!staged) {
#pragma unroll
            for (int c = 0; c < CN; c++)
                o[j * CN + c] = (uint8_t)warp_tap_global<LINEAR>(src + c, sstep, srows, scols, CN, Xf, Yf, (border_rgba >> (8 * c)) & 255);
        } else if (!LINEAR) {
            const uint8_t *B = &box[((Yf >> 10) - by0) * bwb + (Xf >> 10) * CN - bb0];
#pragma unroll
            for (int c = 0; c < CN; c++) o[j * CN + c] = B[c];
        } else {
            const int X = Xf >> 5, Y = Yf >> 5;
            const int fx = X & 31, fy = Y & 31;
            const uint8_t *B = &box[((Y >> 5) - by0) * bwb + (X >> 5) * CN - bb0];
            const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
#pragma unroll
            for (int c = 0; c < CN; c++)
                o[j * CN + c] = st_sat_u8((B[c] * w0 + B[CN + c] * w1 + B[bwb + c] * w2 + B[bwb + CN + c] * w3 + (1 << 14)) >> 15);
        }
    }
    uint8_t *D = dst + (int64_t)y * dstep + (int64_t)x0 * CN;
    if (x0 + 4 <= dcols && ((uintptr_t)D & 3) == 0) {  // packed CONTAIN canvases have odd widths: rows start anywhere
#pragma unroll
        for (int q = 0; q < CN; q++)
            ((uint32_t *)D)[q] = (uint32_t)o[4 * q] | ((uint32_t)o[4 * q + 1] << 8) | ((uint32_t)o[4 * q + 2] << 16) |
                                 ((uint32_t)o[4 * q + 3] << 24);
    } else {
        for (int j = 0; j < 4 * CN && x0 * CN + j < dcols * CN; j++) D[j] = o[j];
    }
}

// 1- and 3-channel warpAffine (NEAREST / LINEAR) through the LDS-staged kernel: the gray helpers and the final
// deskew of the colour scan (omr.rs:408-445, core/src/main.rs:72-81).  Returns hipErrorInvalidValue for other
// channel counts (the caller uses the generic kernel).
hipError_t launch_warp_fast(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                            int64_t dstep, int drows, int dcols, const double Minv[6], int interp, uint32_t border_rgba,
                            hipStream_t s)
{
    if (cn != 1 && cn != 3) return hipErrorInvalidValue;
    WarpM W;
    for (int i = 0; i < 6; i++) W.m[i] = Minv[i];
    dim3 grid((dcols + WL_TW - 1) / WL_TW, (drows + WL_TH - 1) / WL_TH);
#define WARP_LAUNCH(CN_, LIN_)                                                                                         \
    hipLaunchKernelGGL((warp_lds_kernel<CN_, LIN_>), grid, dim3(256), 0, s, d_src, sstep, srows, scols, d_dst, dstep, \
                       drows, dcols, W, border_rgba)
    if (cn == 1 && interp == 0) WARP_LAUNCH(1, false);
    else if (cn == 1) WARP_LAUNCH(1, true);
    else if (interp == 0) WARP_LAUNCH(3, false);
    else WARP_LAUNCH(3, true);
#undef WARP_LAUNCH
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
