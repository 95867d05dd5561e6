// deskew.hip -- the batch's last stage: rotate every scan by the angle its sweep found, CONTAIN geometry
// (transfer.rs:487-519), NEAREST as correct_default does (omr.rs:408-445) or LINEAR as the benchmark drivers
// do (core/src/main.rs:72-81, app/src-tauri/src/test.rs:322-331).  The winning candidate's index is read on the
// device, so sweep -> arg-max -> warp need no host round trip; the fixed-point tables of OpenCV's warpAffine
// (adelta / bdelta per destination column, X0 / Y0 per destination row, AB_BITS = 10) exist per CANDIDATE, built
// once per batch context from host-computed matrices (fp64, same libm as the oracle), so a tap costs a few
// integer operations and the result is bit-identical to omr_rotate_device on the same scan.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace omr {

#define DW_TW 128  // destination tile: 128 x 32 pixels per 256-thread workgroup, 4 x 4 pixels per thread
#define DW_TH 32
#define DW_LDS 24576

__device__ __forceinline__ uint8_t dw_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

template <bool LINEAR>
__device__ __forceinline__ int dw_tap_global(const uint8_t *__restrict__ src, int64_t sstep, int srows, int scols, int Xf,
                                             int Yf, int border)
{
    if (!LINEAR) {
        const int X = max(-32768, min(32767, Xf >> 10)), Y = max(-32768, min(32767, Yf >> 10));
        return ((unsigned)X < (unsigned)scols && (unsigned)Y < (unsigned)srows) ? src[(int64_t)Y * sstep + X] : border;
    }
    const int X = Xf >> 5, Y = Yf >> 5;
    const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
    const int fx = X & 31, fy = Y & 31;
    if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) return border;
    const bool in_x0 = sx >= 0 && sx < scols, in_x1 = sx + 1 >= 0 && sx + 1 < scols;
    const bool in_y0 = sy >= 0 && sy < srows, in_y1 = sy + 1 >= 0 && sy + 1 < srows;
    const int v0 = in_x0 && in_y0 ? src[(int64_t)sy * sstep + sx] : border;
    const int v1 = in_x1 && in_y0 ? src[(int64_t)sy * sstep + sx + 1] : border;
    const int v2 = in_x0 && in_y1 ? src[(int64_t)(sy + 1) * sstep + sx] : border;
    const int v3 = in_x1 && in_y1 ? src[(int64_t)(sy + 1) * sstep + sx + 1] : border;
    const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    return dw_sat_u8((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
}

// grid = (tiles across the largest canvas, tiles down it, scans of the launch); a tile outside its scan's canvas
// leaves at once.  The source bounding box of a tile (four corner samples; one pixel of slack for the rounding of
// the fixed-point tables, one more for the bilinear taps) is staged in LDS with row-contiguous dword loads, border
// value outside the image, so a tap is one LDS byte read with no bounds test.
template <bool LINEAR>
__global__ __launch_bounds__(256) void deskew_warp_kernel(const DeskewPass p)
{
    __shared__ __attribute__((aligned(16))) uint8_t box[DW_LDS];
    const int z = blockIdx.z;
    const int a = __builtin_amdgcn_readfirstlane(p.best[z]);
    const int drows = p.wsize[2 * a], dcols = p.wsize[2 * a + 1];
    const int tx0 = blockIdx.x * DW_TW, ty0 = blockIdx.y * DW_TH;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && p.out_size) {
        p.out_size[2 * z] = drows;
        p.out_size[2 * z + 1] = dcols;
    }
    if (tx0 >= dcols || ty0 >= drows) return;
    const uint8_t *__restrict__ src = p.src + (int64_t)z * p.scan_stride;
    uint8_t *__restrict__ dst = p.dst + (int64_t)z * p.out_stride;
    const int32_t *__restrict__ AD = p.adelta + (int64_t)a * p.DC, *__restrict__ BD = p.bdelta + (int64_t)a * p.DC;
    const int2_t *__restrict__ XY = p.xy0 + (int64_t)a * p.DR;
    const int rd = LINEAR ? 16 : 512;
    const int tx1 = min(dcols, tx0 + DW_TW) - 1, ty1 = min(drows, ty0 + DW_TH) - 1;
    // fixed-point source coordinates of the tile's corner samples: wave-uniform.  X0(y) and adelta(x) are both
    // monotone, so the four corners bound every sample of the tile.
    const int2_t r0 = XY[ty0], r1 = XY[ty1];
    const int a0 = AD[tx0], a1 = AD[tx1], b0 = BD[tx0], b1 = BD[tx1];
    const int cx[4] = {(r0.x + rd + a0) >> 10, (r0.x + rd + a1) >> 10, (r1.x + rd + a0) >> 10, (r1.x + rd + a1) >> 10};
    const int cy[4] = {(r0.y + rd + b0) >> 10, (r0.y + rd + b1) >> 10, (r1.y + rd + b0) >> 10, (r1.y + rd + b1) >> 10};
    const int bx0 = min(min(cx[0], cx[1]), min(cx[2], cx[3])) - 1;
    const int bx1 = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 1 + (LINEAR ? 1 : 0);
    const int by0 = min(min(cy[0], cy[1]), min(cy[2], cy[3])) - 1;
    const int by1 = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 1 + (LINEAR ? 1 : 0);
    const int bb0 = bx0 & ~3, bb1 = (bx1 + 4) & ~3;  // the box in bytes of a source row, widened to whole dwords: [bb0, bb1)
    const int bwb = bb1 - bb0, bh = by1 - by0 + 1;
    const bool staged = bwb > 0 && bh > 0 && (int64_t)bwb * bh <= DW_LDS && bx0 > -30000 && bx1 < 30000 && by0 > -30000 &&
                        by1 < 30000;
    const uint32_t border4 = (uint32_t)p.border * 0x01010101u;
    if (staged) {
        const int bq = bwb >> 2;
        const bool aligned = ((p.sstep | p.scan_stride | (int64_t)(uintptr_t)p.src) & 3) == 0;
        for (int i = threadIdx.x; i < bq * bh; i += 256) {
            const int ly = i / bq, lq = i - ly * bq;
            const int gy = by0 + ly, gb = bb0 + lq * 4;
            uint32_t v;
            if (aligned && (unsigned)gy < (unsigned)p.srows && gb >= 0 && gb + 4 <= p.scols) {
                v = *(const uint32_t *)(src + (int64_t)gy * p.sstep + gb);
            } else if ((unsigned)gy >= (unsigned)p.srows || gb + 4 <= 0 || gb >= p.scols) {
                v = border4;
            } else {
                v = 0;
                for (int j = 0; j < 4; j++) {
                    const int b = gb + j;
                    const uint32_t px = (b >= 0 && b < p.scols) ? src[(int64_t)gy * p.sstep + b] : (uint32_t)p.border;
                    v |= px << (8 * j);
                }
            }
            *(uint32_t *)&box[ly * bwb + lq * 4] = v;
        }
    }
    __syncthreads();
    const int lx = (threadIdx.x & 31) * 4, x0 = tx0 + lx;
    if (x0 >= dcols) return;
    // the thread's four columns: DC is a multiple of 4 and the tables are 16-byte aligned
    const int4 ad = *(const int4 *)(AD + x0), bd = *(const int4 *)(BD + x0);
    const int adv[4] = {ad.x, ad.y, ad.z, ad.w}, bdv[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
    for (int k = 0; k < DW_TH / 8; k++) {
        const int y = ty0 + (threadIdx.x >> 5) + 8 * k;
        if (y >= drows) break;
        const int2_t r = XY[y];
        const int X0 = r.x + rd, Y0 = r.y + rd;
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Xf = X0 + adv[j], Yf = Y0 + bdv[j];
            int v;
            if (!staged) {
                v = dw_tap_global<LINEAR>(src, p.sstep, p.srows, p.scols, Xf, Yf, p.border);
            } else if (!LINEAR) {
                v = box[((Yf >> 10) - by0) * bwb + (Xf >> 10) - bb0];
            } else {
                const int X = Xf >> 5, Y = Yf >> 5;
                const int fx = X & 31, fy = Y & 31;
                const uint8_t *B = &box[((Y >> 5) - by0) * bwb + (X >> 5) - bb0];
                const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
                v = dw_sat_u8((B[0] * w0 + B[1] * w1 + B[bwb] * w2 + B[bwb + 1] * w3 + (1 << 14)) >> 15);
            }
            out |= (uint32_t)v << (8 * j);
        }
        uint8_t *D = dst + (int64_t)y * p.dstep + x0;
        if (x0 + 4 <= dcols && ((uintptr_t)D & 3) == 0) {
            *(uint32_t *)D = out;
        } else {
            for (int j = 0; j < 4 && x0 + j < dcols; j++) D[j] = (uint8_t)(out >> (8 * j));
        }
    }
}

hipError_t launch_deskew_warp(const DeskewPass &p, int scans, int interp, hipStream_t s)
{
    if (scans <= 0) return hipSuccess;
    if ((p.DC & 3) != 0) return hipErrorInvalidValue;
    dim3 grid((p.DC + DW_TW - 1) / DW_TW, (p.DR + DW_TH - 1) / DW_TH, scans);
    if (interp == 0) hipLaunchKernelGGL(deskew_warp_kernel<false>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(deskew_warp_kernel<true>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace omr
