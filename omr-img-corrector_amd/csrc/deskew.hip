// deskew.hip -- the batch's last stage: rotate every scan by the angle its sweep found, CONTAIN geometry
// (transfer.rs:487-519), NEAREST as correct_default does (omr.rs:408-445) or LINEAR as the benchmark drivers
// do (core/src/main.rs:72-81, app/src-tauri/src/test.rs:322-331).  The winning candidate's index is read on the
// device, so sweep -> arg-max -> warp need no host round trip; the fixed-point tables of OpenCV's warpAffine
// (adelta / bdelta per destination column, X0 / Y0 per destination row, AB_BITS = 10) exist per CANDIDATE, built
// once per batch context from host-computed matrices (fp64, same libm as the oracle), so a tap costs a few
// integer operations and the result is bit-identical to omr_rotate_device on the same scan.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.hpp"

namespace omr {

#define DW_TW 128  // destination tile: 128 x 32 pixels per 256-thread workgroup, 4 x 4 pixels per thread
// tile height: 64 rows for NEAREST (the fixed chain of memory round trips per tile is spread over more pixels),
// 32 for LINEAR (four taps per pixel: at 64 rows the pixels in flight cost 200 VGPRs)
#define DW_TH (LINEAR ? 32 : 64)
#define DW_LDS 16384

__device__ __forceinline__ int dw_mad24(int a, int b, int c) { return __mul24(a, b) + c; }  // v_mad_i32_i24
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
// leaves at once.  (Scans fastest instead -- XCD x warps only scan x, so overlapping boxes meet in one L2 -- halves the
// HBM fetch, 71.8 -> 34.3 MB per 8 scans = the scans' own bytes, but is no faster: 65 / 114 us against 66 / 105.  The
// kernel is not waiting for HBM.  Nor for its chain of round trips alone: a software-pipelined form -- a vertical strip of
// four 128 x 32 tiles per workgroup, tile i sampled from one LDS box while the box of tile i + 1 is in flight and the
// table entries of tile i + 2 are requested -- was built, bit-exact, and slower: 103 / 138 us at 112 / 168 VGPRs and
// 32 KB of LDS per workgroup.)  The source bounding box of a tile (four corner samples; one pixel of slack for the rounding of
// the fixed-point tables, one more for the bilinear taps) is staged in LDS with row-contiguous dword loads, border
// value outside the image, so a tap is one LDS byte read with no bounds test.  The kernel is a chain of dependent
// memory round trips (winner -> canvas size -> table entries of the corners -> box -> taps), so everything that
// does not depend on the box -- the thread's own column and row table entries -- is requested before the box is.
template <bool LINEAR>
// (forcing more waves per SIMD was measured and lost: __launch_bounds__(256, 6) spills, 127 / 126 us per 8 A4 scans
// against 62 / 104 at the compiler's own 100 / 92 VGPRs)
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
    // ---- requests that need nothing but the tile position: the corners' and this thread's table entries
    const int2_t r0 = XY[ty0], r1 = XY[ty1];
    const int a0 = AD[tx0], a1 = AD[tx1], b0 = BD[tx0], b1 = BD[tx1];
    const int lx = (threadIdx.x & 31) * 4, x0 = tx0 + lx, yq = ty0 + (threadIdx.x >> 5);
    const int xl = min(x0, p.DC - 4);  // DC is a multiple of 4 and the tables are 16-byte aligned
    const int4 ad = *(const int4 *)(AD + xl), bd = *(const int4 *)(BD + xl);
    // (rows past the tile's last one repeat it and columns past its last one repeat that column below: every sample a
    // thread computes then lies inside the tile's box, stored or not, and a tap needs no bounds test)
    int2_t rows[DW_TH / 8];
#pragma unroll
    for (int k = 0; k < DW_TH / 8; k++) rows[k] = XY[min(yq + 8 * k, ty1)];
    // ---- the box: fixed-point source coordinates of the tile's corner samples (wave-uniform).  X0(y) and
    // adelta(x) are both monotone, so the four corners bound every sample of the tile.
    const int cx[4] = {(r0.x + rd + a0) >> 10, (r0.x + rd + a1) >> 10, (r1.x + rd + a0) >> 10, (r1.x + rd + a1) >> 10};
    const int cy[4] = {(r0.y + rd + b0) >> 10, (r0.y + rd + b1) >> 10, (r1.y + rd + b0) >> 10, (r1.y + rd + b1) >> 10};
    const int bx0 = min(min(cx[0], cx[1]), min(cx[2], cx[3])) - 1;
    const int bx1 = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 1 + (LINEAR ? 1 : 0);
    const int by0 = min(min(cy[0], cy[1]), min(cy[2], cy[3])) - 1;
    const int by1 = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 1 + (LINEAR ? 1 : 0);
    const int bb0 = bx0 & ~3, bb1 = (bx1 + 4) & ~3;  // the box in bytes of a source row, widened to whole dwords: [bb0, bb1)
    const int bwb = bb1 - bb0, bh = by1 - by0 + 1;
    // (a scan whose rows are not whole aligned dwords -- width, pitch or address not a multiple of 4 -- takes the
    // unstaged path below: the staging loop then has no partial dwords and no branches)
    const bool dwords = ((p.sstep | p.scan_stride | (int64_t)(uintptr_t)p.src | (int64_t)p.scols) & 3) == 0;
    const bool staged = dwords && bwb > 0 && bh > 0 && (int64_t)bwb * bh <= DW_LDS && bx0 > -30000 && bx1 < 30000 &&
                        by0 > -30000 && by1 < 30000;
    const uint32_t border4 = (uint32_t)p.border * 0x01010101u;
    // A tile whose box lies wholly outside the scan (the corners of a CONTAIN canvas: up to a fifth of it at 10 degrees) is
    // the border value: every tap of every sample is outside, NEAREST takes it as it is and the bilinear weights of four
    // equal taps sum to 2^15.  Workgroup-uniform: the four corners bound every sample of the tile.
    if (bx1 < 0 || by1 < 0 || bx0 >= p.scols || by0 >= p.srows) {
        if (x0 >= dcols) return;
        const bool whole4 = x0 + 4 <= dcols && ((p.dstep | p.out_stride | (int64_t)(uintptr_t)p.dst) & 3) == 0;
#pragma unroll
        for (int k = 0; k < DW_TH / 8; k++) {
            uint8_t *D = dst + (int64_t)min(yq + 8 * k, ty1) * p.dstep + x0;
            if (whole4) {
                *(uint32_t *)D = border4;
            } else {
                for (int j = 0; j < 4 && x0 + j < dcols; j++) D[j] = (uint8_t)p.border;
            }
        }
        return;
    }
    if (staged) {
        // the box as dwords, thread t takes dwords t, t + 256, ..: ALL loads are issued before the first LDS write, and
        // every one of them unconditionally (a dword outside the image reads the scan's first dword and is replaced by
        // the border value afterwards) -- a load inside a divergent branch is waited for before the next is issued, a
        // memory round trip per dword.  (row, dword in row) of a thread's next piece follows from the previous one
        // without a division.  (Skipping the pieces a small box does not need -- a 128 x 32 tile at 5 degrees needs 7 of the
        // 16 -- by a scalar branch per piece was measured in the bench: 5.4 k deskewed images/s against 10.1 k; the branches
        // serialise the loads just like divergent ones.)
        constexpr int NP = DW_LDS / 4 / 256;  // 16 pieces per thread at most
        const int bq = bwb >> 2, total = bq * bh;
        const int dq = 256 / bq, dr = 256 - dq * bq;  // 256 = dq * bq + dr
        // ... but ONE branch on the (workgroup-uniform) size of the box, in front of three straight-line versions (8, 12, 16
        // pieces), costs nothing: most LINEAR tiles of a small-angle warp need at most half the pieces (+1.8 % deskewed
        // images/s in the bench)
        auto stage = [&](auto np_tag) {
            constexpr int NPV = decltype(np_tag)::value;
            int ly = (int)threadIdx.x / bq, lq = (int)threadIdx.x - ly * bq;
            uint32_t v[NPV];
            bool in[NPV];
#pragma unroll
            for (int n = 0; n < NPV; n++) {
                const int gy = by0 + ly, gb = bb0 + lq * 4;
                in[n] = (int)threadIdx.x + n * 256 < total && (unsigned)gy < (unsigned)p.srows && (unsigned)gb < (unsigned)p.scols;
                v[n] = *(const uint32_t *)(src + (in[n] ? (int64_t)gy * p.sstep + gb : 0));
                ly += dq, lq += dr;
                if (lq >= bq) lq -= bq, ly++;
            }
#pragma unroll
            for (int n = 0; n < NPV; n++)
                if ((int)threadIdx.x + n * 256 < total) *(uint32_t *)&box[((int)threadIdx.x + n * 256) * 4] = in[n] ? v[n] : border4;
        };
        if (total <= (NP / 2) * 256) stage(std::integral_constant<int, NP / 2>{});
        else if (total <= (3 * NP / 4) * 256) stage(std::integral_constant<int, 3 * NP / 4>{});
        else stage(std::integral_constant<int, NP>{});
    }
    __syncthreads();
    if (x0 >= dcols) return;
    int adv[4] = {ad.x, ad.y, ad.z, ad.w}, bdv[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {  // (xl == x0 unless the thread starts past the tables' last four columns, and then past tx1 too)
        const bool past = x0 + j > tx1;
        adv[j] = past ? a1 : adv[j];
        bdv[j] = past ? b1 : bdv[j];
    }
    if (!staged) {  // workgroup-uniform: a box too large for LDS (strong shear) or rows that are not whole dwords; taps from global memory
        for (int k = 0; k < DW_TH / 8; k++) {
            const int y = yq + 8 * k;
            if (y >= drows) break;
            uint8_t *D = dst + (int64_t)y * p.dstep + x0;
            for (int j = 0; j < 4 && x0 + j < dcols; j++)
                D[j] = (uint8_t)dw_tap_global<LINEAR>(src, p.sstep, p.srows, p.scols, rows[k].x + rd + adv[j], rows[k].y + rd + bdv[j],
                                                      p.border);
        }
        return;
    }
    // box[(sy - by0) * bwb + (sx - bb0)]: the box's origin rides on the row's X0 / Y0 (multiples of 1024: the fraction
    // bits stay; |coordinates| < 30000 pixels, so nothing overflows).  Straight-line: the 16 / 32 pixels' taps are all requested before the first is blended,
    // and nothing below is conditional except the last tile column's byte stores -- a row past the tile's last one
    // recomputes that row's pixels (its table entry was clamped) and stores them there again, the same bytes.
    const int orgx = rd - (bb0 << 10), orgy = rd - (by0 << 10);
    uint32_t outs[DW_TH / 8];
#pragma unroll
    for (int k = 0; k < DW_TH / 8; k++) {
        const int X0 = rows[k].x + orgx, Y0 = rows[k].y + orgy;
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int Xf = X0 + adv[j], Yf = Y0 + bdv[j];
            const int idx = dw_mad24(Yf >> 10, bwb, Xf >> 10);
            int v;
            if (!LINEAR) {
                v = box[idx];
            } else {
                // ((v0 w0 + v1 w1 + v2 w2 + v3 w3 + 2^14) >> 15 with w = 32 (32 - fy)(32 - fx), ..) as two horizontal
                // blends and a vertical one: the same integer, and <= 255 without a clamp
                const int fx = (Xf >> 5) & 31, fy = (Yf >> 5) & 31;
                const uint8_t *B = &box[idx];
                const int b0 = B[0], b1 = B[1], b2 = B[bwb], b3 = B[bwb + 1];
                const int top = dw_mad24(fx, b1 - b0, b0 << 5), bot = dw_mad24(fx, b3 - b2, b2 << 5);
                v = dw_mad24(fy, bot - top, (top << 5) + 512) >> 10;
            }
            out |= (uint32_t)v << (8 * j);
        }
        outs[k] = out;
    }
    const bool whole = x0 + 4 <= dcols && ((p.dstep | p.out_stride | (int64_t)(uintptr_t)p.dst) & 3) == 0;
#pragma unroll
    for (int k = 0; k < DW_TH / 8; k++) {
        uint8_t *D = dst + (int64_t)min(yq + 8 * k, ty1) * p.dstep + x0;
        if (whole) {
            *(uint32_t *)D = outs[k];
        } else {
            for (int j = 0; j < 4 && x0 + j < dcols; j++) D[j] = (uint8_t)(outs[k] >> (8 * j));
        }
    }
}

hipError_t launch_deskew_warp(const DeskewPass &p, int scans, int interp, hipStream_t s)
{
    if (scans <= 0) return hipSuccess;
    if ((p.DC & 3) != 0) return hipErrorInvalidValue;
    const bool LINEAR = interp != 0;
    dim3 grid((p.DC + DW_TW - 1) / DW_TW, (p.DR + DW_TH - 1) / DW_TH, scans);
    if (interp == 0) hipLaunchKernelGGL(deskew_warp_kernel<false>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(deskew_warp_kernel<true>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace omr
