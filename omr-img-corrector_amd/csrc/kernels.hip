// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the OMR deskew engine.
//
// Hot path (BASELINE.json north_star): for every candidate angle, nearest-neighbour affine
// rotation of the binarised scan, per-column / per-row black-pixel counts, and the population
// std-dev of both projections.  Reference formulation: packages/lib/src/projection.rs:47-65,
// transfer.rs:459-486 (rotate_mat DEFAULT) + :527-536, omr.rs:8-39, calculate.rs:13-23, with
// OpenCV 4.6.0's warpAffine(INTER_NEAREST) fixed-point map.
//
// Design: the rotated image is never materialised.  The scan is bit-packed once (1.09 MB for
// A4, L2-resident); one wave owns 64 destination columns (lane = column, adelta/bdelta in
// registers) and walks destination rows (X0/Y0 arrive as scalar loads), so a row's 64 samples
// cost a handful of integer VALU ops, one LDS (or L1) gather and one ballot.  No MFMA: this is
// integer remap + count, not a contraction.  Built with -ffp-contract=off: the table kernel
// must round exactly like OpenCV's scalar code.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace omr {

#define OMR_WAVE 64

// v_writelane_b32: park a wave-uniform value in lane `lane` of a VGPR (one VALU op; the clang
// builtin is not exposed by this ROCm's hipcc).
__device__ __forceinline__ int write_lane(int vreg, int value, int lane)
{
    // gfx9 reads one SGPR per VALU op over the constant bus; the lane select goes through M0
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vreg) : "s"(value), "s"(lane) : "m0");
    return vreg;
}

// ------------------------------------------------------------------------------------------
// pack: u8 -> 1 bit / pixel.  One wave ballots 64 consecutive pixels of a row into two words.
// Fuses transfer_gray_image_to_thresh_binary (transfer.rs:294-301) when black_max = 127:
// threshold(127,255,BINARY) leaves 0 exactly where gray <= 127, and the projections count == 0.
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ img, int64_t step,
                                                        int rows, int cols, int black_max,
                                                        uint32_t *__restrict__ bits, int wpr, int64_t img_stride)
{
    img += (int64_t)blockIdx.z * img_stride;  // blockIdx.z = scan of the launch
    bits += (int64_t)blockIdx.z * rows * wpr;
    const int y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    bool black = false;
    if (x < cols) black = (int)img[(int64_t)y * step + x] <= black_max;
    const unsigned long long m = __ballot(black);
    const int lane = threadIdx.x & 63;
    const int w = x >> 5;  // word index of this lane's pixel
    if ((lane & 31) == 0 && w < wpr) bits[(int64_t)y * wpr + w] = (uint32_t)(lane ? (m >> 32) : m);
}

// Wide form for 16-byte aligned rows: a lane loads 16 pixels (one dwordx4), turns each dword
// into a 4-bit mask with a SWAR "byte < n" test (n = black_max + 1 <= 128) and a multiply that
// gathers the four byte flags, and a lane pair assembles one 32-bit word.  8.7 MB A4 scans pack at
// HBM speed instead of at one byte load per lane.
__device__ __forceinline__ uint32_t bytes_lt_nibble(uint32_t x, uint32_t n_rep)
{
    // 0x80 in every byte of x that is < n (1 <= n <= 128), exact per byte: setting bit 7 first keeps
    // the subtraction free of cross-byte borrows; its bit 7 then says (x & 0x7f) >= n
    const uint32_t t = (x | 0x80808080u) - n_rep;
    const uint32_t m = ~x & ~t & 0x80808080u;
    return (((m >> 7) * 0x01020408u) >> 24) & 15u;  // byte flags 0,8,16,24 -> bits 0..3
}

__global__ __launch_bounds__(256) void pack_bits16_kernel(const uint8_t *__restrict__ img, int64_t step, int rows,
                                                          int cols, int black_max, uint32_t *__restrict__ bits,
                                                          int wpr, int64_t img_stride)
{
    img += (int64_t)blockIdx.z * img_stride;  // blockIdx.z = scan of the launch
    bits += (int64_t)blockIdx.z * rows * wpr;
    const int y = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;  // 16-pixel group of this lane
    const int x0 = q * 16;
    uint32_t m16 = 0;
    if (x0 + 16 <= cols) {
        const uint4 v = *(const uint4 *)(img + (int64_t)y * step + x0);
        const uint32_t n_rep = (uint32_t)(black_max + 1) * 0x01010101u;
        m16 = bytes_lt_nibble(v.x, n_rep) | (bytes_lt_nibble(v.y, n_rep) << 4) | (bytes_lt_nibble(v.z, n_rep) << 8) |
              (bytes_lt_nibble(v.w, n_rep) << 12);
    } else if (x0 < cols) {
        for (int j = 0; j < cols - x0; j++) m16 |= ((int)img[(int64_t)y * step + x0 + j] <= black_max ? 1u : 0u) << j;
    }
    const uint32_t other = __shfl_xor(m16, 1);
    const int w = q >> 1;
    if ((threadIdx.x & 1) == 0 && w < wpr) bits[(int64_t)y * wpr + w] = m16 | (other << 16);
}

hipError_t launch_pack_bits(const uint8_t *d_img, int64_t step, int rows, int cols, int black_max,
                            uint32_t *d_bits, int wpr, hipStream_t s, int scans, int64_t img_stride)
{
    if (black_max >= 0 && black_max < 128 && (step & 15) == 0 && ((uintptr_t)d_img & 15) == 0 &&
        (scans == 1 || (img_stride & 15) == 0)) {
        dim3 grid((wpr * 2 + 255) / 256, rows, scans);
        hipLaunchKernelGGL(pack_bits16_kernel, grid, dim3(256), 0, s, d_img, step, rows, cols, black_max, d_bits, wpr,
                           img_stride);
        return hipGetLastError();
    }
    dim3 grid((wpr * 32 + 255) / 256, rows, scans);
    hipLaunchKernelGGL(pack_bits_kernel, grid, dim3(256), 0, s, d_img, step, rows, cols, black_max, d_bits, wpr,
                       img_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// OpenCV hal::warpAffine tables (AB_BITS = 10): adelta[x] = cvRound(M0*x*1024),
// bdelta[x] = cvRound(M3*x*1024), X0[y] = cvRound((M1*y + M2)*1024) + round_delta, Y0 likewise.
// cvRound = round-half-even = rint(); no FMA contraction (file is built -ffp-contract=off).
__global__ __launch_bounds__(256) void tables_kernel(const double *__restrict__ Minv, SweepDims d, int round_delta,
                                                     int32_t *__restrict__ adelta, int32_t *__restrict__ bdelta,
                                                     int2_t *__restrict__ xy0, int32_t *__restrict__ overflow)
{
    const int a = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double *M = Minv + 6 * a;
    const double lim = 1073741824.0;  // 2^30: keeps X0 + adelta inside int32
    if (i < d.cols) {
        const double x = (double)i;
        const double va = rint(M[0] * x * 1024.0);
        const double vb = rint(M[3] * x * 1024.0);
        if (!(fabs(va) < lim) || !(fabs(vb) < lim)) *overflow = 1;
        adelta[(int64_t)a * d.cols + i] = (int32_t)va;
        bdelta[(int64_t)a * d.cols + i] = (int32_t)vb;
    } else if (i - d.cols < d.rows) {
        const int yy = i - d.cols;
        const double y = (double)yy;
        const double vx = rint((M[1] * y + M[2]) * 1024.0);
        const double vy = rint((M[4] * y + M[5]) * 1024.0);
        if (!(fabs(vx) < lim) || !(fabs(vy) < lim)) *overflow = 1;
        int2_t v;
        v.x = (int32_t)vx + round_delta;
        v.y = (int32_t)vy + round_delta;
        xy0[(int64_t)a * d.rows + yy] = v;
    }
}

hipError_t launch_tables(const double *d_Minv, SweepDims d, int round_delta, int32_t *d_adelta, int32_t *d_bdelta,
                         int2_t *d_xy0, int32_t *d_overflow, hipStream_t s)
{
    if (d.A <= 0) return hipSuccess;
    dim3 grid((d.cols + d.rows + 255) / 256, d.A);
    hipLaunchKernelGGL(tables_kernel, grid, dim3(256), 0, s, d_Minv, d, round_delta, d_adelta, d_bdelta, d_xy0,
                       d_overflow);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Generic sweep kernel: handles any matrix.  Block = GEN_WAVES waves side by side (lane =
// destination column), each block walks a band of destination rows.  Per row: scalar load of
// (X0, Y0); per lane the fixed-point add/shift, an unsigned bounds test (remapNearest's
// (unsigned)sx < cols && (unsigned)sy < rows; saturate_cast<short> cannot change the outcome
// because every dimension is < 32767) and a gather of the source word from the L2/L1-resident
// bit image.  Row counts: ballot + s_bcnt1, parked in lane r of a VGPR and flushed through LDS.
#define GEN_WAVES 8
#define GEN_BAND 512

__global__ __launch_bounds__(GEN_WAVES *OMR_WAVE) void sweep_generic_kernel(
    const uint32_t *__restrict__ bits, SweepDims d, const int32_t *__restrict__ adelta,
    const int32_t *__restrict__ bdelta, const int2_t *__restrict__ xy0, const int32_t *__restrict__ list,
    uint32_t *__restrict__ vproj, uint32_t *__restrict__ hproj, int ntx)
{
    __shared__ uint32_t hacc[GEN_BAND];
    const int a = list ? list[blockIdx.z] : (int)blockIdx.z;
    const int lane = threadIdx.x & 63;
    // blockIdx.x = scan of the launch * ntx + tile across: the scans of a launch group share one launch (one launch per scan
    // left a 248 x 230 sheet's 319 gathered candidates with 319 half-empty workgroups per launch)
    const int zs = blockIdx.x / ntx, bx = blockIdx.x - zs * ntx;
    bits += (int64_t)zs * d.rows * d.wpr;
    vproj += (int64_t)zs * d.A * d.cols;
    hproj += (int64_t)zs * d.A * d.rows;
    const int x = bx * (GEN_WAVES * OMR_WAVE) + threadIdx.x;
    const int y0 = blockIdx.y * GEN_BAND;
    const int y1 = min(d.rows, y0 + GEN_BAND);
    const bool active = x < d.cols;
    const int xc = active ? x : d.cols - 1;
    const int ad = adelta[(int64_t)a * d.cols + xc];
    const int bd = bdelta[(int64_t)a * d.cols + xc];
    const int2_t *__restrict__ s0p = xy0 + (int64_t)a * d.rows;

    for (int i = threadIdx.x; i < GEN_BAND; i += GEN_WAVES * OMR_WAVE) hacc[i] = 0;
    __syncthreads();

    uint32_t vacc = 0;
    for (int yc = y0; yc < y1; yc += OMR_WAVE) {
        const int nr = min(OMR_WAVE, y1 - yc);
        int hrow = 0;
        for (int r = 0; r < nr; r++) {
            const int2_t s0 = s0p[yc + r];  // wave-uniform -> scalar load
            const int X = (s0.x + ad) >> 10;
            const int Y = (s0.y + bd) >> 10;
            const bool inb = active && (unsigned)X < (unsigned)d.cols && (unsigned)Y < (unsigned)d.rows;
            uint32_t bit = 0;
            if (inb) bit = (bits[(int64_t)Y * d.wpr + (X >> 5)] >> (X & 31)) & 1u;
            vacc += bit;
            const unsigned long long m = __ballot(bit != 0);
            hrow = write_lane(hrow, __popcll(m), r);
        }
        if (lane < nr && hrow) atomicAdd(&hacc[yc - y0 + lane], (uint32_t)hrow);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < y1 - y0; i += GEN_WAVES * OMR_WAVE) {
        const uint32_t v = hacc[i];
        if (v) atomicAdd(&hproj[(int64_t)a * d.rows + y0 + i], v);
    }
    if (active && vacc) atomicAdd(&vproj[(int64_t)a * d.cols + x], vacc);
}

hipError_t launch_sweep_generic(const uint32_t *d_bits, SweepDims d, const int32_t *d_adelta,
                                const int32_t *d_bdelta, const int2_t *d_xy0, const int32_t *d_list, int n_list,
                                uint32_t *d_vproj, uint32_t *d_hproj, hipStream_t s, int scans)
{
    const int nz = d_list ? n_list : d.A;
    if (nz <= 0 || scans <= 0) return hipSuccess;
    const int ntx = (d.cols + GEN_WAVES * OMR_WAVE - 1) / (GEN_WAVES * OMR_WAVE);
    dim3 grid(ntx * scans, (d.rows + GEN_BAND - 1) / GEN_BAND, nz);
    hipLaunchKernelGGL(sweep_generic_kernel, grid, dim3(GEN_WAVES * OMR_WAVE), 0, s, d_bits, d, d_adelta, d_bdelta,
                       d_xy0, d_list, d_vproj, d_hproj, ntx);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LDS-staged sweep kernel.  Same wave decomposition, but every wave first copies the source
// window its next rows_per_tile destination rows can touch (bounding box of the four corner
// samples: the map is monotone in x and in y separately) from the bit image into a private LDS
// slab, zero-filled outside the image, so the inner loop has no bounds test and gathers from
// LDS (ds_read_b32, odd row pitch).  Waves never share a slab: no block barrier in the loop.
#define LDS_WAVES 8
#define LDS_BAND 512
#define LDS_SLAB_WORDS 1024  // per-wave window budget (4 KiB)

// (LW waves = 64 LW columns per workgroup: 8, or 4 for images of at most 256 columns -- the app's 248 x 230 working size --
// whose other four waves would sit idle on 16 KB of LDS slabs)
template <int LW>
__global__ __launch_bounds__(LW *OMR_WAVE) void sweep_lds_kernel(
    const uint32_t *__restrict__ bits, SweepDims d, const int32_t *__restrict__ adelta,
    const int32_t *__restrict__ bdelta, const int2_t *__restrict__ xy0, const LdsTile *__restrict__ tiles,
    const int32_t *__restrict__ list, uint32_t *__restrict__ vproj, uint32_t *__restrict__ hproj, int ntx)
{
    __shared__ uint32_t slab_all[LW * LDS_SLAB_WORDS];
    __shared__ uint32_t hacc[LDS_BAND];
    const int a = list ? list[blockIdx.z] : (int)blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t *slab = slab_all + wave * LDS_SLAB_WORDS;
    // blockIdx.x = scan of the launch * ntx + tile across (see sweep_generic_kernel)
    const int zs = blockIdx.x / ntx, bx = blockIdx.x - zs * ntx;
    bits += (int64_t)zs * d.rows * d.wpr;
    vproj += (int64_t)zs * d.A * d.cols;
    hproj += (int64_t)zs * d.A * d.rows;
    const int xw = bx * (LW * OMR_WAVE) + wave * OMR_WAVE;  // wave's first column
    const int x = xw + lane;
    const int y0 = blockIdx.y * LDS_BAND;
    const int y1 = min(d.rows, y0 + LDS_BAND);
    const bool wave_active = xw < d.cols;
    const bool active = x < d.cols;
    const int xc = active ? x : d.cols - 1;  // idle lanes shadow the last column: stays inside the window
    const int ad = adelta[(int64_t)a * d.cols + xc];
    const int bd = bdelta[(int64_t)a * d.cols + xc];
    const int2_t *__restrict__ s0p = xy0 + (int64_t)a * d.rows;
    const LdsTile tile = tiles[a];
    const int R = tile.rows_per_tile;
    const int pitch = tile.win_words | 1;
    const int pitch4 = pitch * 4;
    const int slab_byte = wave * LDS_SLAB_WORDS * 4;
    const unsigned long long active_mask = __ballot(active);

    for (int i = threadIdx.x; i < LDS_BAND; i += LW * OMR_WAVE) hacc[i] = 0;
    __syncthreads();

    uint32_t vacc = 0;
    if (wave_active) {
        // extreme lanes' table values (wave-uniform)
        const int ad_lo = __builtin_amdgcn_readfirstlane(ad);
        const int bd_lo = __builtin_amdgcn_readfirstlane(bd);
        const int ad_hi = __builtin_amdgcn_readlane(ad, 63);
        const int bd_hi = __builtin_amdgcn_readlane(bd, 63);
        // staging geometry: lanes form a (64 / WW) x WW grid, WW = power of two >= win_words
        int ww_log = 0;
        while ((1 << ww_log) < tile.win_words) ww_log++;
        const int sw = lane & ((1 << ww_log) - 1);
        const int sr = lane >> ww_log;
        const int rows_per_iter = OMR_WAVE >> ww_log;

        int hrow = 0, hbase = y0;  // hrow lane k = count of row hbase + k
        for (int ty = y0; ty < y1; ty += R) {
            const int tr = min(R, y1 - ty);
            const int2_t c0 = s0p[ty], c1 = s0p[ty + tr - 1];
            const int xa = min(min(c0.x + ad_lo, c0.x + ad_hi), min(c1.x + ad_lo, c1.x + ad_hi)) >> 10;
            const int xb = max(max(c0.x + ad_lo, c0.x + ad_hi), max(c1.x + ad_lo, c1.x + ad_hi)) >> 10;
            const int ya = min(min(c0.y + bd_lo, c0.y + bd_hi), min(c1.y + bd_lo, c1.y + bd_hi)) >> 10;
            const int yb = max(max(c0.y + bd_lo, c0.y + bd_hi), max(c1.y + bd_lo, c1.y + bd_hi)) >> 10;
            const int wx0 = xa >> 5;  // first window word (floor, may be negative)
            const int nwords = (xb >> 5) - wx0 + 1;
            const int nrows = yb - ya + 1;
            // Host sizing (SweepTables::create) makes the window fit; if a candidate ever exceeded
            // it, this tile takes the bounds-checked global-memory gather instead (same results).
            const bool fits = nwords <= tile.win_words && nrows <= tile.win_rows;  // wave-uniform
            if (fits) {
                // stage window rows [ya, ya + nrows) x words [wx0, wx0 + win_words)
                for (int r = sr; r < nrows; r += rows_per_iter) {
                    const int gy = ya + r, gw = wx0 + sw;
                    uint32_t v = 0;
                    if (sw < tile.win_words) {
                        if ((unsigned)gy < (unsigned)d.rows && (unsigned)gw < (unsigned)d.wpr)
                            v = bits[(int64_t)gy * d.wpr + gw];
                        slab[r * pitch + sw] = v;
                    }
                }
            }
            // wave-private slab: the LDS operations of one wave execute in order, no barrier needed
            auto flush_rows = [&](int y_last) {
                if (y_last - hbase == OMR_WAVE - 1) {
                    if (hrow) atomicAdd(&hacc[hbase - y0 + lane], (uint32_t)hrow);
                    hrow = 0;
                    hbase += OMR_WAVE;
                }
            };
            if (fits) {
                const int xoff = wx0 << 15;  // (wx0 * 32) << 10
                const int yoff = ya << 10;
                const int adl = ad - xoff, bdl = bd - yoff;  // window-local lane constants
                // four gathers are in flight before the first is consumed, as long as the group of four
                // rows stays inside the current 64-row block of hrow (R need not be a multiple of 4: the
                // tile sizes 6, 3, 2, 1 of strongly magnified maps start tiles anywhere); the rows in
                // between go one at a time
                for (int r = 0; r < tr;) {
                    const int y = ty + r;
                    if (r + 4 <= tr && y - hbase + 3 < OMR_WAVE) {
                        int2_t s4[4];
                        uint32_t w[4];
                        int sh[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) s4[j] = s0p[y + j];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int sx = s4[j].x + adl;  // >= 0 inside the window
                            const int sy = s4[j].y + bdl;
                            // byte address = Y * pitch4 + 4 * (X >> 5) + slab base (and_or + mad24)
                            const int off = __mul24(sy >> 10, pitch4) + (((sx >> 13) & 0xFFC) | slab_byte);
                            w[j] = *(const uint32_t *)((const char *)slab_all + off);
                            sh[j] = sx >> 10;
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t bit = __builtin_amdgcn_ubfe(w[j], (uint32_t)sh[j], 1u);
                            vacc += bit;
                            const unsigned long long m = __ballot(bit != 0) & active_mask;
                            hrow = write_lane(hrow, __popcll(m), y + j - hbase);
                        }
                        flush_rows(y + 3);
                        r += 4;
                    } else {
                        const int2_t s0 = s0p[y];
                        const int sx = s0.x + adl, sy = s0.y + bdl;
                        const uint32_t bit =
                            __builtin_amdgcn_ubfe(slab[__mul24(sy >> 10, pitch) + (sx >> 15)], (uint32_t)(sx >> 10), 1u);
                        vacc += bit;
                        const unsigned long long m = __ballot(bit != 0) & active_mask;
                        hrow = write_lane(hrow, __popcll(m), y - hbase);
                        flush_rows(y);
                        r++;
                    }
                }
            } else {
                for (int r = 0; r < tr; r++) {
                    const int y = ty + r;
                    const int2_t s0 = s0p[y];
                    const int X = (s0.x + ad) >> 10;
                    const int Y = (s0.y + bd) >> 10;
                    uint32_t bit = 0;
                    if ((unsigned)X < (unsigned)d.cols && (unsigned)Y < (unsigned)d.rows)
                        bit = (bits[(int64_t)Y * d.wpr + (X >> 5)] >> (X & 31)) & 1u;
                    vacc += bit;
                    const unsigned long long m = __ballot(bit != 0) & active_mask;
                    hrow = write_lane(hrow, __popcll(m), y - hbase);
                    flush_rows(y);
                }
            }
        }
        if (hbase < y1 && lane < y1 - hbase && hrow) atomicAdd(&hacc[hbase - y0 + lane], (uint32_t)hrow);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < y1 - y0; i += LW * OMR_WAVE) {
        const uint32_t v = hacc[i];
        if (v) atomicAdd(&hproj[(int64_t)a * d.rows + y0 + i], v);
    }
    if (active && vacc) atomicAdd(&vproj[(int64_t)a * d.cols + x], vacc);
}

hipError_t launch_sweep_lds(const uint32_t *d_bits, SweepDims d, const int32_t *d_adelta, const int32_t *d_bdelta,
                            const int2_t *d_xy0, const LdsTile *d_tiles, const int32_t *d_list, int n_list,
                            uint32_t *d_vproj, uint32_t *d_hproj, hipStream_t s, int scans)
{
    const int nz = d_list ? n_list : d.A;
    if (nz <= 0 || scans <= 0) return hipSuccess;
    if (d.cols <= 4 * OMR_WAVE) {
        dim3 grid(scans, (d.rows + LDS_BAND - 1) / LDS_BAND, nz);
        hipLaunchKernelGGL(sweep_lds_kernel<4>, grid, dim3(4 * OMR_WAVE), 0, s, d_bits, d, d_adelta, d_bdelta, d_xy0, d_tiles, d_list,
                           d_vproj, d_hproj, 1);
        return hipGetLastError();
    }
    const int ntx = (d.cols + LDS_WAVES * OMR_WAVE - 1) / (LDS_WAVES * OMR_WAVE);
    dim3 grid(ntx * scans, (d.rows + LDS_BAND - 1) / LDS_BAND, nz);
    hipLaunchKernelGGL(sweep_lds_kernel<LDS_WAVES>, grid, dim3(LDS_WAVES * OMR_WAVE), 0, s, d_bits, d, d_adelta, d_bdelta, d_xy0,
                       d_tiles, d_list, d_vproj, d_hproj, ntx);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Std-dev of the integer projections, bit-exact with calculate.rs:13-23:
//   mean = (v[0] + v[1] + ...) / n      -- the counts are small integers, every partial sum is
//                                          exact in f64, so the integer total converted once is
//                                          the same number as the reference's sequential f64 sum
//   sum  = (v[0]-mean)^2; sum = sum + (v[i]-mean)^2 for i = 1..n-1   -- STRICTLY sequential
//   sd   = sqrt(sum / n)                -- IEEE f64 divide and sqrt (correctly rounded on gfx950)
// Latency form, one block per (candidate, axis): all threads square the deviations of a chunk into
// LDS, then thread 0 folds the chunk in index order (a tree reduction would round differently).
// Used when a single scan is waited for (plan API): 40 us instead of the 170 us of the lane form.
#define SD_THREADS 256
#define SD_CHUNK 512

__global__ __launch_bounds__(SD_THREADS) void stddev_kernel(const uint32_t *__restrict__ vproj,
                                                            const uint32_t *__restrict__ hproj, SweepDims d,
                                                            double *__restrict__ v_sd, double *__restrict__ h_sd)
{
    __shared__ double sq[SD_CHUNK];
    __shared__ unsigned long long part[SD_THREADS / OMR_WAVE];
    __shared__ double mean_s;
    const int a = blockIdx.x >> 1;
    const int axis = blockIdx.x & 1;  // 0: vertical projection (per column), 1: horizontal (per row)
    const int n = axis ? d.rows : d.cols;
    const uint32_t *__restrict__ p = axis ? hproj + (int64_t)a * d.rows : vproj + (int64_t)a * d.cols;

    // the integer total: every partial sum of the reference's sequential f64 loop is exact, so the
    // total converted once is the same number
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += SD_THREADS) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < SD_THREADS / OMR_WAVE; w++) t += part[w];
        mean_s = (double)t / (double)n;
    }
    __syncthreads();
    const double mean = mean_s;
    double acc = 0.0;  // 0.0 + (v0-mean)^2 == (v0-mean)^2 exactly
    for (int base = 0; base < n; base += SD_CHUNK) {
        const int m = min(SD_CHUNK, n - base);
        for (int i = threadIdx.x; i < m; i += SD_THREADS) {
            const double dv = (double)p[base + i] - mean;
            sq[i] = dv * dv;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // the adds form one dependent chain (that order IS the specification); two batches of 16
            // LDS reads are kept in flight so the chain runs at f64-add latency, not LDS latency
            int i = 0;
            double t0[16], t1[16];
            if (m >= 16) {
#pragma unroll
                for (int j = 0; j < 16; j++) t0[j] = sq[j];
            }
            for (; i + 32 <= m; i += 32) {
#pragma unroll
                for (int j = 0; j < 16; j++) t1[j] = sq[i + 16 + j];
#pragma unroll
                for (int j = 0; j < 16; j++) acc = acc + t0[j];
                if (i + 48 <= m) {
#pragma unroll
                    for (int j = 0; j < 16; j++) t0[j] = sq[i + 32 + j];
                }
#pragma unroll
                for (int j = 0; j < 16; j++) acc = acc + t1[j];
            }
            if (i + 16 <= m) {  // t0 holds sq[i .. i+15]
#pragma unroll
                for (int j = 0; j < 16; j++) acc = acc + t0[j];
                i += 16;
            }
            for (; i < m; i++) acc = acc + sq[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sd = sqrt(acc / (double)n);
        if (axis) h_sd[a] = sd;
        else v_sd[a] = sd;
    }
}

// Throughput form.  One LANE per (candidate, axis) chain: 2A chains = 2A / 64 single-wave
// blocks.  The chains cannot be shortened (the add order is the specification), but 64 of them can
// run side by side in one wave, every lane streaming its own row 64 elements ahead of its add
// chain.  A sweep of 400 candidates needs 13 waves instead of 800 four-wave blocks, so the kernel no
// longer pushes blocks of the next scan's sweep off their CUs while it waits on its add chain (it
// runs on the post stream, concurrently with that sweep).
#define SD_B 32  // elements per batch; three batches are in flight (the kernel runs beside the sweep,
                 // whose table traffic owns the L1: a row load is an L2 round trip, about 1 us)
struct SdBatch {
    uint32_t v[SD_B];
};
__device__ __forceinline__ SdBatch sd_load(const uint32_t *__restrict__ p, int i, int n, bool vec)
{
    SdBatch b;
    if (vec && i + SD_B <= n) {
        const uint4 *q = (const uint4 *)(p + i);
#pragma unroll
        for (int k = 0; k < SD_B / 4; k++) {
            const uint4 t = q[k];
            b.v[4 * k] = t.x, b.v[4 * k + 1] = t.y, b.v[4 * k + 2] = t.z, b.v[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SD_B; k++) b.v[k] = i + k < n ? p[i + k] : 0u;
    }
    return b;
}

__global__ __launch_bounds__(64) void stddev_lanes_kernel(const uint32_t *__restrict__ vproj,
                                                          const uint32_t *__restrict__ hproj, SweepDims d,
                                                          double *__restrict__ v_sd, double *__restrict__ h_sd,
                                                          int scans)
{
    // 13 waves of pure latency chain sharing SIMDs with the sweep's VALU-bound waves: ask the
    // instruction arbiter to serve this wave first (it issues one dependent op at a time anyway)
    __builtin_amdgcn_s_setprio(3);
    // per scan: chains 0..A-1 are the vertical projections, A..2A-1 the horizontal ones
    const int gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= 2 * d.A * scans) return;
    const int scan = gid / (2 * d.A), id = gid - scan * 2 * d.A;
    vproj += (int64_t)scan * d.A * d.cols;
    hproj += (int64_t)scan * d.A * d.rows;
    v_sd += (int64_t)scan * d.A;
    h_sd += (int64_t)scan * d.A;
    const bool vert = id < d.A;
    const int n = vert ? d.cols : d.rows;
    const uint32_t *__restrict__ p = vert ? vproj + (int64_t)id * d.cols : hproj + (int64_t)(id - d.A) * d.rows;
    const bool vec = (((uintptr_t)p) & 15u) == 0;
    unsigned long long s = 0;  // the integer total (exact; see above)
    {
        SdBatch cur = sd_load(p, 0, n, vec), n1 = sd_load(p, SD_B, n, vec);
        for (int i = 0; i < n; i += SD_B) {
            const SdBatch n2 = sd_load(p, i + 2 * SD_B, n, vec);
#pragma unroll
            for (int k = 0; k < SD_B; k++) s += cur.v[k];  // zeros past the end
            cur = n1;
            n1 = n2;
        }
    }
    const double mean = (double)s / (double)n;
    double acc = 0.0;  // 0.0 + (v0-mean)^2 == (v0-mean)^2 exactly
    SdBatch cur = sd_load(p, 0, n, vec), n1 = sd_load(p, SD_B, n, vec);
    for (int i = 0; i < n; i += SD_B) {
        const SdBatch n2 = sd_load(p, i + 2 * SD_B, n, vec);
        if (i + SD_B <= n) {
#pragma unroll
            for (int k = 0; k < SD_B; k++) {
                const double dv = (double)cur.v[k] - mean;
                acc = acc + dv * dv;  // strictly in index order; no contraction (-ffp-contract=off)
            }
        } else {
            for (int k = 0; k < n - i; k++) {
                const double dv = (double)cur.v[k] - mean;
                acc = acc + dv * dv;
            }
        }
        cur = n1;
        n1 = n2;
    }
    const double sd = sqrt(acc / (double)n);
    if (vert) v_sd[id] = sd;
    else h_sd[id - d.A] = sd;
}

hipError_t launch_stddev(const uint32_t *d_vproj, const uint32_t *d_hproj, SweepDims d, double *d_v_sd,
                         double *d_h_sd, hipStream_t s, int scans, bool latency)
{
    if (d.A <= 0 || scans <= 0) return hipSuccess;
    if (latency && scans == 1) {
        hipLaunchKernelGGL(stddev_kernel, dim3(2 * d.A), dim3(SD_THREADS), 0, s, d_vproj, d_hproj, d, d_v_sd, d_h_sd);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(stddev_lanes_kernel, dim3((2 * d.A * scans + 63) / 64), dim3(64), 0, s, d_vproj, d_hproj, d,
                       d_v_sd, d_h_sd, scans);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// projection.rs:125-190.  Max sets of v_sd and h_sd ("possibles"; the reference seeds each list
// with index 0 and re-visits it, so index 0 counts twice while it holds the maximum -- that only
// matters for the len()==1 test); unique-and-equal -> that index; otherwise the candidate of
// the union with the largest v^2+h^2 (strict < from 0.0); the reference iterates a HashMap, we
// take the lowest index among exact ties (quirk B5); nothing positive -> n/2.
#define AM_THREADS 256

// 64-bit key helpers: (value, index) packed so that max() prefers the larger value and, on
// equal values, the LOWER index.  The scores are finite and >= 0, so their IEEE bit patterns
// order like the values.
__device__ __forceinline__ unsigned long long am_block_max(unsigned long long v, unsigned long long *sh)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(v, off);
        v = o > v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = sh[0];
    for (int w = 1; w < AM_THREADS / OMR_WAVE; w++) r = sh[w] > r ? sh[w] : r;
    return r;
}

__device__ __forceinline__ unsigned am_block_sum(unsigned v, unsigned *sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned r = 0;
    for (int w = 0; w < AM_THREADS / OMR_WAVE; w++) r += sh[w];
    return r;
}

__global__ __launch_bounds__(AM_THREADS) void argmax_path1_kernel(const double *__restrict__ v,
                                                                  const double *__restrict__ h, int n,
                                                                  int32_t *__restrict__ best)
{
    __shared__ unsigned long long shm[AM_THREADS / OMR_WAVE];
    __shared__ unsigned shc[AM_THREADS / OMR_WAVE];
    v += (int64_t)blockIdx.x * n;  // blockIdx.x = scan of the launch
    h += (int64_t)blockIdx.x * n;
    best += blockIdx.x;
    // pass 1: maxima of both score vectors
    double lv = 0.0, lh = 0.0;  // scores are >= 0
    for (int i = threadIdx.x; i < n; i += AM_THREADS) {
        lv = fmax(lv, v[i]);
        lh = fmax(lh, h[i]);
    }
    const double vmax = __longlong_as_double((long long)am_block_max((unsigned long long)__double_as_longlong(lv), shm));
    const double hmax = __longlong_as_double((long long)am_block_max((unsigned long long)__double_as_longlong(lh), shm));
    // pass 2: sizes of the two "possibles" lists, their first members, and the best candidate
    unsigned vc = 0, hc = 0;
    unsigned vfirst = 0xffffffffu, hfirst = 0xffffffffu;
    unsigned long long key = 0;  // (v^2+h^2 bits, ~index): 0 = "nothing beats 0.0"
    for (int i = threadIdx.x; i < n; i += AM_THREADS) {
        const double vi = v[i], hi = h[i];
        const bool ve = vi == vmax, he = hi == hmax;
        if (ve) {
            vc++;
            vfirst = min(vfirst, (unsigned)i);
        }
        if (he) {
            hc++;
            hfirst = min(hfirst, (unsigned)i);
        }
        if (ve || he) {
            const double cur = vi * vi + hi * hi;
            if (cur > 0.0) {
                // cur <= 2^53-ish: its top bits fit 40 bits?  No -- keep the full pattern and break
                // ties with a second reduction instead (see below)
                const unsigned long long k = (unsigned long long)__double_as_longlong(cur);
                key = k > key ? k : key;
            }
        }
    }
    const unsigned vcount = am_block_sum(vc, shc) + (v[0] == vmax ? 1u : 0u);  // index 0 is listed twice
    const unsigned hcount = am_block_sum(hc, shc) + (h[0] == hmax ? 1u : 0u);
    const unsigned vf = 0xffffffffu - (unsigned)am_block_max(0xffffffffu - vfirst, shm);
    const unsigned hf = 0xffffffffu - (unsigned)am_block_max(0xffffffffu - hfirst, shm);
    const unsigned long long kbest = am_block_max(key, shm);
    // pass 3: lowest index among the candidates that reach the best v^2+h^2
    unsigned cand = 0xffffffffu;
    if (kbest != 0) {
        for (int i = threadIdx.x; i < n; i += AM_THREADS) {
            const double vi = v[i], hi = h[i];
            if (vi == vmax || hi == hmax) {
                const double cur = vi * vi + hi * hi;
                if ((unsigned long long)__double_as_longlong(cur) == kbest) cand = min(cand, (unsigned)i);
            }
        }
    }
    const unsigned cf = 0xffffffffu - (unsigned)am_block_max(0xffffffffu - cand, shm);
    if (threadIdx.x == 0) {
        int result;
        if (vcount == 1 && hcount == 1 && vf == hf) result = (int)vf;
        else if (kbest != 0) result = (int)cf;
        else result = n / 2;
        *best = result;
    }
}

hipError_t launch_argmax_path1(const double *d_v_sd, const double *d_h_sd, int A, int32_t *d_best, hipStream_t s,
                               int scans)
{
    if (A <= 0 || scans <= 0) return hipSuccess;
    hipLaunchKernelGGL(argmax_path1_kernel, dim3(scans), dim3(AM_THREADS), 0, s, d_v_sd, d_h_sd, A, d_best);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Per-image helpers (callers and data formats either side of the sweep).

// transfer.rs:294-301 / omr.rs:129-139: threshold(127, 255, THRESH_BINARY)
__global__ __launch_bounds__(256) void threshold_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                        int cols, uint8_t *__restrict__ dst, int64_t dstep,
                                                        int thresh, int maxval)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < cols) dst[(int64_t)y * dstep + x] = (int)src[(int64_t)y * sstep + x] > thresh ? (uint8_t)maxval : 0;
}

hipError_t launch_threshold(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                            int thresh, int maxval, hipStream_t s)
{
    hipLaunchKernelGGL(threshold_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, s, d_src, sstep, rows, cols,
                       d_dst, dstep, thresh, maxval);
    return hipGetLastError();
}

// transfer.rs:283-290 / omr.rs:88-92: cvtColor(COLOR_RGB2GRAY) 8U,
// (c0*9798 + c1*19235 + c2*3735 + 16384) >> 15 in memory order (quirk B8 kept).
__global__ __launch_bounds__(256) void rgb2gray_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                       int cols, int cn, uint8_t *__restrict__ dst, int64_t dstep)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < cols) {
        const uint8_t *S = src + (int64_t)y * sstep + (int64_t)x * cn;
        dst[(int64_t)y * dstep + x] = (uint8_t)((S[0] * 9798 + S[1] * 19235 + S[2] * 3735 + (1 << 14)) >> 15);
    }
}

hipError_t launch_rgb2gray(const uint8_t *d_src, int64_t sstep, int rows, int cols, int cn, uint8_t *d_dst,
                           int64_t dstep, hipStream_t s)
{
    hipLaunchKernelGGL(rgb2gray_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, s, d_src, sstep, rows, cols, cn,
                       d_dst, dstep);
    return hipGetLastError();
}

// omr.rs:98-112: one pass of erode with the 3x3 "ellipse" (= cross) element, border = +inf.
__global__ __launch_bounds__(256) void erode_cross3_kernel(const uint8_t *__restrict__ src, int64_t sstep, int rows,
                                                           int cols, uint8_t *__restrict__ dst, int64_t dstep)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= cols) return;
    const uint8_t *S = src + (int64_t)y * sstep + x;
    int m = S[0];
    if (y > 0) m = min(m, (int)S[-sstep]);
    if (y + 1 < rows) m = min(m, (int)S[sstep]);
    if (x > 0) m = min(m, (int)S[-1]);
    if (x + 1 < cols) m = min(m, (int)S[1]);
    dst[(int64_t)y * dstep + x] = (uint8_t)m;
}

hipError_t launch_erode_cross3(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                               hipStream_t s)
{
    hipLaunchKernelGGL(erode_cross3_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, s, d_src, sstep, rows, cols,
                       d_dst, dstep);
    return hipGetLastError();
}

__device__ inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// resize INTER_AREA, integer factors (OpenCV resizeAreaFast_): transfer.rs:66-91, omr.rs:114-126.
// One thread per destination byte (x runs over dcols*cn).
__global__ __launch_bounds__(256) void resize_area_int_kernel(const uint8_t *__restrict__ src, int64_t sstep,
                                                              int srows, int scols, int cn,
                                                              uint8_t *__restrict__ dst, int64_t dstep, int drows,
                                                              int dcols, int kx, int ky)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    const int dwidth = dcols * cn, swidth = scols * cn;
    if (dx >= dwidth) return;
    const int sy0 = dy * ky;
    uint8_t out;
    if (sy0 >= srows) {
        out = 0;
    } else {
        const int dwidth1 = (scols / kx) * cn;
        const int w = sy0 + ky <= srows ? dwidth1 : 0;
        const int sx0 = kx * (dx / cn) * cn + dx % cn;
        if (dx < w) {
            int sum = 0;
            for (int sy = 0; sy < ky; sy++)
                for (int sx = 0; sx < kx; sx++) sum += src[(int64_t)(sy0 + sy) * sstep + sx0 + sx * cn];
            if (kx == 2 && ky == 2) out = (uint8_t)((sum + 2) >> 2);
            else out = sat_u8((int)rintf((float)sum * (1.f / (float)(kx * ky))));
        } else {
            int sum = 0, count = 0;
            for (int sy = 0; sy < ky; sy++) {
                if (sy0 + sy >= srows) break;
                for (int sx = 0; sx < kx * cn; sx += cn) {
                    if (sx0 + sx >= swidth) break;
                    sum += src[(int64_t)(sy0 + sy) * sstep + sx0 + sx];
                    count++;
                }
            }
            out = count ? sat_u8((int)rintf((float)sum / (float)count)) : 0;
        }
    }
    dst[(int64_t)dy * dstep + dx] = out;
}

hipError_t launch_resize_area_int(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                                  int64_t dstep, int drows, int dcols, int kx, int ky, hipStream_t s)
{
    hipLaunchKernelGGL(resize_area_int_kernel, dim3((dcols * cn + 255) / 256, drows), dim3(256), 0, s, d_src, sstep,
                       srows, scols, cn, d_dst, dstep, drows, dcols, kx, ky);
    return hipGetLastError();
}

// resize INTER_AREA, general shrink (OpenCV resizeArea_<uchar,float>): per destination byte the
// same float accumulation order as ResizeArea_Invoker: for each source row tap (ascending) the
// row sum buf = sum_k S*alpha_k (ascending k), then sum (+)= beta*buf.  xofs/yofs: CSR offsets
// of the taps of every destination column / row.
__global__ __launch_bounds__(256) void resize_area_general_kernel(const uint8_t *__restrict__ src, int64_t sstep,
                                                                  int cn, uint8_t *__restrict__ dst, int64_t dstep,
                                                                  int drows, int dcols,
                                                                  const AreaTap *__restrict__ xtab,
                                                                  const int32_t *__restrict__ xofs,
                                                                  const AreaTap *__restrict__ ytab,
                                                                  const int32_t *__restrict__ yofs)
{
    const int dxb = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dxb >= dcols * cn) return;
    const int dx = dxb / cn, c = dxb % cn;
    float sum = 0.f;
    bool first = true;
    for (int j = yofs[dy]; j < yofs[dy + 1]; j++) {
        const float beta = ytab[j].alpha;
        const uint8_t *S = src + (int64_t)ytab[j].si * sstep + c;
        float buf = 0.f;
        for (int k = xofs[dx]; k < xofs[dx + 1]; k++) buf += (float)S[xtab[k].si] * xtab[k].alpha;
        if (first) {
            sum = beta * buf;  // ResizeArea_Invoker assigns on the first tap of a destination row
            first = false;
        } else {
            sum += beta * buf;
        }
    }
    dst[(int64_t)dy * dstep + dxb] = sat_u8((int)rintf(sum));
}

hipError_t launch_resize_area_general(const uint8_t *d_src, int64_t sstep, int cn, uint8_t *d_dst, int64_t dstep,
                                      int drows, int dcols, const AreaTap *d_xtab, const int32_t *d_xofs,
                                      const AreaTap *d_ytab, const int32_t *d_yofs, hipStream_t s)
{
    hipLaunchKernelGGL(resize_area_general_kernel, dim3((dcols * cn + 255) / 256, drows), dim3(256), 0, s, d_src,
                       sstep, cn, d_dst, dstep, drows, dcols, d_xtab, d_xofs, d_ytab, d_yofs);
    return hipGetLastError();
}

// warpAffine INTER_NEAREST on a cn-channel u8 image (transfer.rs:477-485, omr.rs:435-443): the
// final deskew of correct_default and rotate_mat's materialising form.  Tables are evaluated
// in place (same f64 expressions, no contraction).
__global__ __launch_bounds__(256) void warp_nn_kernel(const uint8_t *__restrict__ src, int64_t sstep, int srows,
                                                      int scols, int cn, uint8_t *__restrict__ dst, int64_t dstep,
                                                      int drows, int dcols, const double *__restrict__ M,
                                                      uint32_t border)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dcols) return;
    const int adelta = (int)rint(M[0] * (double)x * 1024.0);
    const int bdelta = (int)rint(M[3] * (double)x * 1024.0);
    const int X0 = (int)rint((M[1] * (double)y + M[2]) * 1024.0) + 512;
    const int Y0 = (int)rint((M[4] * (double)y + M[5]) * 1024.0) + 512;
    int X = (X0 + adelta) >> 10, Y = (Y0 + bdelta) >> 10;
    X = max(-32768, min(32767, X));
    Y = max(-32768, min(32767, Y));
    uint8_t *D = dst + (int64_t)y * dstep + (int64_t)x * cn;
    if ((unsigned)X < (unsigned)scols && (unsigned)Y < (unsigned)srows) {
        const uint8_t *S = src + (int64_t)Y * sstep + (int64_t)X * cn;
        for (int k = 0; k < cn; k++) D[k] = S[k];
    } else {
        for (int k = 0; k < cn; k++) D[k] = (uint8_t)(border >> (8 * k));
    }
}

hipError_t launch_warp_nn(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                          int64_t dstep, int drows, int dcols, const double *d_Minv, uint32_t border_rgba,
                          hipStream_t s)
{
    hipLaunchKernelGGL(warp_nn_kernel, dim3((dcols + 255) / 256, drows), dim3(256), 0, s, d_src, sstep, srows, scols,
                       cn, d_dst, dstep, drows, dcols, d_Minv, border_rgba);
    return hipGetLastError();
}

// warpAffine INTER_LINEAR (core/src/main.rs:72-81, app test.rs:322-331 through rotate_mat
// CONTAIN): 5 fractional bits, 15-bit weights (32-fy)(32-fx)*32 ..., (v + 16384) >> 15.
__global__ __launch_bounds__(256) void warp_linear_kernel(const uint8_t *__restrict__ src, int64_t sstep, int srows,
                                                          int scols, int cn, uint8_t *__restrict__ dst,
                                                          int64_t dstep, int drows, int dcols,
                                                          const double *__restrict__ M, uint32_t border)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dcols) return;
    const int adelta = (int)rint(M[0] * (double)x * 1024.0);
    const int bdelta = (int)rint(M[3] * (double)x * 1024.0);
    const int X0 = (int)rint((M[1] * (double)y + M[2]) * 1024.0) + 16;
    const int Y0 = (int)rint((M[4] * (double)y + M[5]) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
    const int fx = X & 31, fy = Y & 31;
    const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    uint8_t *D = dst + (int64_t)y * dstep + (int64_t)x * cn;
    if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) {
        for (int k = 0; k < cn; k++) D[k] = (uint8_t)(border >> (8 * k));
        return;
    }
    const bool in_x0 = sx >= 0 && sx < scols, in_x1 = sx + 1 >= 0 && sx + 1 < scols;
    const bool in_y0 = sy >= 0 && sy < srows, in_y1 = sy + 1 >= 0 && sy + 1 < srows;
    for (int k = 0; k < cn; k++) {
        const int b = (int)((border >> (8 * k)) & 255u);
        const int v0 = in_x0 && in_y0 ? src[(int64_t)sy * sstep + (int64_t)sx * cn + k] : b;
        const int v1 = in_x1 && in_y0 ? src[(int64_t)sy * sstep + (int64_t)(sx + 1) * cn + k] : b;
        const int v2 = in_x0 && in_y1 ? src[(int64_t)(sy + 1) * sstep + (int64_t)sx * cn + k] : b;
        const int v3 = in_x1 && in_y1 ? src[(int64_t)(sy + 1) * sstep + (int64_t)(sx + 1) * cn + k] : b;
        D[k] = sat_u8((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
    }
}

hipError_t launch_warp_linear(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                              int64_t dstep, int drows, int dcols, const double *d_Minv, uint32_t border_rgba,
                              hipStream_t s)
{
    hipLaunchKernelGGL(warp_linear_kernel, dim3((dcols + 255) / 256, drows), dim3(256), 0, s, d_src, sstep, srows,
                       scols, cn, d_dst, dstep, drows, dcols, d_Minv, border_rgba);
    return hipGetLastError();
}

}  // namespace omr
