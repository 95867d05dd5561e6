// hough.hip -- Canny + progressive probabilistic Hough (OpenCV 4.6.0 semantics) for gfx950.
//
// Canny is data-parallel and exact: Sobel / non-maximum suppression are integer stencils and the
// hysteresis result is a set (the candidates 8-connected to a strong pixel), so the order of
// propagation does not matter.  HoughLinesP is a *sequential* randomised algorithm (hough.cpp
// HoughLinesProbabilistic: every drawn point sees the accumulator and the mask left by all points
// before it), so the kernel keeps the sequence, shortens the step (one serving wave per scan: lane =
// accumulator angle for the 180 votes, lane = position for the line walks; a second wave draws ahead,
// three more help with un-votes) and runs scans side by side (one workgroup each).  Same RNG (cv::RNG seed 2^64-1), same
// float32 vote arithmetic (no contraction), same 16.16 walk: the segments are bit-identical to the
// CPU restatement.
//
// Reference call sites: packages/lib/src/hough.rs:27-43, packages/lib/src/omr.rs:236-253, :323-330.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hough.hpp"

namespace omr {

// ------------------------------------------------------------------------------------------
// Canny: Sobel 3x3 (BORDER_REPLICATE) -> L1 magnitude -> sector test -> map {0, 1, 2}
#define CN_TW 64
#define CN_TH 16
#define CN_THREADS 256

template <int CN>
__global__ __launch_bounds__(CN_THREADS) void canny_nms_kernel(const uint8_t *__restrict__ src, int64_t scan_stride,
                                                               int64_t step, int rows, int cols, int low, int high,
                                                               uint8_t *__restrict__ map)
{
    constexpr int SW = CN_TW + 4, SH = CN_TH + 4;  // source tile, 2 px halo
    constexpr int MW = CN_TW + 2, MH = CN_TH + 2;  // gradient tile, 1 px halo
    __shared__ uint8_t s_src[SH][SW * CN];
    __shared__ uint16_t s_mag[MH][MW];
    __shared__ short2 s_g[MH][MW];
    const int x0 = blockIdx.x * CN_TW, y0 = blockIdx.y * CN_TH;
    src += (int64_t)blockIdx.z * scan_stride;
    map += (int64_t)blockIdx.z * rows * cols;
    for (int i = threadIdx.x; i < SH * SW; i += CN_THREADS) {
        const int ty = i / SW, tx = i - ty * SW;
        const int gy = min(max(y0 + ty - 2, 0), rows - 1), gx = min(max(x0 + tx - 2, 0), cols - 1);
        const uint8_t *p = src + (int64_t)gy * step + (int64_t)gx * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) s_src[ty][tx * CN + c] = p[c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MH * MW; i += CN_THREADS) {
        const int my = i / MW, mx = i - my * MW;
        const int y = y0 + my - 1, x = x0 + mx - 1;
        int bm = 0, bx = 0, by = 0;
        if (y >= 0 && y < rows && x >= 0 && x < cols) {
            const int ty = my + 1, tx = mx + 1;
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const int a00 = s_src[ty - 1][(tx - 1) * CN + c], a01 = s_src[ty - 1][tx * CN + c];
                const int a02 = s_src[ty - 1][(tx + 1) * CN + c], a10 = s_src[ty][(tx - 1) * CN + c];
                const int a12 = s_src[ty][(tx + 1) * CN + c], a20 = s_src[ty + 1][(tx - 1) * CN + c];
                const int a21 = s_src[ty + 1][tx * CN + c], a22 = s_src[ty + 1][(tx + 1) * CN + c];
                const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
                const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
                const int m = abs(dx) + abs(dy);
                if (c == 0 || m > bm) bm = m, bx = dx, by = dy;  // first channel with the largest magnitude
            }
        }
        s_mag[my][mx] = (uint16_t)bm;  // <= 2040
        s_g[my][mx] = make_short2((short)bx, (short)by);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CN_TH * CN_TW; i += CN_THREADS) {
        const int py = i / CN_TW, px = i - py * CN_TW;
        const int y = y0 + py, x = x0 + px;
        if (y >= rows || x >= cols) continue;
        const int my = py + 1, mx = px + 1;
        const int m = s_mag[my][mx];
        bool edge = false;
        if (m > low) {
            const int xs = s_g[my][mx].x, ys = s_g[my][mx].y;
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;
            if (ay < tg22x) {
                edge = m > s_mag[my][mx - 1] && m >= s_mag[my][mx + 1];
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    edge = m > s_mag[my - 1][mx] && m >= s_mag[my + 1][mx];
                } else {
                    const int sgn = (xs ^ ys) < 0 ? -1 : 1;
                    edge = m > s_mag[my - 1][mx - sgn] && m > s_mag[my + 1][mx + sgn];
                }
            }
        }
        map[(int64_t)y * cols + x] = edge ? (m > high ? 2 : 0) : 1;
    }
}

hipError_t launch_canny_nms(const uint8_t *d_src, int64_t scan_stride, int64_t row_step, int rows, int cols, int cn,
                            int n, int low, int high, uint8_t *d_map, hipStream_t s)
{
    const dim3 grid((cols + CN_TW - 1) / CN_TW, (rows + CN_TH - 1) / CN_TH, n);
    if (cn == 1)
        hipLaunchKernelGGL(canny_nms_kernel<1>, grid, dim3(CN_THREADS), 0, s, d_src, scan_stride, row_step, rows, cols,
                           low, high, d_map);
    else if (cn == 3)
        hipLaunchKernelGGL(canny_nms_kernel<3>, grid, dim3(CN_THREADS), 0, s, d_src, scan_stride, row_step, rows, cols,
                           low, high, d_map);
    else if (cn == 4)
        hipLaunchKernelGGL(canny_nms_kernel<4>, grid, dim3(CN_THREADS), 0, s, d_src, scan_stride, row_step, rows, cols,
                           low, high, d_map);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// Hysteresis: every tile runs to its local fixed point in LDS; tiles exchange through the map
// between launches, the host relaunches until no tile changed.
#define HY_TW 64
#define HY_TH 32
#define HY_THREADS 256

__global__ __launch_bounds__(HY_THREADS) void canny_hyst_kernel(uint8_t *__restrict__ map, int rows, int cols,
                                                                int *__restrict__ changed)
{
    __shared__ uint8_t t[HY_TH + 2][HY_TW + 2 + 2];  // +2: keep rows 4-byte friendly
    const int x0 = blockIdx.x * HY_TW, y0 = blockIdx.y * HY_TH;
    map += (int64_t)blockIdx.z * rows * cols;
    for (int i = threadIdx.x; i < (HY_TH + 2) * (HY_TW + 2); i += HY_THREADS) {
        const int ty = i / (HY_TW + 2), tx = i - ty * (HY_TW + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        t[ty][tx] = (y >= 0 && y < rows && x >= 0 && x < cols) ? map[(int64_t)y * cols + x] : 1;
    }
    __syncthreads();
    bool any = false;
    for (;;) {
        bool ch = false;
        for (int i = threadIdx.x; i < HY_TH * HY_TW; i += HY_THREADS) {
            const int py = i / HY_TW + 1, px = i % HY_TW + 1;
            if (t[py][px] == 0) {
                const bool near = t[py - 1][px - 1] == 2 || t[py - 1][px] == 2 || t[py - 1][px + 1] == 2 ||
                                  t[py][px - 1] == 2 || t[py][px + 1] == 2 || t[py + 1][px - 1] == 2 ||
                                  t[py + 1][px] == 2 || t[py + 1][px + 1] == 2;
                if (near) {
                    t[py][px] = 2;  // monotone 0 -> 2: a neighbour reading the old value just waits a round
                    ch = true;
                }
            }
        }
        if (!__syncthreads_or(ch)) break;
        any = true;
    }
    if (!any) return;
    for (int i = threadIdx.x; i < HY_TH * HY_TW; i += HY_THREADS) {
        const int py = i / HY_TW, px = i % HY_TW;
        const int y = y0 + py, x = x0 + px;
        if (y < rows && x < cols) map[(int64_t)y * cols + x] = t[py + 1][px + 1];
    }
    if (threadIdx.x == 0) atomicOr(changed, 1);
}

hipError_t launch_canny_hysteresis(uint8_t *d_map, int rows, int cols, int n, int *d_changed, hipStream_t s)
{
    const dim3 grid((cols + HY_TW - 1) / HY_TW, (rows + HY_TH - 1) / HY_TH, n);
    hipLaunchKernelGGL(canny_hyst_kernel, grid, dim3(HY_THREADS), 0, s, d_map, rows, cols, d_changed);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// raster-order point list (hough.cpp stage 1) and the tiled mask
__host__ __device__ __forceinline__ int64_t mask_word(int y, int x, int tiles_x)
{
    return (int64_t)(y >> 3) * tiles_x + (x >> 3);
}
__host__ __device__ __forceinline__ int mask_bit(int y, int x) { return (y & 7) * 8 + (x & 7); }

__device__ __forceinline__ int block_sum_256(int v, int *sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void edges_rowcount_kernel(uint8_t *__restrict__ img, int rows, int cols,
                                                             int from_map, int32_t *__restrict__ rowcnt)
{
    __shared__ int sh[4];
    const int y = blockIdx.x, scan = blockIdx.y;
    uint8_t *p = img + ((int64_t)scan * rows + y) * cols;
    int c = 0;
    for (int x = threadIdx.x; x < cols; x += 256) {
        uint8_t v = p[x];
        if (from_map) {
            v = v == 2 ? 255 : 0;
            p[x] = v;
        }
        c += v != 0;
    }
    c = block_sum_256(c, sh);
    if (threadIdx.x == 0) rowcnt[(int64_t)scan * rows + y] = c;
}

hipError_t launch_edges_rowcount(uint8_t *d_map, int rows, int cols, int n, int from_map, int32_t *d_rowcnt,
                                 hipStream_t s)
{
    hipLaunchKernelGGL(edges_rowcount_kernel, dim3(rows, n), dim3(256), 0, s, d_map, rows, cols, from_map, d_rowcnt);
    return hipGetLastError();
}

__global__ __launch_bounds__(1024) void edges_rowscan_kernel(const int32_t *__restrict__ rowcnt, int rows,
                                                             int32_t *__restrict__ rowoff, int32_t *__restrict__ total)
{
    __shared__ int sh[1024];
    const int scan = blockIdx.x, tid = threadIdx.x;
    rowcnt += (int64_t)scan * rows;
    rowoff += (int64_t)scan * rows;
    const int per = (rows + 1023) / 1024;
    const int r0 = min(tid * per, rows), r1 = min(r0 + per, rows);
    int s = 0;
    for (int r = r0; r < r1; r++) s += rowcnt[r];
    sh[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan of the 1024 partials
        const int v = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    int run = sh[tid] - s;  // exclusive
    for (int r = r0; r < r1; r++) {
        rowoff[r] = run;
        run += rowcnt[r];
    }
    if (tid == 1023) total[scan] = sh[1023];
}

hipError_t launch_edges_rowscan(const int32_t *d_rowcnt, int rows, int n, int32_t *d_rowoff, int32_t *d_total,
                                hipStream_t s)
{
    hipLaunchKernelGGL(edges_rowscan_kernel, dim3(n), dim3(1024), 0, s, d_rowcnt, rows, d_rowoff, d_total);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void edges_compact_kernel(const uint8_t *__restrict__ img, int rows, int cols,
                                                            const int32_t *__restrict__ rowoff,
                                                            const int64_t *__restrict__ scan_off,
                                                            uint32_t *__restrict__ nz, uint8_t *__restrict__ tiled)
{
    __shared__ int sh[4];
    const int y = blockIdx.x, scan = blockIdx.y;
    const uint8_t *p = img + ((int64_t)scan * rows + y) * cols;
    uint32_t *out = nz + scan_off[scan] + rowoff[(int64_t)scan * rows + y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int xb = 0; xb < cols; xb += 256) {
        const int x = xb + threadIdx.x;
        const bool f = x < cols && p[x] != 0;
        const unsigned long long m = __ballot(f);
        if ((lane & 7) == 0 && x < cols) {  // this lane's 8 pixels = one row of a tile
            const unsigned long long byte = (m >> lane) & 0xffull;
            if (byte)
                atomicOr((unsigned long long *)tiled + (ppht_mask_bytes(rows, cols) / 8) * scan +
                             mask_word(y, x, ppht_tiles_x(cols)),
                         byte << ((y & 7) * 8));
        }
        __syncthreads();
        if (lane == 0) sh[wave] = __popcll(m);
        __syncthreads();
        int before = __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) before += sh[w];
        if (f) out[base + before] = ((uint32_t)y << 16) | (uint32_t)x;
        base += sh[0] + sh[1] + sh[2] + sh[3];
    }
}

hipError_t launch_edges_compact(const uint8_t *d_edges, int rows, int cols, int n, const int32_t *d_rowoff,
                                const int64_t *d_scan_off, uint32_t *d_nz, uint8_t *d_mask_tiled, hipStream_t s)
{
    hipLaunchKernelGGL(edges_compact_kernel, dim3(rows, n), dim3(256), 0, s, d_edges, rows, cols, d_rowoff, d_scan_off,
                       d_nz, d_mask_tiled);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// HoughLinesProbabilistic, one workgroup of FIVE waves per scan and no workgroup barrier after the start.
//
// The algorithm is a dependent chain over the drawn points (every point sees the accumulator and the mask left
// by all points before it), so a scan's time is (points served) x (latency of one step).  The step is cut down
// to two memory round trips -- the vote's accumulator reads and the walk's mask reads -- on ONE serving wave:
//
//   wave 1, the draw stage, runs AHEAD of the rest: the draw order does not depend on the image (cv::RNG and
//     swap-remove on the point list), so this wave turns the list into the sequence of drawn points, 64 draws
//     per round -- lane 0 steps the RNG, every lane takes one draw (index, its slot, the slot swapped in), the
//     64 sequential swap-removes are replayed on those registers (v_readlane of draw t, compare, select) and
//     written back -- and publishes its progress in LDS (release / acquire at workgroup scope: the waves of a
//     workgroup run on one CU and share its L1).
//   wave 0 takes 64 drawn points at a time, tests them against the mask in one round trip and serves those
//     still set, lowest first:
//       vote     lane = accumulator angle (three per lane): load, +1, store; arg-max by DPP (larger value,
//                then the LOWER angle: "if (max_val < val)" keeps the first maximum).  The bins of the NEXT
//                pending point are read ahead while this point walks;
//       walk     lane = step: 128 steps of both directions per round (four mask words per lane in flight); the
//                sequential gap rule without a loop, on the ballots;
//       pass 2   erases the segment's points (atomic AND, nothing waits) from the flags pass 1 already holds;
//                the points of an ACCEPTED segment are collected in LDS and un-voted lane = point, one angle
//                after the other, the angles split between wave 0 and
//   waves 2-4 (a no-return atomic costs a wave >= 72 cycles to issue whatever its lanes do);
//       pending points are re-tested WITHOUT memory: a point that was set is erased by the walk exactly when it
//                is one of the walked positions up to the segment's end -- integer arithmetic on registers.
//
// The accumulator rows are stored compactly (row n holds only the rho range an image of this size can reach:
// 2.8 MB instead of 8.6 MB at A4, so that it stays in the XCD's L2 next to the 1.1 MB mask).  What is left per
// served point is mostly the CU's memory pipeline (one address per cycle, shared by all five waves: 180 bins read
// and written, the mask words, the erasing and un-voting atomics) and the serving wave's own instruction stream.
struct alignas(16) PphtShared {
    float4 ang[OMR_PPHT_MAX_ANGLES];     // (cos / rho, sin / rho, byte offset of the row's bin rho = 0 [bits], -)
    PphtWalk walk[OMR_PPHT_MAX_ANGLES];  // 16 bytes each
    uint32_t pts[1024];                  // points of an accepted segment waiting for their un-votes (y << 16 | x)
    uint32_t r[64];                      // draw round: raw RNG outputs
    int produced;                        // draws whose points are in order[]
    int job_seq, job_npts, job_done;     // un-vote jobs: wave 0 -> the helper waves and back
};

#define PP_WG __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ uint32_t list_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, PP_WG); }
__device__ __forceinline__ void list_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, PP_WG); }

// maximum over the 64 lanes (all active), returned uniform
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

#ifdef OMR_RUNS_DEBUG
// debug build only: wave 0 / lane 0 phase clocks of scan 0 [draw, vote+argmax, walk pass 1, pass 2 + un-vote,
// re-test, served points, pass-1 rounds, pass-2 rounds, un-voted points, accepted segments, un-vote clocks, -];
// read with omr_debug_ppht_stamps()
__device__ unsigned long long g_ppht_stamps[12];
#define PP_CLK(V)                                                                  \
    unsigned long long V = 0;                                                      \
    if (lane == 0 && scan == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V)::"memory");
#define PP_ADD(I, T0, T1) pp_acc[I] += (T1) - (T0);  // accumulated in registers, stored once at the end
#define PP_CNT(I) pp_acc[I] += 1;
#define PP_DECL unsigned long long pp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    auto pp_unvoted = [&](int n_) { pp_acc[8] += n_; };
#define PP_FLUSH                          \
    if (lane == 0 && scan == 0)           \
        for (int i_ = 0; i_ < 12; i_++) g_ppht_stamps[i_] += pp_acc[i_];
hipError_t debug_ppht_stamps(unsigned long long out[12], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ppht_stamps), 12 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ppht_stamps), z, sizeof z);
    }
    return e;
}
#else
#define PP_CLK(V)
#define PP_ADD(I, T0, T1)
#define PP_CNT(I)
#define PP_DECL auto pp_unvoted = [](int) {};
#define PP_FLUSH
#endif

// ---- wave 1: the list -> the sequence of drawn points
__device__ __forceinline__ void ppht_draw(const PphtArgs &a, PphtShared &sh, int scan, int lane)
{
    uint32_t *nz = a.nz + a.scan_off[scan];
    uint32_t *order = a.order + a.scan_off[scan];
    int count = a.count[scan], k0 = 0;
    unsigned long long rng = ~0ull;  // cv::RNG((uint64)-1), stepped by lane 0
    volatile uint32_t *shr = sh.r;
    while (count > 0) {
        const int nd = min(64, count);
        if (lane == 0) {
            for (int t = 0; t < nd; t++) {
                rng = (unsigned long long)(uint32_t)rng * 4164903690u + (uint32_t)(rng >> 32);
                shr[t] = (uint32_t)rng;
            }
        }
        __builtin_amdgcn_wave_barrier();  // same wave: LDS operations complete in order
        const bool act = lane < nd;
        const uint32_t c = (uint32_t)(count - lane);  // list length at this lane's draw
        uint32_t idx = 0xffffffffu, last = 0xfffffffeu, pv = 0, qv = 0, res = 0;
        if (act) {
            idx = shr[lane] % c;
            last = c - 1;
            pv = list_load(nz + idx);
            qv = list_load(nz + last);
        }
        // draw t takes the point in slot idx_t and moves the list's last point (slot c_t - 1) there: replayed in
        // order on the registers of the lanes that hold those slots
        for (int t = 0; t < nd; t++) {
            const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)idx, t);
            const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)qv, t);
            if (lane == t) res = pv;
            if (idx == it) pv = q;
            if (last == it) qv = q;
        }
        if (act) {
            list_store(nz + idx, pv);  // lanes that share a slot hold the same value; slots past the new end are dead
            list_store(order + k0 + lane, res);
        }
        count -= nd;
        k0 += nd;
        // the stores above are complete (and the next round's loads see them) before the progress is published
        __hip_atomic_store(&sh.produced, k0, __ATOMIC_RELEASE, PP_WG);
    }
}

// ---- un-votes of an accepted segment: sh.pts[0 .. npts), angles n0 .. n1 - 1.  Lane = point, one angle after the
// other: the decrements of an instruction fall into a few cache lines of one accumulator row.  A no-return atomic
// costs a wave 70-230 cycles to issue whatever its lanes do, and a segment needs 180 x ceil(points / 64) of them, so
// the angles are split between wave 0 and the helper waves (an accepted segment is rare -- 4 % of the served
// points -- but its un-votes were a fifth of a scan's time on one wave).
//
// The accumulator comes in two widths.  U16: a bin is an unsigned short holding count + 0x8080 (the buffer is filled with
// the byte 0x80).  A bin's count stays within +-(pixels of one rho strip) <= the image diagonal -- every pixel votes at
// most once and is un-voted at most once, and OpenCV's un-votes can run ahead of the votes (negative counts) -- so the
// host may pick U16 whenever rows + cols <= OMR_PPHT_U16_MAX_EXTENT: half the footprint in L2 / Infinity Cache, twice the
// bins per line -- and measurably SLOWER than int32 bins (hough.hpp), so the shipped library never does.
// An un-vote is a 32-bit atomic subtract of 1 or 1 << 16 on the word that holds the bin (no borrow can cross: the
// biased count never reaches 0).  Arg-max compares the biased values (same order) and removes the bias once.
#define OMR_PPHT_U16_BIAS 0x8080
template <bool U16>
__device__ __forceinline__ int acc_load(char *accum, uint32_t off)
{
    if (U16) return (int)__hip_atomic_load((uint16_t *)(accum + off), __ATOMIC_RELAXED, PP_WG);
    return __hip_atomic_load((int32_t *)(accum + off), __ATOMIC_RELAXED, PP_WG);
}
template <bool U16>
__device__ __forceinline__ void acc_store(char *accum, uint32_t off, int v)
{
    if (U16) __hip_atomic_store((uint16_t *)(accum + off), (uint16_t)v, __ATOMIC_RELAXED, PP_WG);
    else __hip_atomic_store((int32_t *)(accum + off), v, __ATOMIC_RELAXED, PP_WG);
}
template <bool U16>
__device__ __forceinline__ void unvote_points(PphtShared &sh, char *accum, int npts, int n0, int n1, int lane)
{
    constexpr uint32_t BIN = U16 ? 2u : 4u;
    for (int g = 0; g < npts; g += 64) {
        if (g + lane < npts) {
            const uint32_t q = sh.pts[g + lane];
            const float fj = (float)(q & 0xffffu), fi = (float)(q >> 16);
#pragma unroll 4
            for (int n = n0; n < n1; n++) {
                const float4 t = sh.ang[n];  // (cos, sin, byte offset of the row's bin 0, -): LDS broadcast
                const uint32_t o = __float_as_uint(t.z) + BIN * (uint32_t)__float2int_rn(__fadd_rn(__fmul_rn(fj, t.x), __fmul_rn(fi, t.y)));
                if (U16) __hip_atomic_fetch_sub((uint32_t *)(accum + (o & ~3u)), 1u << ((o & 2u) << 3), __ATOMIC_RELAXED, PP_WG);
                else __hip_atomic_fetch_sub((int32_t *)(accum + o), 1, __ATOMIC_RELAXED, PP_WG);
            }
        }
    }
}
#define OMR_PPHT_HELPERS 3  // waves 2 .. 4
__device__ __forceinline__ void angle_share(int numangle, int part, int &n0, int &n1)  // part 0 = wave 0
{
    const int per = (numangle + OMR_PPHT_HELPERS) / (OMR_PPHT_HELPERS + 1);
    n0 = min(numangle, part * per);
    n1 = min(numangle, n0 + per);
}
template <bool U16>
__device__ __forceinline__ void ppht_help(const PphtArgs &a, PphtShared &sh, int scan, int lane, int part)
{
    char *accum = (char *)a.accum + (int64_t)scan * a.accum_stride * (U16 ? 2 : 4);
    int n0, n1;
    angle_share(a.numangle, part, n0, n1);
    for (int seen = 0;;) {
        int seq;
        while ((seq = __hip_atomic_load(&sh.job_seq, __ATOMIC_ACQUIRE, PP_WG)) == seen) __builtin_amdgcn_s_sleep(2);
        if (seq < 0) break;  // the scan is finished
        seen = seq;
        unvote_points<U16>(sh, accum, sh.job_npts, n0, n1, lane);
        // release: the decrements are performed before wave 0 reads the accumulator again
        if (lane == 0) __hip_atomic_fetch_add(&sh.job_done, 1, __ATOMIC_RELEASE, PP_WG);
    }
}

// ---- wave 0: votes, walks, segments.  A single wave's instruction stream is the critical path here (about five
// cycles per instruction), so the code below counts instructions: 32-bit byte offsets from uniform bases, 24-bit
// multiplies, no loops over set bits, no predicated loads (a load inside a divergent branch is waited for before
// the next one is issued).
struct PphtPos {
    uint32_t off;  // byte offset of the mask word
    int bit, i1, j1;
    bool inx, iny;  // column / row inside the image (kept apart: a ballot of ONE compare is the compare's own lane mask,
                    // a ballot of "a && b" is re-materialised through a VGPR)
};
struct PphtLine {  // uniform description of the walk of one served point
    int x0, y0, dx0, dy0, shx, shy;
};
__device__ __forceinline__ PphtPos walk_pos(const PphtLine &ln, int d, int t, int W, int H, int TX)
{
    PphtPos p;
    p.j1 = (ln.x0 + __mul24(t, d ? -ln.dx0 : ln.dx0)) >> ln.shx;
    p.i1 = (ln.y0 + __mul24(t, d ? -ln.dy0 : ln.dy0)) >> ln.shy;
    p.inx = (uint32_t)p.j1 < (uint32_t)W;
    p.iny = (uint32_t)p.i1 < (uint32_t)H;
    p.off = (uint32_t)(__mul24(p.i1 >> 3, TX) + (p.j1 >> 3)) << 3;
    p.bit = ((p.i1 & 7) << 3) | (p.j1 & 7);
    return p;
}
__device__ __forceinline__ unsigned long long mask_at(const unsigned long long *mask, uint32_t off)
{
    return __hip_atomic_load((const unsigned long long *)((const char *)mask + off), __ATOMIC_RELAXED, PP_WG);
}

// The sequential gap rule of one direction over the 128 steps of a round (hough.cpp's first pass: a set point
// resets the gap and moves the line end, every other step adds one, "++gap > lineGap" or the border ends the walk)
// without a loop: a set point is unreachable when the run of zeros before it is longer than the gap, the walk ends
// at the first unreachable one.  n0 / n1: set points of steps base .. +63 / +64 .. +127; o0 / o1: steps outside
// the image.  gap, end_t, stop: the direction's state (in / out).
__device__ __forceinline__ void gap_rule(unsigned long long n0, unsigned long long n1, unsigned long long o0, unsigned long long o1,
                                         int base, int line_gap, int lane, int &gap, int &end_t, bool &stop)
{
    const int valid0 = o0 ? __ffsll((long long)o0) - 1 : 64;  // steps before the border
    const int valid1 = valid0 < 64 ? 0 : (o1 ? __ffsll((long long)o1) - 1 : 64);
    if (valid0 < 64) n0 &= (1ull << valid0) - 1ull;
    if (valid1 < 64) n1 &= (1ull << valid1) - 1ull;
    const int valid = valid0 + valid1;
    const int top0 = n0 ? 63 - __clzll((long long)n0) : -1 - gap;  // last set step of the first half (or the state's)
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long b0 = n0 & lt, b1 = n1 & lt;
    const int z0 = b0 ? lane - (63 - __clzll((long long)b0)) - 1 : lane + gap;        // zeros before step lane
    const int z1 = b1 ? lane - (63 - __clzll((long long)b1)) - 1 : 63 + lane - top0;  // zeros before step 64 + lane
    const unsigned long long bad0 = __builtin_amdgcn_ballot_w64(((n0 >> lane) & 1ull) && z0 > line_gap);
    const unsigned long long bad1 = __builtin_amdgcn_ballot_w64(((n1 >> lane) & 1ull) && z1 > line_gap);
    unsigned long long k0 = n0, k1 = n1;  // the set points the walk reaches
    bool st = false;
    if (bad0) {
        k0 = n0 & ((bad0 & (0ull - bad0)) - 1ull);
        k1 = 0;
        st = true;
    } else if (bad1) {
        k1 = n1 & ((bad1 & (0ull - bad1)) - 1ull);
        st = true;
    }
    if (k1) end_t = base + 127 - __clzll((long long)k1);
    else if (k0) end_t = base + 63 - __clzll((long long)k0);
    if (!st) {
        int g;
        if (n1) g = valid - (127 - __clzll((long long)n1)) - 1;
        else if (n0) g = valid - (63 - __clzll((long long)n0)) - 1;
        else g = gap + valid;
        gap = g;
        st = g > line_gap || valid < 128;
    }
    stop = st;
}

// The same rule for lineGap < 64 (the reference passes 50), an order of magnitude fewer instructions: a step ends
// the walk ("++gap > lineGap") exactly when none of the lineGap + 1 steps up to and including it holds a point --
// one shift of the ballots per lane.  Steps outside the image never hold a point and a line that has left the image
// does not come back, so the border needs no masks: the walk stops in this round whenever a step is outside.
__device__ __forceinline__ void gap_rule_short(unsigned long long n0, unsigned long long n1, unsigned long long o0, unsigned long long o1,
                                               int base, int line_gap, int lane, int &gap, int &end_t, bool &stop)
{
    const int up = 63 - lane, down = 63 - line_gap;
    const unsigned long long w0 = (n0 << up) >> down;                             // steps lane - lineGap .. lane
    const unsigned long long w1 = ((n1 << up) | ((n0 >> 1) >> lane)) >> down;     // steps 64 + lane - lineGap .. 64 + lane
    // the state's last point lies lane + 1 + gap steps back: it covers the steps lane < lineGap - gap
    const int first = line_gap - gap;
    const unsigned long long uncovered = first <= 0 ? ~0ull : first >= 64 ? 0ull : ~0ull << first;
    const unsigned long long brk0 = __builtin_amdgcn_ballot_w64(w0 == 0) & uncovered;
    const unsigned long long brk1 = __builtin_amdgcn_ballot_w64(w1 == 0);
    unsigned long long k0 = n0, k1 = n1;  // the points the walk reaches
    if (brk0) {
        k0 &= (brk0 & (0ull - brk0)) - 1ull;
        k1 = 0;
    } else if (brk1) {
        k1 &= (brk1 & (0ull - brk1)) - 1ull;
    }
    if (k1) end_t = base + 127 - __clzll((long long)k1);
    else if (k0) end_t = base + 63 - __clzll((long long)k0);
    stop = (brk0 | brk1 | o0 | o1) != 0;
    gap = n1 ? __clzll((long long)n1) : n0 ? 64 + __clzll((long long)n0) : gap + 128;  // only read when the walk goes on
}

#define OMR_PPHT_PTS 1024  // LDS list of a segment's points waiting for their un-votes
template <int NPL, bool SHORT_GAP, bool U16>  // accumulator angles per lane; lineGap < 64 (the loop-free rule with one shift); bin width
__device__ __forceinline__ void ppht_serve(const PphtArgs &a, PphtShared &sh, int scan, int lane)
{
    const int W = a.width, H = a.height;
    unsigned long long *mask = (unsigned long long *)a.mask + (int64_t)scan * (ppht_mask_bytes(H, W) / 8);
    const int TX = ppht_tiles_x(W);
    const uint32_t *order = a.order + a.scan_off[scan];
    constexpr uint32_t BIN = U16 ? 2u : 4u;
    char *accum = (char *)a.accum + (int64_t)scan * a.accum_stride * (int64_t)BIN;
    int32_t *lines = a.lines + (int64_t)scan * a.cap * 4;
    const int N = a.count[scan];
    // Lanes without an angle (numangle is 180: lanes 52-63 of the third set) vote for a scratch bin of their own
    // behind the scan's rows: no predication around the loads, stores and atomics.
    float tc[NPL], ts[NPL];
    uint32_t rowb[NPL];  // byte offset of the bin rho = 0 of this lane's rows
    bool voter[NPL];
#pragma unroll
    for (int v = 0; v < NPL; v++) {
        const int n = lane + 64 * v;
        voter[v] = n < a.numangle;
        tc[v] = voter[v] ? a.ttab[2 * n] : 0.f;
        ts[v] = voter[v] ? a.ttab[2 * n + 1] : 0.f;
        rowb[v] = BIN * (uint32_t)(voter[v] ? (int64_t)a.row_base[n] : a.accum_stride - 64 + lane);
    }
    auto bin_off = [&](int v, float fj, float fi) -> uint32_t {
        return rowb[v] + BIN * (uint32_t)__float2int_rn(__fadd_rn(__fmul_rn(fj, tc[v]), __fmul_rn(fi, ts[v])));
    };
    int k0 = 0, nl = 0, jobs = 0;
    PP_DECL
    uint32_t pt = 0;              // this lane's drawn point of the current round
    unsigned long long pend = 0;  // lanes whose point is still set and not served yet
    // the bins of the point served next, read ahead while the current point walks
    bool ahead = false;
    int ahead_lane = 0;
    uint32_t aoff[NPL];
    int aval[NPL];
#pragma unroll
    for (int v = 0; v < NPL; v++) {
        aoff[v] = 0;
        aval[v] = 0;
    }

    for (;;) {
        PP_CLK(c0)
        while (pend == 0 && k0 < N) {  // ---- the next 64 drawn points
            const int nd = min(64, N - k0);
            while (__hip_atomic_load(&sh.produced, __ATOMIC_ACQUIRE, PP_WG) < k0 + nd) __builtin_amdgcn_s_sleep(1);
            bool on = false;
            if (lane < nd) {
                pt = list_load(order + k0 + lane);
                const int y = (int)(pt >> 16), x = (int)(pt & 0xffffu);
                on = (mask_at(mask, (uint32_t)(__mul24(y >> 3, TX) + (x >> 3)) << 3) >> (((y & 7) << 3) | (x & 7))) & 1ull;
            }
            pend = __builtin_amdgcn_ballot_w64(on);
            k0 += nd;
            ahead = false;
        }
        if (pend == 0) break;
        const int tsel = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pt, tsel);
        const int pi = (int)(p >> 16), pj = (int)(p & 0xffffu);
        PP_CLK(c1)
        PP_ADD(0, c0, c1)
        PP_CNT(5)
        // ---- vote: r = cvRound(j * cos/rho + i * sin/rho) in float32, no contraction.  Load + store pairs, not
        // RMW atomics: a row is only ever touched by this lane.
        uint32_t off[NPL];
        int val[NPL];
        if (ahead && ahead_lane == tsel) {
#pragma unroll
            for (int v = 0; v < NPL; v++) {
                off[v] = aoff[v];
                val[v] = aval[v];
            }
        } else {
            const float fj = (float)pj, fi = (float)pi;
#pragma unroll
            for (int v = 0; v < NPL; v++) {
                off[v] = bin_off(v, fj, fi);
                val[v] = acc_load<U16>(accum, off[v]);
            }
        }
        int key = (int)0x80000000;
#pragma unroll
        for (int v = 0; v < NPL; v++) {
            val[v] += 1;
            acc_store<U16>(accum, off[v], val[v]);
            const int k = (int)(((uint32_t)val[v] << 8) | (uint32_t)(255 - (lane + 64 * v)));
            key = voter[v] ? max(key, k) : key;
        }
        // read ahead: the bins of the next pending point, after this point's stores (same lane, same address: in order)
        ahead = pend != 0;
        if (ahead) {
            ahead_lane = __ffsll((long long)pend) - 1;
            const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)pt, ahead_lane);
            const float fj = (float)(q & 0xffffu), fi = (float)(q >> 16);
#pragma unroll
            for (int v = 0; v < NPL; v++) {
                aoff[v] = bin_off(v, fj, fi);
                aval[v] = acc_load<U16>(accum, aoff[v]);
            }
        }
        key = wave_max_i32(key);
        const int max_val = (key >> 8) - (U16 ? OMR_PPHT_U16_BIAS : 0), max_n = 255 - (key & 255);
        PP_CLK(c2)
        PP_ADD(1, c1, c2)
        if (max_val < a.threshold) continue;  // with threshold 0 only when un-votes drove the bins negative

        PphtLine ln;
        {
            const int4 wk = *(const int4 *)&sh.walk[max_n];  // one LDS broadcast read
            const int xflag = __builtin_amdgcn_readfirstlane(wk.x);
            ln.dx0 = __builtin_amdgcn_readfirstlane(wk.y);
            ln.dy0 = __builtin_amdgcn_readfirstlane(wk.z);
            ln.x0 = xflag ? pj : (pj << 16) + (1 << 15);
            ln.y0 = xflag ? (pi << 16) + (1 << 15) : pi;
            ln.shx = xflag ? 0 : 16;
            ln.shy = xflag ? 16 : 0;
        }

        // ---- first pass: the segment's two ends.  Slot s = 2 * dir + half: steps base + 64 * half + lane.
        int gap[2] = {0, 0}, end_t[2] = {0, 0};  // step 0 is the served point itself: non-zero
        bool stop[2] = {false, false};
        unsigned long long on0[4];  // round 0's flags and positions, kept for the second pass
        PphtPos ps0[4];
        auto walk_round = [&](int base, PphtPos (&ps)[4], unsigned long long (&bn)[4]) {
            PP_CNT(6)
            unsigned long long word[4], bo[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                ps[s] = walk_pos(ln, s >> 1, base + 64 * (s & 1) + lane, W, H, TX);
                word[s] = mask_at(mask, (ps[s].inx && ps[s].iny) ? ps[s].off : 0u);  // (a stopped direction's reads are ignored)
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {  // (all 64 lanes are active here)
                const unsigned long long in = __builtin_amdgcn_ballot_w64(ps[s].inx) & __builtin_amdgcn_ballot_w64(ps[s].iny);
                bn[s] = __builtin_amdgcn_ballot_w64((word[s] & (1ull << ps[s].bit)) != 0) & in;
                bo[s] = ~in;
            }
#pragma unroll
            for (int d = 0; d < 2; d++)
                if (!stop[d]) {
                    if (SHORT_GAP) gap_rule_short(bn[2 * d], bn[2 * d + 1], bo[2 * d], bo[2 * d + 1], base, a.line_gap, lane, gap[d], end_t[d], stop[d]);
                    else gap_rule(bn[2 * d], bn[2 * d + 1], bo[2 * d], bo[2 * d + 1], base, a.line_gap, lane, gap[d], end_t[d], stop[d]);
                }
        };
        walk_round(0, ps0, on0);
#ifdef OMR_RUNS_DEBUG
        if ((end_t[0] + a.line_gap + 1 < 64) && (end_t[1] + a.line_gap + 1 < 64)) PP_CNT(11)
#endif
        for (int base = 128; !(stop[0] && stop[1]); base += 128) {
            PphtPos ps[4];
            unsigned long long bn[4];
            walk_round(base, ps, bn);
        }
        PP_CLK(c3)
        PP_ADD(2, c2, c3)
        // line ends and the length test
        int ex[2], ey[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            ex[k] = (ln.x0 + end_t[k] * (k ? -ln.dx0 : ln.dx0)) >> ln.shx;
            ey[k] = (ln.y0 + end_t[k] * (k ? -ln.dy0 : ln.dy0)) >> ln.shy;
        }
        const bool good = abs(ex[1] - ex[0]) >= a.line_length || abs(ey[1] - ey[0]) >= a.line_length;

        // ---- second pass: erase the segment's points; un-vote them when the segment is accepted.  Round 0 uses
        // the first pass's flags and positions (the mask has not changed since); later rounds read the mask again.
        // The points of an accepted segment are collected in LDS and un-voted together.
        int npts = 0;
        auto unvote = [&]() {
            PP_CLK(u0)
            pp_unvoted(npts);
            int n0, n1;
            angle_share(a.numangle, 0, n0, n1);
            sh.job_npts = npts;
            __hip_atomic_store(&sh.job_seq, ++jobs, __ATOMIC_RELEASE, PP_WG);  // the points are in LDS before the helpers start
            unvote_points<U16>(sh, accum, npts, n0, n1, lane);
            while (__hip_atomic_load(&sh.job_done, __ATOMIC_ACQUIRE, PP_WG) != OMR_PPHT_HELPERS * jobs) __builtin_amdgcn_s_sleep(1);
            PP_CLK(u1)
            PP_ADD(10, u0, u1)
            npts = 0;
        };
        auto erase_round = [&](int base, const PphtPos (&ps)[4], unsigned long long (&m)[4]) {
            PP_CNT(7)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int d = s >> 1, lo = base + 64 * (s & 1);
                const int cnt = min(64, end_t[d] - lo + 1);  // steps of this slot that belong to the segment
                unsigned long long keep = cnt <= 0 ? 0ull : cnt >= 64 ? ~0ull : (1ull << cnt) - 1ull;
                if (d == 1 && lo == 0) keep &= ~1ull;  // step 0 belongs to direction 0
                m[s] &= keep;
                if (m[s] == 0) continue;  // (uniform: most rounds touch one or two of the four slots)
                if ((m[s] >> lane) & 1ull) {
                    // several lanes may clear bits of one word: atomic AND, no return value
                    __hip_atomic_fetch_and((unsigned long long *)((char *)mask + ps[s].off), ~(1ull << ps[s].bit), __ATOMIC_RELAXED, PP_WG);
                    // (the list is only read when the segment is accepted)
                    const int k = npts + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[s] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[s], 0));
                    sh.pts[k] = ((uint32_t)ps[s].i1 << 16) | (uint32_t)ps[s].j1;
                }
                npts += __popcll(m[s]);
            }
        };
        erase_round(0, ps0, on0);
        const int last_max = max(end_t[0], end_t[1]);
        for (int base = 128; base <= last_max; base += 128) {
            unsigned long long m[4], word[4];
            PphtPos ps[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                ps[s] = walk_pos(ln, s >> 1, base + 64 * (s & 1) + lane, W, H, TX);
                word[s] = mask_at(mask, (ps[s].inx && ps[s].iny) ? ps[s].off : 0u);
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
                m[s] = __builtin_amdgcn_ballot_w64((word[s] & (1ull << ps[s].bit)) != 0) & __builtin_amdgcn_ballot_w64(ps[s].inx) &
                       __builtin_amdgcn_ballot_w64(ps[s].iny);
            if (npts > OMR_PPHT_PTS - 256) {
                if (good) unvote();
                npts = 0;
            }
            erase_round(base, ps, m);
        }
        if (good) {
            unvote();
            ahead = false;  // the bins read ahead may have been decremented
            PP_CNT(9)
            if (lane == 0 && nl < a.cap) {
                lines[4 * nl] = ex[0];
                lines[4 * nl + 1] = ey[0];
                lines[4 * nl + 2] = ex[1];
                lines[4 * nl + 3] = ey[1];
            }
            nl++;
        }
        PP_CLK(c4)
        PP_ADD(3, c3, c4)
        // ---- the points this wave still holds: one of them is erased exactly when it is a walked position of the
        // segment (it was set before the walk, and the second pass clears every set position up to the ends)
        if (pend) {
            const int ci = (int)(pt >> 16), cj = (int)(pt & 0xffffu);
            const int dd = ln.shx ? (ci - pi) * ln.dy0 : (cj - pj) * ln.dx0;  // along the unit axis: step = |dd|, direction = sign
            const int d = dd < 0, t = abs(dd);
            const PphtPos c = walk_pos(ln, d, t, W, H, TX);
            const bool hit = t <= (d ? end_t[1] : end_t[0]) && c.j1 == cj && c.i1 == ci;
            pend &= ~__builtin_amdgcn_ballot_w64(hit);
        }
        PP_CLK(c6)
        PP_ADD(4, c4, c6)
    }
    if (lane == 0) a.n_lines[scan] = nl;
    __hip_atomic_store(&sh.job_seq, -1, __ATOMIC_RELEASE, PP_WG);  // the helper waves leave
    PP_FLUSH
}

// (NPL, SHORT_GAP) are template parameters of the KERNEL: each launch then runs a body that holds one vote width and one
// gap rule only (the reference's parameters: 180 angles, lineGap 15 .. 75).
template <int NPL, bool SHORT_GAP, bool U16>
__global__ __launch_bounds__(OMR_PPHT_THREADS) void ppht_kernel(const PphtArgs a)
{
    __shared__ PphtShared sh;
    __shared__ int next_scan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < a.numangle; k += OMR_PPHT_THREADS) {
        sh.walk[k] = a.walk[k];
        sh.ang[k] = make_float4(a.ttab[2 * k], a.ttab[2 * k + 1], __uint_as_float((U16 ? 2u : 4u) * (uint32_t)a.row_base[k]), 0.f);
    }
    // The grid holds as many workgroups as scans are meant to be in flight (launch_ppht); a workgroup that has finished
    // its scan takes the next one from the queue, so the number of accumulators and masks being worked on -- the
    // footprint in L2 and Infinity Cache -- is the grid size whatever the batch size.  Every wave leaves its role when
    // the scan is finished and the queue runs dry for all workgroups: the loop ends.
    for (int scan = blockIdx.x; scan < a.n_scans;) {
        if (tid == 0) {
            sh.produced = 0;
            sh.job_seq = 0;
            sh.job_done = 0;
        }
        __syncthreads();  // from here on the waves run on their own until the scan is finished
        if (wave == 1) ppht_draw(a, sh, scan, lane);
        else if (wave >= 2) ppht_help<U16>(a, sh, scan, lane, wave - 1);
        else ppht_serve<NPL, SHORT_GAP, U16>(a, sh, scan, lane);
        if (tid == 0) next_scan = (int)gridDim.x + atomicAdd(a.queue, 1);
        __syncthreads();
        scan = next_scan;
    }
}

template <int NPL, bool SHORT_GAP>
static void launch_ppht_width(const PphtArgs &a, int n, hipStream_t s)  // n = workgroups = scans in flight
{
    if (a.acc_u16) hipLaunchKernelGGL((ppht_kernel<NPL, SHORT_GAP, true>), dim3(n), dim3(OMR_PPHT_THREADS), 0, s, a);
    else hipLaunchKernelGGL((ppht_kernel<NPL, SHORT_GAP, false>), dim3(n), dim3(OMR_PPHT_THREADS), 0, s, a);
}
hipError_t launch_ppht(const PphtArgs &a, int in_flight, hipStream_t s)
{
    const int n = in_flight > 0 ? (in_flight < a.n_scans ? in_flight : a.n_scans) : a.n_scans;
    if (n <= 0) return hipSuccess;
    if (!a.queue) return hipErrorInvalidValue;
    if (a.numangle > OMR_PPHT_MAX_ANGLES) return hipErrorInvalidValue;
    if (a.acc_u16 && (a.accum_stride & 1)) return hipErrorInvalidValue;  // a scan's bins start on a 32-bit word
    const bool short_gap = (uint32_t)a.line_gap < 64u;
    if (a.numangle <= 192) {
        if (short_gap) launch_ppht_width<3, true>(a, n, s);
        else launch_ppht_width<3, false>(a, n, s);
    } else {
        if (short_gap) launch_ppht_width<4, true>(a, n, s);
        else launch_ppht_width<4, false>(a, n, s);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void angle_votes_kernel(const float *__restrict__ ang, int n, int as_f64,
                                                          int32_t *__restrict__ counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float ai = ang[i];
    int c = 0;
    if (as_f64) {
        const double di = (double)ai;
        for (int j = 0; j < n; j++) c += fabs(di - (double)ang[j]) < 0.1;
    } else {
        for (int j = 0; j < n; j++) c += fabsf(__fsub_rn(ai, ang[j])) < 0.1f;
    }
    counts[i] = c;
}

__global__ __launch_bounds__(256) void angle_votes_batch_kernel(const float *__restrict__ ang,
                                                                const int64_t *__restrict__ off, int as_f64,
                                                                int32_t *__restrict__ counts)
{
    const int64_t o = off[blockIdx.y];
    const int n = (int)(off[blockIdx.y + 1] - o);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *a = ang + o;
    const float ai = a[i];
    int c = 0;
    if (as_f64) {
        const double di = (double)ai;
        for (int j = 0; j < n; j++) c += fabs(di - (double)a[j]) < 0.1;
    } else {
        for (int j = 0; j < n; j++) c += fabsf(__fsub_rn(ai, a[j])) < 0.1f;
    }
    counts[o + i] = c;
}

hipError_t launch_angle_votes_batch(const float *d_angles, const int64_t *d_off, int n_scans, int max_n, int as_f64,
                                    int32_t *d_counts, hipStream_t s)
{
    if (n_scans <= 0 || max_n <= 0) return hipSuccess;
    hipLaunchKernelGGL(angle_votes_batch_kernel, dim3((max_n + 255) / 256, n_scans), dim3(256), 0, s, d_angles, d_off,
                       as_f64, d_counts);
    return hipGetLastError();
}

hipError_t launch_angle_votes(const float *d_angles, int n, int as_f64, int32_t *d_counts, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(angle_votes_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_angles, n, as_f64, d_counts);
    return hipGetLastError();
}

}  // namespace omr
