// hough.hip -- Canny + progressive probabilistic Hough (OpenCV 4.6.0 semantics) for gfx950.
//
// Canny is data-parallel and exact: Sobel / non-maximum suppression are integer stencils and the
// hysteresis result is a set (the candidates 8-connected to a strong pixel), so the order of
// propagation does not matter.  HoughLinesP is a *sequential* randomised algorithm (hough.cpp
// HoughLinesProbabilistic: every drawn point sees the accumulator and the mask left by all points
// before it), so the kernel keeps the sequence and parallelises inside a step -- one workgroup per
// scan, lane = accumulator angle for the 180 votes, lane = position for the line walks -- and
// across scans (one workgroup each; a batch fills the chip).  Same RNG (cv::RNG seed 2^64-1), same
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
// HoughLinesProbabilistic, one workgroup (4 waves) per scan.
//   wave 0       : the draw stage.  The draw order does not depend on the image (RNG + swap-remove
//                  on the point list), so 64 draws are made per round: lane 0 steps the RNG 64 times,
//                  every lane takes one draw (index, the point, the element swapped in), the swaps
//                  are committed together when no two draws of the round touch the same list slot
//                  (else lane 0 replays the round one by one), and the 64 mask tests are one memory
//                  round trip.  Points still set are then served lowest lane first; after every
//                  walk the remaining ones are re-tested (the walk may have erased them).
//   lane = angle : the 180 accumulator increments of a served point and their arg-max; the
//                  decrements ("un-votes") of an accepted segment's points (fire-and-forget atomics)
//   lane = step  : the two walks along the chosen line run side by side (threads 0-127 one way,
//                  128-255 the other, 128 positions per round); the sequential gap rule is applied
//                  to the rounds' ballots by threads 0 and 128
// Row n of the accumulator is only ever touched by lane n, so its updates are ordered; mask bytes
// and list slots are read and written with device-scope (L2) accesses between barriers.
struct PphtShared {
    int i, j;                      // served point (i = row, j = column), i < 0: list exhausted
    unsigned long long key[4];     // per-wave (value, angle) maxima
    unsigned long long nzb[4];     // walk round: non-zero ballots
    unsigned long long oob[4];     // walk round: out-of-image ballots
    int stop[2], gap[2], end_t[2]; // walk state of both directions
    uint32_t r[64];                // draw round: raw RNG outputs (replay: the drawn points)
    int pts[OMR_PPHT_THREADS];     // points to un-vote in this round (y << 16 | x)
};

// device-scope accesses: served by L2, so a wave sees what other waves did before the last barrier
__device__ __forceinline__ bool mask_test(const unsigned long long *mask, int y, int x, int tx)
{
    const unsigned long long w = __hip_atomic_load(mask + mask_word(y, x, tx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (w >> mask_bit(y, x)) & 1ull;
}
__device__ __forceinline__ void mask_clear(unsigned long long *mask, int y, int x, int tx)
{
    // several lanes may clear bits of one word: atomic AND, no return value
    __hip_atomic_fetch_and(mask + mask_word(y, x, tx), ~(1ull << mask_bit(y, x)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t list_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void list_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef OMR_RUNS_DEBUG
// debug build only: wave 0 / lane 0 phase clocks of scan 0 [draw, vote+argmax, walk pass 1, pass 2 + un-vote,
// re-test, served points, pass-1 rounds, pass-2 rounds]; read with omr_debug_ppht_stamps()
__device__ unsigned long long g_ppht_stamps[8];
#define PP_CLK(V)                                                                  \
    unsigned long long V = 0;                                                      \
    if (tid == 0 && scan == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V)::"memory");
#define PP_ADD(I, T0, T1) \
    if (tid == 0 && scan == 0) g_ppht_stamps[I] += (T1) - (T0);
#define PP_CNT(I) \
    if (tid == 0 && scan == 0) g_ppht_stamps[I] += 1;
hipError_t debug_ppht_stamps(unsigned long long out[8], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ppht_stamps), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ppht_stamps), z, sizeof z);
    }
    return e;
}
#else
#define PP_CLK(V)
#define PP_ADD(I, T0, T1)
#define PP_CNT(I)
#endif

__global__ __launch_bounds__(OMR_PPHT_THREADS) void ppht_kernel(const PphtArgs a)
{
    __shared__ PphtShared sh;
    const int scan = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.width, H = a.height;
    unsigned long long *mask = (unsigned long long *)a.mask + (int64_t)scan * (ppht_mask_bytes(H, W) / 8);
    const int TX = ppht_tiles_x(W);
    uint32_t *nz = a.nz + a.scan_off[scan];
    int32_t *accum = a.accum + (int64_t)scan * a.numangle * a.numrho;
    int32_t *lines = a.lines + (int64_t)scan * a.cap * 4;
    const bool voter = tid < a.numangle;
    float tc = 0.f, ts = 0.f;
    if (voter) {
        tc = a.ttab[2 * tid];
        ts = a.ttab[2 * tid + 1];
    }
    int32_t *row = accum + (int64_t)(voter ? tid : 0) * a.numrho + (a.numrho - 1) / 2;
    const int dir = tid >> 7, slot = tid & 127;  // walk role
    // wave 0 state
    unsigned long long rng = ~0ull;  // cv::RNG((uint64)-1), stepped by lane 0
    int count = a.count[scan];
    uint32_t pt = 0;                 // this lane's drawn point of the current round
    unsigned long long pend = 0;     // lanes whose point is still set and not served yet
    int nl = 0;
    volatile uint32_t *shr = sh.r;

    for (;;) {
        PP_CLK(c0)
        if (wave == 0) {
            while (pend == 0 && count > 0) {  // ---- a draw round
                const int nd = min(64, count);
                if (lane == 0) {
                    for (int t = 0; t < nd; t++) {
                        rng = (unsigned long long)(uint32_t)rng * 4164903690u + (uint32_t)(rng >> 32);
                        shr[t] = (uint32_t)rng;
                    }
                }
                __builtin_amdgcn_wave_barrier();  // same wave: LDS operations complete in order
                const bool act = lane < nd;
                const int c = count - lane;  // list length at this lane's draw
                uint32_t idx = 0xffffffffu, p = 0, q = 0;
                if (act) {
                    idx = shr[lane] % (uint32_t)c;
                    p = list_load(nz + idx);
                    q = list_load(nz + (c - 1));
                }
                // draw s writes slot idx_s; a later draw t reads slots idx_t and c_t - 1
                bool conf = false;
                for (int s = 0; s + 1 < nd; s++) {
                    const uint32_t is = (uint32_t)__builtin_amdgcn_readlane((int)idx, s);
                    conf |= act && lane > s && (idx == is || (uint32_t)(c - 1) == is);
                }
                if (__ballot(conf)) {  // rare: replay the round in order
                    if (lane == 0) {
                        int cc = count;
                        for (int t = 0; t < nd; t++) {
                            const uint32_t ix = shr[t] % (uint32_t)cc;
                            const uint32_t pp = list_load(nz + ix);
                            list_store(nz + ix, list_load(nz + (cc - 1)));
                            cc--;
                            shr[t] = pp;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    p = act ? shr[lane] : 0;
                } else if (act) {
                    list_store(nz + idx, q);
                }
                count -= nd;
                pt = p;
                const bool on = act && mask_test(mask, (int)(p >> 16), (int)(p & 0xffffu), TX);
                pend = __ballot(on);
            }
            int pi = -1, pj = -1;
            if (pend) {
                const int t = __ffsll((long long)pend) - 1;
                pend &= pend - 1;
                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pt, t);
                pi = (int)(p >> 16);
                pj = (int)(p & 0xffffu);
            }
            if (lane == 0) {
                sh.i = pi;
                sh.j = pj;
                sh.stop[0] = sh.stop[1] = 0;
                sh.gap[0] = sh.gap[1] = 0;
                sh.end_t[0] = sh.end_t[1] = 0;  // step 0 is the served point itself: non-zero
            }
        }
        __syncthreads();
        const int pi = sh.i, pj = sh.j;
        if (pi < 0) break;
        PP_CLK(c1)
        PP_ADD(0, c0, c1)
        PP_CNT(5)
        // ---- vote: r = cvRound(j * cos/rho + i * sin/rho) in float32, no contraction
        long long key = (long long)0x8000000000000000ull;
        if (voter) {
            const float fr = __fadd_rn(__fmul_rn((float)pj, tc), __fmul_rn((float)pi, ts));
            // a plain load + store pair, not an RMW atomic: the row belongs to this lane alone, and L2
            // keeps far more load misses in flight than atomic misses
            int32_t *bin = row + __float2int_rn(fr);
            const int val = __hip_atomic_load(bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            __hip_atomic_store(bin, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // larger value first, then the LOWER angle ("if (max_val < val)" keeps the first maximum)
            key = ((long long)val << 32) | (long long)(uint32_t)(0x7fffffff - tid);
        }
        for (int off = 32; off > 0; off >>= 1) {
            const long long o = __shfl_down(key, off);
            key = o > key ? o : key;
        }
        if (lane == 0) sh.key[wave] = (unsigned long long)key;
        __syncthreads();
        long long best = (long long)sh.key[0];
        for (int w = 1; w < 4; w++)
            if ((long long)sh.key[w] > best) best = (long long)sh.key[w];
        const int max_val = (int)(best >> 32);
        const int max_n = 0x7fffffff - (int)(uint32_t)(best & 0xffffffffll);
        PP_CLK(c2)
        PP_ADD(1, c1, c2)
        if (max_val >= a.threshold) {  // with threshold 0 false only when un-votes drove the bins negative
            const PphtWalk wk = a.walk[max_n];
            int x0 = pj, y0 = pi;
            if (wk.xflag) y0 = (y0 << 16) + (1 << 15);
            else x0 = (x0 << 16) + (1 << 15);
            const int dx = dir ? -wk.dx0 : wk.dx0, dy = dir ? -wk.dy0 : wk.dy0;

            // ---- first pass: the segment's two ends, both directions side by side
            bool on0 = false;  // this thread's position of round 0 holds a point
            for (int base = 0;; base += 128) {
                PP_CNT(6)
                const bool live = !sh.stop[dir];
                bool out = false, on = false;
                if (live) {
                    const int t = base + slot;
                    const int x = x0 + t * dx, y = y0 + t * dy;
                    const int j1 = wk.xflag ? x : x >> 16, i1 = wk.xflag ? y >> 16 : y;
                    out = j1 < 0 || j1 >= W || i1 < 0 || i1 >= H;
                    on = !out && mask_test(mask, i1, j1, TX);
                }
                if (base == 0) on0 = on;
                const unsigned long long bn = __ballot(on), bo = __ballot(out);
                if (lane == 0) {
                    sh.nzb[wave] = bn;
                    sh.oob[wave] = bo;
                }
                __syncthreads();
                if (slot == 0 && live) {  // threads 0 and 128: the sequential gap rule over this round
                    int gap = sh.gap[dir], end_t = sh.end_t[dir], stop = 0;
                    for (int w = 0; w < 2 && !stop; w++) {
                        unsigned long long n = sh.nzb[2 * dir + w];
                        const unsigned long long o = sh.oob[2 * dir + w];
                        const int valid = o ? __ffsll((long long)o) - 1 : 64;  // steps before the border
                        if (valid < 64) n &= (1ull << valid) - 1ull;
                        int prev = -1;
                        while (n) {
                            const int q = __ffsll((long long)n) - 1;
                            n &= n - 1;
                            if (gap + (q - prev - 1) > a.line_gap) {
                                stop = 1;
                                break;
                            }
                            gap = 0;
                            end_t = base + w * 64 + q;
                            prev = q;
                        }
                        if (!stop) {
                            gap += valid - prev - 1;
                            if (gap > a.line_gap || valid < 64) stop = 1;
                        }
                    }
                    sh.gap[dir] = gap;
                    sh.end_t[dir] = end_t;
                    sh.stop[dir] = stop;
                }
                __syncthreads();
                if (sh.stop[0] && sh.stop[1]) break;
            }
            PP_CLK(c3)
            PP_ADD(2, c2, c3)
            // line ends and the length test
            int ex[2], ey[2];
            for (int k = 0; k < 2; k++) {
                const int t = sh.end_t[k];
                const int kx = k ? -wk.dx0 : wk.dx0, ky = k ? -wk.dy0 : wk.dy0;
                const int x = x0 + t * kx, y = y0 + t * ky;
                ex[k] = wk.xflag ? x : x >> 16;
                ey[k] = wk.xflag ? y >> 16 : y;
            }
            const bool good = abs(ex[1] - ex[0]) >= a.line_length || abs(ey[1] - ey[0]) >= a.line_length;

            // ---- second pass: erase the segment's points; un-vote them when the segment is accepted.
            // Round 0 reuses the first pass's flags (the mask has not changed since).
            const int last = sh.end_t[dir], last_max = max(sh.end_t[0], sh.end_t[1]);
            for (int base = 0; base <= last_max; base += 128) {
                PP_CNT(7)
                const int t = base + slot;
                bool on = false;
                uint32_t ptw = 0;
                if (t <= last && !(dir == 1 && t == 0)) {  // step 0 belongs to direction 0
                    const int x = x0 + t * dx, y = y0 + t * dy;
                    const int j1 = wk.xflag ? x : x >> 16, i1 = wk.xflag ? y >> 16 : y;
                    on = base == 0 ? on0 : mask_test(mask, i1, j1, TX);
                    if (on) {
                        ptw = ((uint32_t)i1 << 16) | (uint32_t)j1;
                        mask_clear(mask, i1, j1, TX);
                    }
                }
                if (good) {  // block-uniform
                    const unsigned long long bn = __ballot(on);
                    if (lane == 0) sh.nzb[wave] = bn;
                    __syncthreads();
                    int before = __popcll(bn & ((1ull << lane) - 1ull)), np = 0;
                    for (int w = 0; w < 4; w++) {
                        const int c = __popcll(sh.nzb[w]);
                        if (w < wave) before += c;
                        np += c;
                    }
                    if (on) sh.pts[before] = (int)ptw;
                    __syncthreads();
                    if (voter) {
                        // Un-vote as load / store pairs (the row is this lane's alone), eight points at
                        // a time with all loads in flight together; points of a chunk that share a bin
                        // are merged first (the last one carries the sum), chunks follow each other in
                        // program order (same lane, same address: L2 keeps the order).
                        {
                            const int q1 = np;
                            for (int qb = 0; qb < q1; qb += 8) {
                                int bin[8], c[8], v[8];
#pragma unroll
                                for (int i = 0; i < 8; i++) {
                                    const int q = qb + i;
                                    bin[i] = -0x40000000 + i;  // distinct sentinels
                                    c[i] = 0;
                                    if (q < q1) {
                                        const uint32_t p = (uint32_t)sh.pts[q];
                                        bin[i] = __float2int_rn(__fadd_rn(__fmul_rn((float)(p & 0xffffu), tc),
                                                                          __fmul_rn((float)(p >> 16), ts)));
                                        c[i] = 1;
                                    }
                                }
#pragma unroll
                                for (int i = 1; i < 8; i++)
#pragma unroll
                                    for (int j = 0; j < i; j++)
                                        if (bin[i] == bin[j]) {
                                            c[i] += c[j];
                                            c[j] = 0;
                                        }
                                if (a.latency_mode) {
                                    // a single scan that is waited for: fire-and-forget RMW atomics -- nothing in this
                                    // lane waits for the bins' old values (the load + store pairs below expose one
                                    // memory round trip per chunk of eight points)
#pragma unroll
                                    for (int i = 0; i < 8; i++)
                                        if (c[i]) __hip_atomic_fetch_sub(row + bin[i], c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                } else {
#pragma unroll
                                    for (int i = 0; i < 8; i++)
                                        if (c[i]) v[i] = __hip_atomic_load(row + bin[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                                    for (int i = 0; i < 8; i++)
                                        if (c[i])
                                            __hip_atomic_store(row + bin[i], v[i] - c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        }
                    }
                }
            }
            PP_CLK(c4)
            PP_ADD(3, c3, c4)
            if (tid == 0 && good) {
                if (nl < a.cap) {
                    lines[4 * nl] = ex[0];
                    lines[4 * nl + 1] = ey[0];
                    lines[4 * nl + 2] = ex[1];
                    lines[4 * nl + 3] = ey[1];
                }
                nl++;
            }
        }
        PP_CLK(c5)
        __syncthreads();  // erasures are complete: wave 0 re-tests the points it still holds
        if (wave == 0 && pend) {
            const bool on = ((pend >> lane) & 1ull) && mask_test(mask, (int)(pt >> 16), (int)(pt & 0xffffu), TX);
            pend = __ballot(on);
        }
        PP_CLK(c6)
        PP_ADD(4, c5, c6)
    }
    if (tid == 0) a.n_lines[scan] = nl;
}

hipError_t launch_ppht(const PphtArgs &a, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    if (a.numangle > OMR_PPHT_THREADS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ppht_kernel, dim3(n), dim3(OMR_PPHT_THREADS), 0, s, a);
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
