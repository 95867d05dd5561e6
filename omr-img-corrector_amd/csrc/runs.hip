// runs.hip -- the run-merging ("S") sweep kernel for gfx950.
//
// Why: the gather kernels (kernels.hip) spend ~12 VALU operations per destination sample and are
// VALU-issue bound (profiles/r01_pmc_sweep.md).  For the small angles of a deskew sweep a
// destination row is a sequence of RUNS of consecutive source bits: the source row changes every
// 1/|sin t| pixels and the source column stutters every 1/(1-cos t) pixels.  This kernel builds 32
// destination pixels at once:
//     bit i of word (r, w) = src[Rb + gy(i)][Xb + i - st(i)],
//     gy(i) = (fy + cb(i)) >> 10,  st(i) = -((fx - ea(i)) >> 10),   ea(i) = 1024 i - ca(i)
// where (Xb, fx) / (Rb, fy) are the integer / 10-bit fraction parts of the word's first sample and
// ca, cb are the column tables relative to the word start.  gy and st depend only on
// (candidate, word, fraction): the plan ENUMERATES all 1024 fractions per (candidate, word) from the
// integer tables of OpenCV's warpAffine (no threshold arithmetic, so bit-exact by construction) and
// stores the <= 33 distinct mask tuples (RunTab).  Per word the kernel then needs two byte look-ups,
// one 32-bit unaligned window per source row (two LDS dwords + v_alignbit), and one and/or per row.
//
// Round-3 structure: nothing on the way into LDS passes through registers and
// the waves of a workgroup run free.  The scan's bit image is kept TRANSPOSED in HBM (word column x =
// rows contiguous dwords); every WAVE owns the source window of its 64 destination rows, column-major
// in LDS, filled by `buffer_load_dwordx4 ... lds` (whole cache lines, no bank conflicts for lane = row
// at any window height) and waited for with the wave's own vmcnt -- no workgroup barrier per band.
// The run tables of the next word group arrive by LDS-DMA while the current group is swept (two table
// sets).  A workgroup walks a chunk of word groups of one candidate, so the row counts stay in LDS
// until the chunk is done; the waves meet only when a group's column counters are reduced.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace omr {

#ifndef RUN_PRIO_REST
#define RUN_PRIO_REST 0
#endif
#define RUN_K OMR_RUN_K    // destination words per word group (kernels.hpp)
#define RUN_BAND 512       // destination rows per band (8 waves x 64 lanes)
#define RUN_WAVES 8
#define RUN_WCOLS 7        // a wave's window: word columns ...
#define RUN_WROWS 112      // ... of this many rows each (column-major, 16-byte pieces of 4 rows)
#define RUN_WPIECES (RUN_WROWS / 4)
#define RUN_WIN_BYTES (RUN_WCOLS * RUN_WROWS * 4)
#define RUN_TAB_BYTES 3648
#define RUN_TUPHI_OFS 640
#define RUN_TUPX_OFS 1280
#define RUN_IDXY_OFS 1600
#define RUN_IDXX_OFS 2624
#define RUN_TABSET_BYTES (RUN_K * RUN_TAB_BYTES)
// LDS map.  The two table sets start at multiples of 1024 (a set's base is added under the 10-bit
// fraction / the shifted tuple id, the word's offset is the ds_read immediate).
#define RUN_TAB0_OFS 0
#define RUN_META_OFS RUN_TABSET_BYTES                       // [2 sets][RUN_K] (ca0, cb0)
#define RUN_TAB1_OFS 15360
#define RUN_WIN_OFS (RUN_TAB1_OFS + RUN_TABSET_BYTES)       // 29952: one window per wave
#define RUN_PARK_OFS (RUN_WIN_OFS + RUN_WAVES * RUN_WIN_BYTES)  // column counters of a flush: [word][plane][wave][32 pairs]
#define RUN_PARK_BYTES (RUN_K * 4 * 256 * 4)
#define RUN_HROW_OFS (RUN_PARK_OFS + RUN_PARK_BYTES)        // row counts of the chunk, two u16 per dword
#define RUN_GEO_OFS (RUN_HROW_OFS + OMR_RUN_MAX_ROWS * 2)   // word-group constants of the chunk
#define RUN_LDS_BYTES (RUN_GEO_OFS + OMR_RUN_GC * 32)
static_assert(RUN_K == 4, "one (word, 16-column half) per wave in the column reduction");
static_assert(RUN_META_OFS + 2 * RUN_K * 8 <= RUN_TAB1_OFS, "meta fits the gap between the table sets");
static_assert(RUN_TAB1_OFS % 1024 == 0 && RUN_TAB0_OFS % 1024 == 0, "table sets are 1024-aligned");
static_assert(2 * RUN_LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(RUN_WROWS % 4 == 0 && 2 * RUN_WPIECES <= 64, "two window columns per wave-instruction");
static_assert(RUN_WCOLS == 7 && RUN_WROWS == 112, "rungeo_kernel sizes the windows with the same numbers");

static_assert(sizeof(RunTab) == RUN_TAB_BYTES, "RunTab layout");
static_assert(sizeof(RunMeta) == 32, "RunMeta layout");
static_assert(sizeof(RunBlk) == 32, "RunBlk layout");

// ------------------------------------------------------------------------------------------
// RunTab builder: block = one (candidate, word), thread = one 10-bit fraction f.
__device__ __forceinline__ int block_dedupe_1024(bool change, int *s_wave)
{
    // exclusive count of `change` flags before this thread, over a 1024-thread block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(change);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; k++) base += s_wave[k];
    return base + before;
}

// Words NW .. NWp-1 pad the last word group: copies of the last real word (every lane address stays
// meaningful) whose level masks are EMPTY, so they add nothing to any count.
__global__ __launch_bounds__(1024) void runtab_kernel(const int32_t *__restrict__ CA, const int32_t *__restrict__ CB,
                                                      int NC, int NW, int NWp, RunTab *__restrict__ tabs,
                                                      RunMeta *__restrict__ meta, int2_t *__restrict__ metac)
{
    __shared__ int s_ca[32], s_cb[32];
    __shared__ uint32_t s_tup[1024][9];  // padded: conflict-free row compare
    __shared__ int s_wave[16];
    __shared__ int s_bad, s_smax;
    const int wp = blockIdx.x, a = blockIdx.y, w = min(wp, NW - 1), c0 = w * 32, f = threadIdx.x;
    const bool real = wp < NW;
    const int32_t *ca_row = CA + (int64_t)a * NC, *cb_row = CB + (int64_t)a * NC;
    if (f < 32) {
        // columns past the end of the row (last word only) are don't-cares: their destination bits are
        // masked out of the level masks below.  Give them lag 0 and the row of the last real column so
        // they add neither a level nor a validity failure.
        const int c = min(c0 + f, NC - 1);
        s_ca[f] = (c0 + f < NC) ? ca_row[c] - ca_row[c0] : 1024 * f;
        s_cb[f] = cb_row[c] - cb_row[c0];
    }
    if (f == 0) {
        s_bad = 0;
        s_smax = 0;
    }
    __syncthreads();
    int cbmin = 0, cbmax = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        cbmin = min(cbmin, s_cb[i]);
        cbmax = max(cbmax, s_cb[i]);
    }
    bool bad = false;
    int base_off = 0, nlev = 1;
    if (cbmin >= 0) {
        nlev = ((1023 + cbmax) >> 10) + 1;
    } else if (cbmax <= 0) {
        base_off = cbmin >> 10;  // floor: the lowest level any fraction can reach
        nlev = 1 - base_off;
    } else {
        bad = true;
    }
    if (nlev > 8) {
        bad = true;
        nlev = 8;
    }
    // ---- row-level masks of this fraction
    uint32_t my[8];
#pragma unroll
    for (int lv = 0; lv < 8; lv++) my[lv] = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const int lv = ((f + s_cb[i]) >> 10) - base_off;
        if (lv < 0 || lv > 7) bad = true;
#pragma unroll
        for (int k = 0; k < 8; k++) my[k] |= (lv == k ? 1u : 0u) << i;
    }
#pragma unroll
    for (int lv = 0; lv < 8; lv++) s_tup[f][lv] = my[lv];
    __syncthreads();
    bool change = f == 0;
    if (f > 0) {
#pragma unroll
        for (int lv = 0; lv < 8; lv++) change |= s_tup[f - 1][lv] != my[lv];
    }
    int idy = block_dedupe_1024(change, s_wave) + (change ? 1 : 0) - 1;  // inclusive count - 1
    RunTab *T = tabs + ((int64_t)a * NWp + wp);
    // destination bits that exist (last word of a row; none in a padding word): folded into the level
    // masks, so the sweep kernel needs no per-word validity mask (the lag merge happens before the
    // level select)
    const uint32_t valid = !real ? 0u : ((c0 + 32 <= NC) ? 0xffffffffu : ((1u << (NC - c0)) - 1u));
    if (idy >= OMR_RUN_TUPLES) {
        bad = true;
        idy = 0;
    } else if (change) {
#pragma unroll
        for (int lv = 0; lv < 4; lv++) {
            T->tupYlo[idy][lv] = my[lv] & valid;
            T->tupYhi[idy][lv] = my[lv + 4] & valid;
        }
    }
    T->idxY[f] = (uint8_t)idy;
    // ---- column-lag masks of this fraction: st(i) = -((f - ea(i)) >> 10), ea(i) = 1024 i - ca(i)
    uint32_t sx1 = 0, sx2 = 0;
    int smax = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const int ea = 1024 * i - s_ca[i];
        const int st = -((f - ea) >> 10);
        if (st < 0 || st > 2 || st > i) bad = true;
        sx1 |= (st == 1 ? 1u : 0u) << i;
        sx2 |= (st == 2 ? 1u : 0u) << i;
        smax = max(smax, st);
    }
    __syncthreads();
    s_tup[f][0] = sx1;
    s_tup[f][1] = sx2;
    __syncthreads();
    change = f == 0 || s_tup[f - 1][0] != sx1 || s_tup[f - 1][1] != sx2;
    int idx = block_dedupe_1024(change, s_wave) + (change ? 1 : 0) - 1;
    if (idx >= OMR_RUN_TUPLES) {
        bad = true;
        idx = 0;
    } else if (change) {
        T->tupX[idx][0] = sx1;
        T->tupX[idx][1] = sx2;
    }
    T->idxX[f] = (uint8_t)idx;
    if (bad) s_bad = 1;
    if (smax > 0) atomicMax(&s_smax, min(smax, 2));
    __syncthreads();
    if (f == 0) {
        RunMeta m;
        m.ca0 = ca_row[c0];
        m.cb0 = cb_row[c0] + base_off * 1024;
        m.nlev = nlev;
        m.smax = s_smax;
        m.valid = valid;
        m.ok = s_bad ? 0 : 1;
        m.pad0 = m.pad1 = 0;
        if (real) meta[(int64_t)a * NW + w] = m;
        metac[(int64_t)a * NWp + wp] = int2_t{m.ca0, m.cb0};
    }
}

// RunBlk of every (candidate, word group): thread = one group
__global__ __launch_bounds__(256) void runblk_kernel(const int32_t *__restrict__ CA, const int32_t *__restrict__ CB,
                                                     int A, int NC, int NW, int G,
                                                     const RunMeta *__restrict__ meta, RunBlk *__restrict__ blk)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A * G) return;
    const int a = i / G, g = i - a * G;
    const int w0 = g * RUN_K, kw = min(RUN_K, NW - w0);
    const RunMeta *mt = meta + ((int64_t)a * NW + w0);
    RunBlk b;
    b.nlev = 1;
    b.smax = 0;
    b.lagbits = 0;
    for (int k = 0; k < kw; k++) {
        b.nlev = max(b.nlev, mt[k].nlev);
        b.smax = max(b.smax, mt[k].smax);
        b.lagbits |= mt[k].smax << (2 * (k >> 1));  // smax is 0, 1 or 2: OR of a pair = its maximum ...
    }
    for (int k = 0; k < RUN_K; k += 2)              // ... except 1 | 2 = 3, which means 2
        if (((b.lagbits >> k) & 3) == 3) b.lagbits &= ~(1 << k);
    b.valid_last = mt[kw - 1].valid;
    const int c_first = w0 * 32, c_last = min(NC, (w0 + kw) * 32) - 1;
    const int ca_f = CA[(int64_t)a * NC + c_first], ca_l = CA[(int64_t)a * NC + c_last];
    const int cb_f = CB[(int64_t)a * NC + c_first], cb_l = CB[(int64_t)a * NC + c_last];
    b.ca_min = min(ca_f, ca_l);
    b.ca_max = max(ca_f, ca_l);
    b.cb_min = min(cb_f, cb_l);
    b.cb_max = max(cb_f, cb_l);
    blk[i] = b;
}

hipError_t launch_runtab(const int32_t *d_CA, const int32_t *d_CB, int A, int NC, int NW, RunTab *d_tabs,
                         RunMeta *d_meta, int2_t *d_metac, RunBlk *d_blk, hipStream_t s)
{
    const int G = (NW + RUN_K - 1) / RUN_K;
    hipLaunchKernelGGL(runtab_kernel, dim3(G * RUN_K, A), dim3(1024), 0, s, d_CA, d_CB, NC, NW, G * RUN_K, d_tabs, d_meta,
                       d_metac);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(runblk_kernel, dim3((A * G + 255) / 256), dim3(256), 0, s, d_CA, d_CB, A, NC, NW, G, d_meta,
                       d_blk);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Window origins of the sweep kernel, once per plan: thread = one (candidate, word group, band, wave).
// A wave sweeps 64 destination rows x one word group per band; the source bounding box of that patch (the
// map is monotone in r and in c, so the four corner samples bound it) fixes the first word column and the
// first row (a multiple of 4: the window travels in 16-byte pieces of 4 rows) of its LDS window.  An entry
// whose box does not fit the window is marked (the candidate then goes to the gather kernel); ext[] collects
// how far the windows reach beyond the image, which sizes the zero guard around the transposed bit image.
#define RUN_WCOLS_ 7
#define RUN_WROWS_ 112
__global__ __launch_bounds__(256) void rungeo_kernel(const int2_t *__restrict__ RT, const RunBlk *__restrict__ blk, int A, int G,
                                                     int NR, int NBt, int2_t *__restrict__ wgeo, int32_t *__restrict__ ext)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A * G * NBt * 8) return;
    const int wave = i & 7, band = (i >> 3) % NBt, ag = (i >> 3) / NBt, a = ag / G;
    const int r0 = min(band * 512 + wave * 64, NR - 1), r1 = min(band * 512 + wave * 64 + 63, NR - 1);
    const int2_t t0 = RT[(int64_t)a * NR + r0], t1 = RT[(int64_t)a * NR + r1];
    const RunBlk b = blk[ag];
    const int minbit = (min(t0.x, t1.x) + b.ca_min) >> 10;  // min over the four corners
    const int maxbit = (max(t0.x, t1.x) + b.ca_max) >> 10;
    const int minrow = (min(t0.y, t1.y) + b.cb_min) >> 10;
    const int maxrow = (max(t0.y, t1.y) + b.cb_max) >> 10;
    int2_t q;
    q.x = minbit >> 5;
    q.y = (minrow - 7) & ~3;  // a word reads up to 7 rows beside its true samples; pieces are 4 rows
    // every selected sample lies in word columns q.x .. maxbit >> 5; the column after a sample's is read too
    // but none of its bits is ever selected: it only has to exist.  (Fetching only the pieces and columns that
    // hold samples -- a third fewer bytes -- was measured: 2 % SLOWER, the lane masks cost more VALU cycles than
    // the bytes save; the sweep is bound by VALU cycles)
    const bool fits = (maxbit >> 5) - q.x + 2 <= RUN_WCOLS_ && maxrow + 8 - q.y <= RUN_WROWS_ && q.x > -(1 << 20) &&
                      q.x < (1 << 20) && q.y > -(1 << 20) && q.y < (1 << 20);
    if (fits) {
        atomicMin(&ext[0], q.x);
        atomicMax(&ext[1], q.x + RUN_WCOLS_);
        atomicMin(&ext[2], q.y);
        atomicMax(&ext[3], q.y + RUN_WROWS_);
    } else {
        q.x = 0x7fffffff;
    }
    wgeo[i] = q;
}

hipError_t launch_rungeo(const int2_t *d_RT, const RunBlk *d_blk, int A, int G, int NR, int2_t *d_wgeo, int32_t *d_ext,
                         hipStream_t s)
{
    const int NBt = (NR + 511) / 512, n = A * G * NBt * 8;
    hipLaunchKernelGGL(rungeo_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_RT, d_blk, A, G, NR, NBt, d_wgeo, d_ext);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Bit image -> transposed bit image with a zero guard: T[GX + x][GY + y] = word x of row y, rowsT dwords per
// word column.  32 x 32 word tiles through LDS: 128-byte reads and 128-byte writes.  The guard (GX word
// columns left and right, GY rows above and below, zeroed once when the buffer is made) holds every window
// origin of the plan, so the sweep fetches its windows without a range test.
__global__ __launch_bounds__(256) void transpose_bits_kernel(const uint32_t *__restrict__ bits, int rows, int wpr,
                                                             uint32_t *__restrict__ T, int NW, int NWt, int rowsT, int GX, int GY)
{
    __shared__ uint32_t tile[32][33];
    bits += (int64_t)blockIdx.z * rows * wpr;  // blockIdx.z = scan of the launch
    T += (int64_t)blockIdx.z * NWt * rowsT;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int y = y0 + ty + 8 * j, x = x0 + tx;
        tile[ty + 8 * j][tx] = (y < rows && x < wpr) ? bits[(int64_t)y * wpr + x] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + ty + 8 * j, y = y0 + tx;
        if (x < NW && y < rows) T[(int64_t)(GX + x) * rowsT + GY + y] = tile[tx][ty + 8 * j];
    }
}

hipError_t launch_transpose_bits(const uint32_t *d_bits, int rows, int wpr, uint32_t *d_T, int NW, int NWt, int rowsT, int GX,
                                 int GY, hipStream_t s, int scans)
{
    hipLaunchKernelGGL(transpose_bits_kernel, dim3((NW + 31) / 32, (rows + 31) / 32, scans), dim3(256), 0, s, d_bits, rows,
                       wpr, d_T, NW, NWt, rowsT, GX, GY);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The sweep kernel.  Workgroup = (scan, chunk of word groups, candidate); it walks the chunk's word
// groups (RUN_K words = 128 destination columns each) one after the other and, inside a group, ALL
// destination rows in bands of 512 (8 waves x 64 lanes, lane = destination row, every lane builds the
// group's RUN_K words of its row).
//   row counts   : popcount of the lane's words, added to the chunk's per-row count in LDS (two u16 per
//                  dword); written once per workgroup as a u16 partial per (row, chunk)
//   column counts: every lane keeps a 3-plane bit-sliced counter per word (its rows of up to 7 bands);
//                  at the end of a group lane pairs add their counters (DPP), park them in their own LDS
//                  region (RUN_PARK_OFS -- NOT in the wave's window buffer: the next window's LDS-DMA has
//                  already been issued into that buffer when flush_columns() runs, so reusing it would race
//                  with the DMA), and wave w sums the 256 parked numbers of (word w/2,
//                  column half w%2) lane-wise, then over its 64 lanes with one v_add_co_u32 (shift +
//                  carry-out = ballot of the top bit) and one s_bcnt1 per bit.  Plain stores, no atomics.
// LDS accesses of the inner loop take INTEGER byte addresses (the dynamic segment starts at LDS address
// 0, checked once per block): table offsets then fold into the ds_read immediates.
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v2u32 lds_cu2;
typedef __attribute__((address_space(3))) const v4u32 lds_cu4;
__device__ __forceinline__ uint32_t ldsr_u8(uint32_t a) { return *(lds_cu8 *)(uintptr_t)a; }
__device__ __forceinline__ uint32_t ldsr_u32(uint32_t a) { return *(lds_cu32 *)(uintptr_t)a; }
__device__ __forceinline__ uint2 ldsr_u2(uint32_t a)
{
    const v2u32 v = *(lds_cu2 *)(uintptr_t)a;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 ldsr_u4(uint32_t a)
{
    const v4u32 v = *(lds_cu4 *)(uintptr_t)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ldsr_i4(uint32_t a)
{
    const v4u32 v = *(lds_cu4 *)(uintptr_t)a;
    return make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
}
__device__ __forceinline__ void ldsw_u32(uint32_t a, uint32_t v) { *(lds_u32 *)(uintptr_t)a = v; }

// ---- LDS-DMA.  `buffer_load_dword(x4) ... lds` writes M0 + 4 (16) * lane: 64 consecutive dwords (pieces)
// per wave-instruction, each from the lane's own buffer offset; a lane whose offset is outside
// [0, num_records) lands as zeros (tools/lds_dma_window.hip checks both on the hardware).  Issued as
// inline assembly: the compiler then neither tracks them in its vmcnt bookkeeping (it would wait for
// them before the first LDS read that follows) nor needs to -- the band loop waits with an explicit
// `s_waitcnt vmcnt(0)` at the one place where their data is needed.
// raw buffer descriptor (stride 0, no swizzle, DATA_FORMAT 32)
__device__ __forceinline__ v4u32 make_rsrc(const void *base, uint32_t nbytes)
{
    const uint64_t b = (uint64_t)base;
    v4u32 r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(nbytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void dma_b32(const v4u32 rsrc, uint32_t voff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
}
__device__ __forceinline__ void dma_b128(const v4u32 rsrc, uint32_t voff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
}

template <int LANE>
__device__ __forceinline__ uint32_t write_lane_imm(uint32_t vreg, uint32_t value)
{
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vreg) : "s"(value), "n"(LANE));
    return vreg;
}

// Six bit planes of a bit-sliced per-lane number: shift every plane left by one (v_add_co_u32: the
// carry-out mask IS the ballot of the old top bit), count the carries (s_bcnt1) and return
// sum_j count_j << j, i.e. the sum over the 64 lanes of the column held in the top bit.  The six
// VALU adds issue back to back, so their VALU->SGPR latency is overlapped.
__device__ __forceinline__ uint32_t shl1_sum6(uint32_t &p0, uint32_t &p1, uint32_t &p2, uint32_t &p3, uint32_t &p4,
                                              uint32_t &p5)
{
    uint32_t n, t;
    unsigned long long m0, m1, m2, m3, m4, m5;
    asm("v_add_co_u32_e64 %0, %8, %0, %0\n\t"
        "v_add_co_u32_e64 %1, %9, %1, %1\n\t"
        "v_add_co_u32_e64 %2, %10, %2, %2\n\t"
        "v_add_co_u32_e64 %3, %11, %3, %3\n\t"
        "v_add_co_u32_e64 %4, %12, %4, %4\n\t"
        "v_add_co_u32_e64 %5, %13, %5, %5\n\t"
        "s_bcnt1_i32_b64 %6, %13\n\t"
        "s_bcnt1_i32_b64 %7, %12\n\t"
        "s_lshl1_add_u32 %6, %6, %7\n\t"
        "s_bcnt1_i32_b64 %7, %11\n\t"
        "s_lshl1_add_u32 %6, %6, %7\n\t"
        "s_bcnt1_i32_b64 %7, %10\n\t"
        "s_lshl1_add_u32 %6, %6, %7\n\t"
        "s_bcnt1_i32_b64 %7, %9\n\t"
        "s_lshl1_add_u32 %6, %6, %7\n\t"
        "s_bcnt1_i32_b64 %7, %8\n\t"
        "s_lshl1_add_u32 %6, %6, %7"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "=&s"(n), "=&s"(t), "=&s"(m0), "=&s"(m1),
          "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5)
        :
        : "scc");
    return n;
}

// Issue costs that shaped this loop (tools/valu_ops.hip on MI355X, >= 2 waves per SIMD; profiles/r02_valu_issue.md):
// only the plain two-VGPR-source VOP2 ops (v_add/sub, v_and/or/xor, v_lshrrev, v_ashrrev, v_mov) issue in
// 2 cycles per wave64; every VOP3 op (v_alignbit, v_bfi, v_and_or, v_lshl_add, v_mad_u32_u24, v_bcnt, v_bfe),
// v_lshlrev_b32, v_mul_u32_u24 and any VOP2 with an SGPR source take 4; v_cndmask on an SGPR mask far more.

// one word: NLEV unaligned 32-bit windows -> destination word.  Four VALU operations per level when the
// word has a column lag (v_alignbit, v_lshlrev, v_bfi, v_and_or): spelled out as instructions because the
// compiler's three-input folding (v_bitop3) settles on five and a half.
__device__ __forceinline__ uint32_t v_bfi(uint32_t mask, uint32_t a, uint32_t b)  // (mask & a) | (~mask & b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t v_and_or(uint32_t a, uint32_t b, uint32_t c)  // (a & b) | c
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int NLEV, int SMAX>
__device__ __forceinline__ uint32_t merge_word(const uint2 (&d)[NLEV], const uint32_t sh, const uint4 s03, const uint4 s47,
                                               const uint2 sx)
{
    const uint32_t sel[8] = {s03.x, s03.y, s03.z, s03.w, s47.x, s47.y, s47.z, s47.w};
    uint32_t D = 0;
#pragma unroll
    for (int lv = 0; lv < NLEV; lv++) {
        const uint32_t W = __builtin_amdgcn_alignbit(d[lv].y, d[lv].x, sh);  // uses sh & 31
        uint32_t U = W;
        if (SMAX >= 1) U = v_bfi(sx.x, W << 1, U);
        if (SMAX >= 2) U = v_bfi(sx.y, W << 2, U);
        D = lv == 0 ? (U & sel[0]) : v_and_or(U, sel[lv], D);
    }
    return D;
}

// bit-sliced add of a 1-bit value per column into a 3-plane counter: five 2-cycle ops
__device__ __forceinline__ void count_columns(uint32_t &c0, uint32_t &c1, uint32_t &c2, const uint32_t D)
{
    // t = c0 & D; c0 ^= D; c2 |= c1 & t; c1 ^= t -- spelled out so that the counters stay IN PLACE: the compiler's form
    // renames them and copies them back at the loop edge (four v_mov_b64 per band)
    uint32_t t;
    asm("v_and_b32 %3, %0, %4\n\t"
        "v_xor_b32 %0, %0, %4\n\t"
        "v_and_or_b32 %2, %1, %3, %2\n\t"
        "v_xor_b32 %1, %1, %3"
        : "+v"(c0), "+v"(c1), "+v"(c2), "=&v"(t)
        : "v"(D));
}

// One pair of words of one band for one wave.  Both words are in flight together: both fraction
// look-ups and all 2 x NLEV window reads are issued before anything is consumed (LDS latency hiding at
// 4 waves per SIMD).  Padding words were built with empty level masks, the last word's masks carry the
// row-end mask (runtab_kernel), rows past the end of the image are EXEC-masked by the caller: nothing
// to mask here.  `tabv` is the byte address of the group's table set (a multiple of 1024); (ca0, cb0) of
// the group's words sit in registers; the window's base and first row are folded into (rx, ry) by the
// caller.
struct RunWordK {
    int ca0, cb0;
};
// the row-fraction look-ups of a pair (the head of its chain of dependent LDS reads: look-up -> level masks -> merge)
template <int K>
__device__ __forceinline__ uint2 pair_lookup_y(const int ry, const uint32_t tabv, const RunWordK (&wk)[RUN_K])
{
    constexpr uint32_t ta = K * RUN_TAB_BYTES, tb = ta + RUN_TAB_BYTES;
    const int B0a = ry + wk[K].cb0, B0b = ry + wk[K + 1].cb0;
    return make_uint2(ldsr_u8((((uint32_t)B0a & 1023u) | tabv) + (ta + RUN_IDXY_OFS)),
                      ldsr_u8((((uint32_t)B0b & 1023u) | tabv) + (tb + RUN_IDXY_OFS)));
}
template <int NLEV, int SMAX, int K>
__device__ __forceinline__ void pair_words(const int rx, const int ry, const uint32_t tabv, const RunWordK (&wk)[RUN_K],
                                           uint32_t (&D)[RUN_K], const uint2 idy)
{
    constexpr int k = K;
    constexpr uint32_t ta = k * RUN_TAB_BYTES, tb = ta + RUN_TAB_BYTES;
    const int A0a = rx + wk[k].ca0, B0a = ry + wk[k].cb0, A0b = rx + wk[k + 1].ca0, B0b = ry + wk[k + 1].cb0;
    uint32_t Da, Db;
    const uint32_t idya = idy.x, idyb = idy.y;
    uint32_t idxa = 0, idxb = 0;
    if (SMAX > 0) {
        idxa = ldsr_u8((((uint32_t)A0a & 1023u) | tabv) + (ta + RUN_IDXX_OFS));
        idxb = ldsr_u8((((uint32_t)A0b & 1023u) | tabv) + (tb + RUN_IDXX_OFS));
    }
    // window byte address: (word column * RUN_WROWS + row) * 4, column-major; window-local coordinates
    // are >= 0 and the window's base rides in ry
    const uint32_t addra = (uint32_t)__mul24(A0a >> 15, RUN_WROWS * 4) + (((uint32_t)B0a >> 8) & ~3u);
    const uint32_t addrb = (uint32_t)__mul24(A0b >> 15, RUN_WROWS * 4) + (((uint32_t)B0b >> 8) & ~3u);
    uint2 wa[NLEV], wb[NLEV];
#pragma unroll
    for (int lv = 0; lv < NLEV; lv++) {  // the same rows of two adjacent word columns
        wa[lv].x = ldsr_u32(addra + lv * 4);
        wa[lv].y = ldsr_u32(addra + lv * 4 + RUN_WROWS * 4);
    }
#pragma unroll
    for (int lv = 0; lv < NLEV; lv++) {
        wb[lv].x = ldsr_u32(addrb + lv * 4);
        wb[lv].y = ldsr_u32(addrb + lv * 4 + RUN_WROWS * 4);
    }
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const uint32_t tya = (idya << 4) | tabv, tyb = (idyb << 4) | tabv;
    const uint4 sa03 = ldsr_u4(tya + ta), sa47 = NLEV > 4 ? ldsr_u4(tya + (ta + RUN_TUPHI_OFS)) : z4;
    const uint4 sb03 = ldsr_u4(tyb + tb), sb47 = NLEV > 4 ? ldsr_u4(tyb + (tb + RUN_TUPHI_OFS)) : z4;
    uint2 sxa = make_uint2(0, 0), sxb = make_uint2(0, 0);
    if (SMAX > 0) {
        sxa = ldsr_u2(((idxa << 3) | tabv) + (ta + RUN_TUPX_OFS));
        sxb = ldsr_u2(((idxb << 3) | tabv) + (tb + RUN_TUPX_OFS));
    }
    Da = merge_word<NLEV, SMAX>(wa, (uint32_t)A0a >> 10, sa03, sa47, sxa);
    Db = merge_word<NLEV, SMAX>(wb, (uint32_t)A0b >> 10, sb03, sb47, sxb);
    D[k] = Da;
    D[k + 1] = Db;
}

// A pair's column lag bound (0, 1 or 2; wave-uniform) picks its specialisation: most words of a
// candidate need no lag handling even when some word of the group does.
template <int NLEV, int K>
__device__ __forceinline__ void pair_dispatch(const int rx, const int ry, const uint32_t tabv, const RunWordK (&wk)[RUN_K],
                                              uint32_t (&D)[RUN_K], const int lagbits, const uint2 idy)
{
    // keep the pairs apart: without this fence the scheduler hoists both pairs' loads
    __builtin_amdgcn_sched_barrier(0);
    const int lag = (lagbits >> K) & 3;
    if (lag == 0) return pair_words<NLEV, 0, K>(rx, ry, tabv, wk, D, idy);
    if (lag == 1) return pair_words<NLEV, 1, K>(rx, ry, tabv, wk, D, idy);
    return pair_words<NLEV, 2, K>(rx, ry, tabv, wk, D, idy);
}

// The RUN_K words of one band for one wave.  NLEV is uniform for the whole group (maximum over its
// words; unused levels have empty masks), so the wave dispatches once per band to a straight-line
// specialisation.
template <int NLEV>
__device__ __forceinline__ void band_words(const int rx, const int ry, const uint32_t tabv, const RunWordK (&wk)[RUN_K],
                                           uint32_t (&D)[RUN_K], const int lagbits)
{
    // the second pair's look-ups travel while the first pair is merged (two registers held across it): its chain of
    // dependent LDS reads is then one round trip shorter
    const uint2 idy0 = pair_lookup_y<0>(ry, tabv, wk), idy2 = pair_lookup_y<2>(ry, tabv, wk);
    pair_dispatch<NLEV, 0>(rx, ry, tabv, wk, D, lagbits, idy0);
    pair_dispatch<NLEV, 2>(rx, ry, tabv, wk, D, lagbits, idy2);
}

// The specialisations only BUILD the band's four destination words; the column counters and the row count are
// updated once, after the branches have met: with the counters inside them every join of the dispatch copied the
// twelve counter registers around (ten v_mov_b64 per band).
__device__ __forceinline__ uint32_t band_words_s(const int rx, const int ry, const uint32_t tabv, const RunWordK (&wk)[RUN_K],
                                                 uint32_t (&c0)[RUN_K], uint32_t (&c1)[RUN_K], uint32_t (&c2)[RUN_K],
                                                 const int nlev, const int lagbits)
{
    uint32_t D[RUN_K];
    switch (nlev) {
    case 1: band_words<1>(rx, ry, tabv, wk, D, lagbits); break;
    case 2: band_words<2>(rx, ry, tabv, wk, D, lagbits); break;
    case 3: band_words<3>(rx, ry, tabv, wk, D, lagbits); break;
    case 4: band_words<4>(rx, ry, tabv, wk, D, lagbits); break;
    case 5: band_words<5>(rx, ry, tabv, wk, D, lagbits); break;
    case 6: band_words<6>(rx, ry, tabv, wk, D, lagbits); break;
    case 7: band_words<7>(rx, ry, tabv, wk, D, lagbits); break;
    default: band_words<8>(rx, ry, tabv, wk, D, lagbits); break;
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < RUN_K; k++) {
        count_columns(c0[k], c1[k], c2[k], D[k]);
        cnt += __popc(D[k]);
    }
    return cnt;
}

// bit-sliced a += b for two NP-plane numbers; a grows to NP + 1 planes
template <int NP>
__device__ __forceinline__ void bs_add(uint32_t (&a)[6], const uint32_t (&b)[6])
{
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint32_t t = a[i] ^ b[i];
        const uint32_t g = a[i] & b[i];
        a[i] = t ^ c;
        c = g | (c & t);
    }
    a[NP] = c;
}

// Column counts of one word group.  Every lane holds a 3-plane bit-sliced counter per word (its rows of
// the last <= 7 bands).  Lane pairs add their counters (4 planes, DPP), the even lane parks the sum in
// LDS: [word][plane][wave][32 pairs].  After the barrier wave w owns (word w / 2, columns
// 16 (1 - w % 2) .. + 15): lanes 0-31 add the parked numbers of waves 0-3, lanes 32-63 those of waves 4-7
// (6 planes), the 64 lanes are summed with shl1_sum6 -- one column per step, most significant bit first --
// and the 16 column totals are added to lanes 0-15 of `acc`.
// (Deliberately NOT inlined: its scalar totals and asm temporaries would otherwise raise the register
// pressure of the band loop.)
__device__ __noinline__ uint32_t reduce_columns(const int tid, uint32_t acc)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int k = wave >> 1, hh = wave & 1;
    const uint32_t base = RUN_PARK_OFS + (uint32_t)(((k * 4) * 256 + (lane >> 5) * 128 + (lane & 31)) * 4);
    uint32_t x[4][6];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int pl = 0; pl < 4; pl++) x[j][pl] = ldsr_u32(base + (uint32_t)((pl * 256 + j * 32) * 4));
        x[j][4] = x[j][5] = 0;
    }
    bs_add<4>(x[0], x[1]);
    bs_add<4>(x[2], x[3]);
    bs_add<5>(x[0], x[2]);
    uint32_t s0 = x[0][0], s1 = x[0][1], s2 = x[0][2], s3 = x[0][3], s4 = x[0][4], s5 = x[0][5];
    if (hh) {  // wave-uniform: this wave takes columns 15..0
        s0 <<= 16, s1 <<= 16, s2 <<= 16, s3 <<= 16, s4 <<= 16, s5 <<= 16;
    }
    uint32_t tot[16];
#pragma unroll
    for (int b = 15; b >= 0; b--) tot[b] = shl1_sum6(s0, s1, s2, s3, s4, s5);
    uint32_t v = 0;
#define WL(B) v = write_lane_imm<B>(v, tot[B]);
    WL(0) WL(1) WL(2) WL(3) WL(4) WL(5) WL(6) WL(7) WL(8) WL(9) WL(10) WL(11) WL(12) WL(13) WL(14) WL(15)
#undef WL
    return acc + v;  // lanes 16-63 add 0
}

__device__ __forceinline__ uint32_t dpp_pair_swap(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// The waves run free between flushes, so two barriers frame the parking: the first says that every wave
// has finished the PREVIOUS reduction (its reads of the park) and this group's sweep, the second that
// every wave has parked.
__device__ __forceinline__ uint32_t flush_columns(uint32_t (&c0)[RUN_K], uint32_t (&c1)[RUN_K], uint32_t (&c2)[RUN_K],
                                                  const int tid, uint32_t acc)
{
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t mine = RUN_PARK_OFS + (uint32_t)((wave * 32 + (lane >> 1)) * 4);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RUN_K; k++) {
        // this lane's 3-plane counter + its neighbour's -> 4 planes (the same in both lanes of the pair)
        const uint32_t b0 = dpp_pair_swap(c0[k]), b1 = dpp_pair_swap(c1[k]), b2 = dpp_pair_swap(c2[k]);
        const uint32_t t0 = c0[k] ^ b0, g0 = c0[k] & b0;
        const uint32_t t1 = c1[k] ^ b1, g1 = (c1[k] & b1) | (g0 & t1);
        const uint32_t t2 = c2[k] ^ b2, g2 = (c2[k] & b2) | (g1 & t2);
        if ((lane & 1) == 0) {
            ldsw_u32(mine + (uint32_t)(((k * 4 + 0) * 256) * 4), t0);
            ldsw_u32(mine + (uint32_t)(((k * 4 + 1) * 256) * 4), t1 ^ g0);
            ldsw_u32(mine + (uint32_t)(((k * 4 + 2) * 256) * 4), t2 ^ g1);
            ldsw_u32(mine + (uint32_t)(((k * 4 + 3) * 256) * 4), g2);
        }
        c0[k] = c1[k] = c2[k] = 0;
    }
    __syncthreads();
    return reduce_columns(tid, acc);
}

#define RUN_FLUSH_BANDS 7  // 3-plane counters hold up to 7 rows per lane

struct RunGeom {  // source window of one wave and band (wave-uniform)
    int wxw;      // first word column
    int wy0;      // first row, a multiple of 4
    bool fits;
};

// Diagnostic build only (make debug, -DOMR_RUNS_DEBUG): per-block phase clocks of wave 0, summed into a global
// array [prologue, wait for the window, compute, issue of the next window, flush, total, blocks]; read back with
// omr_debug_runs_stamps().  No stamp exists in the release library.
#ifdef OMR_RUNS_DEBUG
__device__ unsigned long long g_run_stamps[8];
__device__ __forceinline__ unsigned long long run_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define RUN_STAMP_DECL unsigned long long st_t0 = run_clock(), st_prev = st_t0, st_acc[5] = {0, 0, 0, 0, 0};
#define RUN_STAMP(I)                                 \
    {                                                \
        const unsigned long long now_ = run_clock(); \
        st_acc[I] += now_ - st_prev;                 \
        st_prev = now_;                              \
    }
#define RUN_STAMP_END                                                          \
    if (tid == 0) {                                                            \
        for (int i = 0; i < 5; i++) atomicAdd(&g_run_stamps[i], st_acc[i]);    \
        atomicAdd(&g_run_stamps[5], run_clock() - st_t0);                      \
        atomicAdd(&g_run_stamps[6], 1ull);                                     \
    }
#else
#define RUN_STAMP_DECL
#define RUN_STAMP(I)
#define RUN_STAMP_END
#endif

__global__ __launch_bounds__(RUN_BAND, 4) void runs_kernel(const RunPass p, const int32_t *__restrict__ list,
                                                           int32_t *__restrict__ guard, uint32_t *__restrict__ vproj)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    RUN_STAMP_DECL
    // grid = (scans, chunks, candidates), scans fastest.  Workgroups go round-robin to the 8 XCDs, so with 8
    // (4, 2) scans per launch XCD x only ever sweeps scan x (x mod 4, x mod 2): that scan's 1.1 MB bit
    // image stays in the XCD's 4 MB L2 and every window fetch hits it.
    const int a = __builtin_amdgcn_readfirstlane(list[blockIdx.z]);
    const int pc = blockIdx.y / p.RCH, rc = blockIdx.y - pc * p.RCH;  // chunk of word groups, chunk of rows
    const int zscan = blockIdx.x;  // scan of the launch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the inner loop addresses LDS by integer: the dynamic segment must start at LDS address 0 (no
    // static LDS in this kernel).  If a toolchain ever places it elsewhere the candidate is handed to
    // the gather kernel instead of computing from wrong addresses.
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds != 0u) {
        if (tid == 0) guard[a] = 1;
        return;
    }
    const int g_begin = pc * p.GC, g_end = min(p.G, g_begin + p.GC);
    const int ngroups = g_end - g_begin;
    // images taller than the LDS row counters are cut into row chunks of p.RB bands; the column counts of
    // the chunks then meet in vproj by atomics
    const int band0 = rc * p.RB, row0 = band0 * RUN_BAND;
    const int NB = min(p.RB, (p.NR + RUN_BAND - 1) / RUN_BAND - band0);  // bands per word group in this chunk
    const int NRh = (min(p.NRp - row0, p.RB * RUN_BAND) + 1) >> 1;      // dwords of the row-count array
    const int2_t *__restrict__ RT = p.RT + (int64_t)a * p.NR;

    // ---- descriptors (wave-uniform): the scan's transposed bit image, the candidate's run tables
    const v4u32 rs_img = make_rsrc(p.srcT + (int64_t)zscan * p.NWt * p.rowsT, (uint32_t)p.NWt * (uint32_t)p.rowsT * 4u);
    const char *tab_base = (const char *)(p.tabs + ((int64_t)a * p.NWp + (int64_t)g_begin * RUN_K));
    const char *met_base = (const char *)(p.metac + ((int64_t)a * p.NWp + (int64_t)g_begin * RUN_K));
    // per-row (X0, Y0): one 8-byte buffer load per lane and band; the last row stands in for rows past the end
    const __amdgpu_buffer_rsrc_t rs_rt = __builtin_amdgcn_make_buffer_rsrc((void *)RT, 0, p.NR * 8, 0x00020000);
    auto load_rt = [&](const int band) -> int2_t {
        const int r = min(row0 + band * RUN_BAND + tid, p.NR - 1);
        const v2u32 v = __builtin_bit_cast(v2u32, __builtin_amdgcn_raw_buffer_load_b64(rs_rt, r * 8, 0, 0));
        return int2_t{(int32_t)v.x, (int32_t)v.y};
    };

    // ---- prologue: the chunk's word-group constants go to LDS, the row counts start at 0
    if (tid < ngroups * 8) {
        const int32_t *src = (const int32_t *)(p.blk + ((int64_t)a * p.G + g_begin));
        *(int32_t *)(lds + RUN_GEO_OFS + tid * 4) = src[tid];
    }
    for (int i = tid; i < NRh; i += RUN_BAND) *(uint32_t *)(lds + RUN_HROW_OFS + i * 4) = 0u;

    // This wave's window origin of a (group, band): one scalar load from the plan's table (rungeo_kernel)
    const int NBt = (p.NR + RUN_BAND - 1) / RUN_BAND;
    const int2_t *__restrict__ wgeo_a = p.wgeo + ((int64_t)a * p.G + g_begin) * NBt * RUN_WAVES;
    // (an s_load spelled out: the compiler will not use the scalar unit for a pointer out of the argument struct,
    // and a vector load costs a dozen VALU operations of address arithmetic and a vmcnt wait.  The result is valid
    // after the caller's next `s_waitcnt lgkmcnt(0)`.)
    // The table is walked in order -- (group, band) -> the next band, or band 0 of the next group -- so its byte offset
    // is a running SCALAR: computed from the loop counters it was a 32-bit vector multiply, a select and a
    // v_readfirstlane per band (the compiler keeps the group counter in a VGPR).
    uint32_t goff = (uint32_t)__builtin_amdgcn_readfirstlane((band0 * RUN_WAVES + wave) * 8);
    const uint32_t goff_band = RUN_WAVES * 8, goff_group = (uint32_t)__builtin_amdgcn_readfirstlane((NBt - NB + 1) * RUN_WAVES * 8);
    auto geometry_issue = [&](const uint32_t off) -> unsigned long long {
        unsigned long long q;
        asm volatile("s_load_dwordx2 %0, %1, %2" : "=&s"(q) : "s"(wgeo_a), "s"(off) : "memory");
        return q;
    };
    auto geometry_take = [&](const unsigned long long q) -> RunGeom {
        RunGeom g;
        g.wxw = (int)(uint32_t)q;
        g.wy0 = (int)(uint32_t)(q >> 32);
        g.fits = g.wxw != 0x7fffffff;
        return g;
    };
    // This wave's window -> LDS, column-major, 16-byte pieces of 4 rows, two word columns per
    // wave-instruction (lanes 0-27 and 28-55).  The transposed bit image carries a zero guard that holds every
    // window of the plan: no range test, one VALU add per wave-instruction.
    const int lane_hi = lane >= RUN_WPIECES ? 1 : 0;
    const uint32_t lane_off = (uint32_t)((lane_hi * p.rowsT + (lane - lane_hi * RUN_WPIECES) * 4) * 4);
    const uint32_t winbase = RUN_WIN_OFS + (uint32_t)wave * RUN_WIN_BYTES;
    auto fetch_window = [&](const RunGeom &q) {
        const uint32_t sbase = (uint32_t)(((q.wxw + p.GX) * p.rowsT + q.wy0 + p.GY) * 4);
#pragma unroll
        for (int n = 0; n < (RUN_WCOLS + 1) / 2; n++) {
            const uint32_t voff = lane_off + (sbase + (uint32_t)(2 * n) * (uint32_t)p.rowsT * 4u);
            const int nl = 2 * n + 1 < RUN_WCOLS ? 2 * RUN_WPIECES : RUN_WPIECES;
            if (lane < nl) dma_b128(rs_img, voff, winbase + (uint32_t)(2 * n * RUN_WROWS * 4));
        }
    };
    // run tables + (ca0, cb0) pairs of one word group -> LDS (all waves share the work)
    auto fetch_tables = [&](const int gl, const int set) {
        const v4u32 rs_tab = make_rsrc(tab_base + (int64_t)gl * RUN_TABSET_BYTES, RUN_TABSET_BYTES);
        const v4u32 rs_met = make_rsrc(met_base + (int64_t)gl * (RUN_K * 8), RUN_K * 8);
        const uint32_t tabbase = set ? RUN_TAB1_OFS : RUN_TAB0_OFS;
#pragma unroll
        for (int n = 0; n < (RUN_TABSET_BYTES / 16 + RUN_BAND - 1) / RUN_BAND; n++) {
            const int i = n * RUN_BAND + tid;
            if (i < RUN_TABSET_BYTES / 16) dma_b128(rs_tab, (uint32_t)(i * 16), tabbase + (uint32_t)(n * RUN_BAND * 16 + wave * 1024));
        }
        if (tid < RUN_K * 2) dma_b32(rs_met, (uint32_t)(tid * 4), (uint32_t)(RUN_META_OFS + set * (RUN_K * 8)));
    };

    __syncthreads();  // word-group constants are in LDS
    fetch_tables(0, 0);
    int2_t rt = load_rt(0);  // this band's rows (the next band's are loaded while this one is swept)
    unsigned long long gq = geometry_issue(goff);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RunGeom cur = geometry_take(gq);
    if (cur.fits) fetch_window(cur);
    const uint32_t hrow_lane = RUN_HROW_OFS + (uint32_t)(tid >> 1) * 4u, hrow_shift = (uint32_t)(tid & 1) * 16u;

    uint32_t c0[RUN_K], c1[RUN_K], c2[RUN_K];
#pragma unroll
    for (int k = 0; k < RUN_K; k++) c0[k] = c1[k] = c2[k] = 0;
    uint32_t acc = 0;  // lanes 0-15: column totals of this wave's (word, half) of the current group
    bool bad = false;
    RUN_STAMP(0)

    for (int gl = 0; gl < ngroups; gl++) {
        const int tset = gl & 1;
        // the group's tables: this wave's share has landed; after the barrier every wave's has (the first
        // group's barrier is this one, later groups met at the end of the previous group's flush)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (gl == 0) __syncthreads();
        if (gl + 1 < ngroups) fetch_tables(gl + 1, tset ^ 1);  // every wave has left the other set behind
        const uint32_t tabv = tset ? RUN_TAB1_OFS : RUN_TAB0_OFS;
        RunWordK wk[RUN_K];
#pragma unroll
        for (int k = 0; k < RUN_K; k++) {
            const int2 m = *(const int2 *)(lds + RUN_META_OFS + tset * (RUN_K * 8) + k * 8);
            wk[k].ca0 = m.x, wk[k].cb0 = m.y;
        }
        const int32_t *bk = (const int32_t *)(lds + RUN_GEO_OFS + gl * 32);
        const int nlev = __builtin_amdgcn_readfirstlane(bk[0]), lagbits = __builtin_amdgcn_readfirstlane(bk[7]);
        int bands_pending = 0;
        for (int band = 0; band < NB; band++) {
            // ---- this wave's window of (group, band) has landed (its own DMA: no barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RUN_STAMP(1)
            const RunGeom now = cur;
            bad |= !now.fits;
            // the next step's window origin (a scalar load) and rows (X0, Y0) travel while this band is swept
            const bool last_band = band == NB - 1;
            const bool more = !last_band || gl + 1 < ngroups;
            int2_t rt_n = rt;
            if (more) {
                const int band_n = last_band ? 0 : band + 1;
                goff = (uint32_t)__builtin_amdgcn_readfirstlane(goff + (last_band ? goff_group : goff_band));
                gq = geometry_issue(goff);
                rt_n = load_rt(band_n);
            }
            const int r = band * RUN_BAND + tid;  // row within the chunk
            // rows past the end of the image are masked off by EXEC for the whole compute phase (their
            // counters must not move)
            if (now.fits && row0 + r < p.NR) {
                const int rx = rt.x - (now.wxw << 15);  // window-local fixed point
                const int ry = rt.y - (now.wy0 << 10) + (int)((winbase / 4) << 10);
                // (the instruction arbiter serves a wave that is merging before one that is issuing DMAs or reducing
                // counters: -0.9 %, measured side by side)
                __builtin_amdgcn_s_setprio(1);
                const uint32_t cnt = band_words_s(rx, ry, tabv, wk, c0, c1, c2, nlev, lagbits);
                __builtin_amdgcn_s_setprio(RUN_PRIO_REST);
                // two u16 row counts per dword: the lanes of a pair add into the same dword
                __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)(hrow_lane + (uint32_t)(band * (RUN_BAND * 2))), cnt << hrow_shift,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            RUN_STAMP(2)
            // ---- the next window of this wave goes out as soon as its own reads of this one are done
            if (more) {
                rt = rt_n;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                cur = geometry_take(gq);
                // (... and one that is about to send its next window's DMAs before either: another -0.8 %)
                __builtin_amdgcn_s_setprio(2);
                if (cur.fits) fetch_window(cur);
                __builtin_amdgcn_s_setprio(0);
            }
            RUN_STAMP(3)
            // the 3-plane counters hold at most 7 rows per lane: reduce them across lanes in time
            if (++bands_pending == RUN_FLUSH_BANDS || last_band) {
                bands_pending = 0;
                acc = flush_columns(c0, c1, c2, tid, acc);
                if (last_band) {
                    // column counts of this wave's 16 columns over ALL rows: one plain store each
                    const int col = ((g_begin + gl) * RUN_K + (wave >> 1)) * 32 + ((wave & 1) ? 0 : 16) + lane;
                    if (lane < 16 && col < p.NC) {
                        uint32_t *dst = vproj + ((int64_t)zscan * p.A + a) * p.NC + col;
                        if (p.RCH == 1) *dst = acc;
                        else atomicAdd(dst, acc);  // vproj was zeroed before the launch
                    }
                    acc = 0;
                }
            }
            RUN_STAMP(4)
        }
    }
    if (__ballot(bad) != 0ull && lane == 0) guard[a] = 1;
    __syncthreads();
    // row counts of this chunk of word groups: u16 partials, two per dword
    uint32_t *__restrict__ out = (uint32_t *)(p.part + (((int64_t)zscan * p.A + a) * p.P + pc) * p.NRp + row0);
    for (int i = tid; i < NRh; i += RUN_BAND) out[i] = *(const uint32_t *)(lds + RUN_HROW_OFS + i * 4);
    RUN_STAMP_END
}

#ifdef OMR_RUNS_DEBUG
hipError_t debug_runs_stamps(unsigned long long out[8], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_run_stamps), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_run_stamps), z, sizeof z);
    }
    return e;
}
#endif

hipError_t launch_runs(const RunPass &p0, const int32_t *d_list, int n_list, int32_t *d_guard, uint32_t *d_vproj,
                       hipStream_t s)
{
    if (n_list <= 0) return hipSuccess;
    RunPass p = p0;
    if (p.RB < 1 || p.RB * RUN_BAND > OMR_RUN_MAX_ROWS || p.RCH < 1 || p.RCH * p.RB * RUN_BAND < p.NR || p.GC < 1 ||
        p.GC > OMR_RUN_GC || (p.NRp & 1) || p.NRp < p.NR)
        return hipErrorInvalidValue;
    // per device and idempotent; cheap enough to repeat (the batch entry points use every device)
    hipError_t e = hipFuncSetAttribute((const void *)runs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RUN_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (p.scans < 1) p.scans = 1;
    hipLaunchKernelGGL(runs_kernel, dim3(p.scans, p.P * p.RCH, n_list), dim3(RUN_BAND), RUN_LDS_BYTES, s, p, d_list, d_guard,
                       d_vproj);
    return hipGetLastError();
}

// proj[a][r] = sum_p part[a][p][r] for the listed candidates
__global__ __launch_bounds__(256) void fold_parts_kernel(const uint16_t *__restrict__ part, int P, int NR, int NRp,
                                                         const int32_t *__restrict__ list,
                                                         uint32_t *__restrict__ proj, int A)
{
    part += (int64_t)blockIdx.z * A * P * NRp;  // blockIdx.z = scan of the launch
    proj += (int64_t)blockIdx.z * A * NR;
    const int a = list[blockIdx.y];
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= NR) return;
    uint32_t s = 0;
    for (int q = 0; q < P; q++) s += part[((int64_t)a * P + q) * NRp + r];
    proj[(int64_t)a * NR + r] = s;
}

hipError_t launch_fold_parts(const uint16_t *d_part, int P, int NR, int NRp, const int32_t *d_list, int n_list,
                             uint32_t *d_proj, hipStream_t s, int scans, int A)
{
    if (n_list <= 0 || scans <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_parts_kernel, dim3((NR + 255) / 256, n_list, scans), dim3(256), 0, s, d_part, P, NR, NRp,
                       d_list, d_proj, A);
    return hipGetLastError();
}

}  // namespace omr
