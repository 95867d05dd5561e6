// runs.hip -- the run-merging ("S") sweep kernels for gfx950.
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
// one 32-bit unaligned window per source row (LDS, ds_read2 + v_alignbit), and one and/or per row.
// Row counts come from this pass on the bit image, column counts from the SAME pass on the
// transposed bit image with the tables' roles swapped -- no cross-lane reduction anywhere.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace omr {

// Development switches exist only in the debug build (`make debug`, -DOMR_RUNS_DEBUG, loaded by
// tools/kstamps.py): a release library reads no environment variable and has no stage-skipping path.
#ifdef OMR_RUNS_DEBUG
#define RUN_DBG(p) ((p).dbg)
#else
#define RUN_DBG(p) 0
#endif

#define RUN_K OMR_RUN_K    // destination words per block column group (kernels.hpp)
#define RUN_BAND 512       // destination rows per band (8 waves x 64 lanes)
#define RUN_QUADS ((RUN_K * 32 + 127 + 96) / 128 + 2)  // aligned 4-word pieces per window row
#define RUN_PITCH (RUN_QUADS * 4 + 1)  // window row pitch in words (data words + 1 spill word), odd
#define RUN_PITCHB (RUN_PITCH * 4)
#define RUN_WIN_ROWS 588   // window rows; 588 x 84 B also holds the 8 x 3 x 512 counter words of a flush
#define RUN_TAB_BYTES 3648
#define RUN_TUPHI_OFS 640
#define RUN_TUPX_OFS 1280
#define RUN_IDXY_OFS 1600
#define RUN_IDXX_OFS 2624
// LDS layout: window first (so a level's row offset fits ds_read2's 8-bit offset fields), then
// the RUN_K (ca0, cb0) pairs, then the RUN_K run tables
#define RUN_WIN_OFS 0
#define RUN_META_OFS (RUN_WIN_ROWS * RUN_PITCHB)
#define RUN_TABS_OFS (RUN_META_OFS + RUN_K * 8)
#define RUN_COL_OFS (RUN_TABS_OFS + RUN_K * RUN_TAB_BYTES)  // column totals of the block, u32[RUN_K * 32]
#define RUN_LDS_BYTES (RUN_COL_OFS + RUN_K * 32 * 4)
// no static LDS in runs_kernel: the dynamic segment then starts at LDS address 0 and every table
// offset folds into the ds_read immediate
static_assert(2 * RUN_LDS_BYTES <= 160 * 1024, "two blocks per CU");

static_assert(sizeof(RunTab) == RUN_TAB_BYTES, "RunTab layout");
static_assert(sizeof(RunMeta) == 32, "RunMeta layout");

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

__global__ __launch_bounds__(1024) void runtab_kernel(const int32_t *__restrict__ CA, const int32_t *__restrict__ CB,
                                                      int NC, int NW, RunTab *__restrict__ tabs,
                                                      RunMeta *__restrict__ meta)
{
    __shared__ int s_ca[32], s_cb[32];
    __shared__ uint32_t s_tup[1024][9];  // padded: conflict-free row compare
    __shared__ int s_wave[16];
    __shared__ int s_bad, s_smax;
    const int w = blockIdx.x, a = blockIdx.y, c0 = w * 32, f = threadIdx.x;
    const int32_t *ca_row = CA + (int64_t)a * NC, *cb_row = CB + (int64_t)a * NC;
    if (f < 32) {
        // columns past the end of the row (last word only) are don't-cares: the pass kernel masks
        // them with RunMeta::valid.  Give them lag 0 and the row of the last real column so they add
        // neither a level nor a validity failure.
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
    RunTab *T = tabs + ((int64_t)a * NW + w);
    if (idy >= OMR_RUN_TUPLES) {
        bad = true;
        idy = 0;
    } else if (change) {
        // destination bits that exist (last word of a row): folded into the level masks, so the sweep
        // kernel needs no per-word validity mask (the lag merge happens before the level select)
        const uint32_t valid = (c0 + 32 <= NC) ? 0xffffffffu : ((1u << (NC - c0)) - 1u);
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
        m.valid = (c0 + 32 <= NC) ? 0xffffffffu : ((1u << (NC - c0)) - 1u);
        m.ok = s_bad ? 0 : 1;
        m.pad0 = m.pad1 = 0;
        meta[(int64_t)a * NW + w] = m;
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
                         RunMeta *d_meta, RunBlk *d_blk, hipStream_t s)
{
    hipLaunchKernelGGL(runtab_kernel, dim3(NW, A), dim3(1024), 0, s, d_CA, d_CB, NC, NW, d_tabs, d_meta);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int G = (NW + RUN_K - 1) / RUN_K;
    hipLaunchKernelGGL(runblk_kernel, dim3((A * G + 255) / 256), dim3(256), 0, s, d_CA, d_CB, A, NC, NW, G, d_meta,
                       d_blk);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The sweep kernel.  Block = (candidate, group of RUN_K words = 256 destination columns); it walks
// ALL destination rows in bands of 512 (8 waves x 64 lanes, lane = destination row).
//   row counts   : popcount of the lane's words, one u16 partial per (row, word group)
//   column counts: every lane keeps a 3-plane bit-sliced counter per word (its rows of up to 7
//                  bands); the counters are reduced across the 64 lanes with one
//                  v_add_co_u32 (shift + carry-out = ballot of the top bit) and one s_bcnt1 per bit,
//                  added up in LDS over the 8 waves and stored once per block (no atomics).
// LDS accesses of the inner loop take INTEGER byte addresses (the dynamic segment starts at LDS address
// 0, checked once per block): table offsets then fold into the ds_read immediates and no per-access
// "base + offset" VALU add is left.
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
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
// One word = 22 cycles of addressing + (8 NLEV - 2) of merge + 10 of column counters + 4 of row count.

// one word: NLEV unaligned 32-bit windows -> destination word
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
        if (SMAX >= 1) U = (sx.x & (W << 1)) | (~sx.x & U);
        if (SMAX >= 2) U = (sx.y & (W << 2)) | (~sx.y & U);
        D = lv == 0 ? (U & sel[0]) : (D | (U & sel[lv]));
    }
    return D;
}

// bit-sliced add of a 1-bit value per column into a 3-plane counter: five 2-cycle ops
__device__ __forceinline__ void count_columns(uint32_t &c0, uint32_t &c1, uint32_t &c2, const uint32_t D)
{
    const uint32_t t = c0 & D;
    c0 ^= D;
    const uint32_t u = c1 & t;
    c1 ^= t;
    c2 |= u;
}

// One pair of words of one band for one wave.  Both words are in flight together: both fraction
// look-ups and all 2 x NLEV window reads are issued before anything is consumed (LDS latency hiding at
// 4 waves per SIMD).  Words past the end of the row were staged with
// empty level masks, the last word's masks carry the row-end mask (runtab_kernel), rows past the end
// of the image are EXEC-masked by the caller: nothing to mask here.
template <int NLEV, int SMAX, int K>
__device__ __forceinline__ uint32_t pair_words(const int rx, const int ry, uint32_t (&c0)[RUN_K], uint32_t (&c1)[RUN_K],
                                               uint32_t (&c2)[RUN_K])
{
    constexpr int k = K;
    constexpr uint32_t ta = RUN_TABS_OFS + k * RUN_TAB_BYTES, tb = ta + RUN_TAB_BYTES;
    const int4 m = ldsr_i4(RUN_META_OFS + k * 8);  // (ca0, cb0) of both words: same address in every lane
    const int A0a = rx + m.x, B0a = ry + m.y, A0b = rx + m.z, B0b = ry + m.w;
    uint32_t Da, Db;
    const uint32_t idya = ldsr_u8(ta + RUN_IDXY_OFS + (uint32_t)(B0a & 1023));
    const uint32_t idyb = ldsr_u8(tb + RUN_IDXY_OFS + (uint32_t)(B0b & 1023));
    uint32_t idxa = 0, idxb = 0;
    if (SMAX > 0) {
        idxa = ldsr_u8(ta + RUN_IDXX_OFS + (uint32_t)(A0a & 1023));
        idxb = ldsr_u8(tb + RUN_IDXX_OFS + (uint32_t)(A0b & 1023));
    }
    // window byte address: row * pitch + first word * 4 (v_mad_i32_i24; window-local coordinates are >= 0)
    const uint32_t addra = RUN_WIN_OFS + (uint32_t)(__mul24(B0a >> 10, RUN_PITCHB) + ((A0a >> 13) & ~3));
    const uint32_t addrb = RUN_WIN_OFS + (uint32_t)(__mul24(B0b >> 10, RUN_PITCHB) + ((A0b >> 13) & ~3));
    uint2 wa[NLEV], wb[NLEV];
#pragma unroll
    for (int lv = 0; lv < NLEV; lv++) {  // two adjacent dwords, 4-byte aligned: ds_read2_b32
        wa[lv].x = ldsr_u32(addra + lv * RUN_PITCHB);
        wa[lv].y = ldsr_u32(addra + lv * RUN_PITCHB + 4);
    }
#pragma unroll
    for (int lv = 0; lv < NLEV; lv++) {
        wb[lv].x = ldsr_u32(addrb + lv * RUN_PITCHB);
        wb[lv].y = ldsr_u32(addrb + lv * RUN_PITCHB + 4);
    }
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const uint32_t tya = ta + (idya << 4), tyb = tb + (idyb << 4);
    const uint4 sa03 = ldsr_u4(tya), sa47 = NLEV > 4 ? ldsr_u4(tya + RUN_TUPHI_OFS) : z4;
    const uint4 sb03 = ldsr_u4(tyb), sb47 = NLEV > 4 ? ldsr_u4(tyb + RUN_TUPHI_OFS) : z4;
    uint2 sxa = make_uint2(0, 0), sxb = make_uint2(0, 0);
    if (SMAX > 0) {
        sxa = ldsr_u2(ta + RUN_TUPX_OFS + (idxa << 3));
        sxb = ldsr_u2(tb + RUN_TUPX_OFS + (idxb << 3));
    }
#ifdef RUN_FENCE_MERGE
    __builtin_amdgcn_sched_barrier(0);
#endif
    Da = merge_word<NLEV, SMAX>(wa, (uint32_t)A0a >> 10, sa03, sa47, sxa);
#ifdef RUN_FENCE_MERGE
    count_columns(c0[k], c1[k], c2[k], Da);
    __builtin_amdgcn_sched_barrier(0);
    Db = merge_word<NLEV, SMAX>(wb, (uint32_t)A0b >> 10, sb03, sb47, sxb);
#else
    Db = merge_word<NLEV, SMAX>(wb, (uint32_t)A0b >> 10, sb03, sb47, sxb);
    count_columns(c0[k], c1[k], c2[k], Da);
#endif
    count_columns(c0[k + 1], c1[k + 1], c2[k + 1], Db);
    return __popc(Da) + __popc(Db);
}

// A pair's column lag bound (0, 1 or 2; wave-uniform) picks its specialisation: most words of a
// candidate need no lag handling even when some word of the block does.
template <int NLEV, int K>
__device__ __forceinline__ uint32_t pair_dispatch(const int rx, const int ry, uint32_t (&c0)[RUN_K],
                                                  uint32_t (&c1)[RUN_K], uint32_t (&c2)[RUN_K], const int lagbits)
{
    // keep the pairs apart: without this fence the scheduler hoists all four pairs' loads and the
    // kernel needs 180 VGPRs (2 waves per SIMD instead of 4)
    __builtin_amdgcn_sched_barrier(0);
    const int lag = (lagbits >> K) & 3;
    if (lag == 0) return pair_words<NLEV, 0, K>(rx, ry, c0, c1, c2);
    if (lag == 1) return pair_words<NLEV, 1, K>(rx, ry, c0, c1, c2);
    return pair_words<NLEV, 2, K>(rx, ry, c0, c1, c2);
}

// The RUN_K words of one band for one wave.  NLEV is uniform for the whole block (maximum over its
// words; unused levels have empty masks), so the block dispatches once per band to a straight-line
// specialisation.
template <int NLEV>
__device__ __forceinline__ uint32_t band_words(const int rx, const int ry, uint32_t (&c0)[RUN_K], uint32_t (&c1)[RUN_K],
                                               uint32_t (&c2)[RUN_K], const int lagbits)
{
    static_assert(RUN_K == 8, "four pairs");
    uint32_t cnt = pair_dispatch<NLEV, 0>(rx, ry, c0, c1, c2, lagbits);
    cnt += pair_dispatch<NLEV, 2>(rx, ry, c0, c1, c2, lagbits);
    cnt += pair_dispatch<NLEV, 4>(rx, ry, c0, c1, c2, lagbits);
    cnt += pair_dispatch<NLEV, 6>(rx, ry, c0, c1, c2, lagbits);
    return cnt;
}

__device__ __forceinline__ uint32_t band_words_s(const int rx, const int ry, uint32_t (&c0)[RUN_K], uint32_t (&c1)[RUN_K],
                                                 uint32_t (&c2)[RUN_K], const int nlev, const int lagbits)
{
    switch (nlev) {
    case 1: return band_words<1>(rx, ry, c0, c1, c2, lagbits);
    case 2: return band_words<2>(rx, ry, c0, c1, c2, lagbits);
    case 3: return band_words<3>(rx, ry, c0, c1, c2, lagbits);
    case 4: return band_words<4>(rx, ry, c0, c1, c2, lagbits);
    case 5: return band_words<5>(rx, ry, c0, c1, c2, lagbits);
    case 6: return band_words<6>(rx, ry, c0, c1, c2, lagbits);
    case 7: return band_words<7>(rx, ry, c0, c1, c2, lagbits);
    default: return band_words<8>(rx, ry, c0, c1, c2, lagbits);
    }
}

// Column counts of one word group: every lane holds a 3-plane bit-sliced counter per word (its rows
// of the last <= 7 bands).  All 512 lanes park their counters in LDS (the window region is free
// between bands); wave w then owns word w: it adds the 8 waves' counters lane-wise into a 6-plane
// number (bit-sliced ripple adds), sums that over its 64 lanes with shl1_sum6 -- one column per
// step, most significant bit first -- and adds the 32 column totals to colacc.
// (reduce_columns is deliberately NOT inlined: its 32 scalar totals and asm temporaries would
// otherwise raise the register pressure of the band loop, which runs at exactly 128 VGPRs.)
__device__ __noinline__ void reduce_columns(const char *lds, uint32_t *colacc, const int tid)
{
    const uint32_t *park = (const uint32_t *)(lds + RUN_WIN_OFS);  // [word][plane][512 lanes]
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t *mine = park + (wave * 3) * RUN_BAND + lane;  // word `wave`
    uint32_t s0 = mine[0], s1 = mine[RUN_BAND], s2 = mine[2 * RUN_BAND], s3 = 0, s4 = 0, s5 = 0;
#pragma unroll
    for (int wv = 1; wv < 8; wv++) {
        const uint32_t x0 = mine[wv * 64], x1 = mine[RUN_BAND + wv * 64], x2 = mine[2 * RUN_BAND + wv * 64];
        uint32_t c = s0 & x0;  // bit-sliced s += x
        s0 ^= x0;
        uint32_t t = s1 ^ x1 ^ c;
        c = (s1 & x1) | (c & (s1 ^ x1));
        s1 = t;
        t = s2 ^ x2 ^ c;
        c = (s2 & x2) | (c & (s2 ^ x2));
        s2 = t;
        t = s3 ^ c;
        c &= s3;
        s3 = t;
        t = s4 ^ c;
        c &= s4;
        s4 = t;
        s5 ^= c;
    }
    uint32_t tot[32];
#pragma unroll
    for (int b = 31; b >= 0; b--) tot[b] = shl1_sum6(s0, s1, s2, s3, s4, s5);
    uint32_t v = 0;
#define WL(B) v = write_lane_imm<B>(v, tot[B]);
    WL(0) WL(1) WL(2) WL(3) WL(4) WL(5) WL(6) WL(7) WL(8) WL(9) WL(10) WL(11) WL(12) WL(13) WL(14) WL(15)
    WL(16) WL(17) WL(18) WL(19) WL(20) WL(21) WL(22) WL(23) WL(24) WL(25) WL(26) WL(27) WL(28) WL(29) WL(30) WL(31)
#undef WL
    if (lane < 32) colacc[wave * 32 + lane] += v;  // only this wave touches word `wave`
}

__device__ __forceinline__ void flush_columns(char *lds, uint32_t (&c0)[RUN_K], uint32_t (&c1)[RUN_K],
                                              uint32_t (&c2)[RUN_K], uint32_t *colacc, const int tid)
{
    static_assert(RUN_K == 8 && RUN_BAND == 512, "one wave per word");
    static_assert(RUN_K * 3 * RUN_BAND * 4 <= RUN_WIN_ROWS * RUN_PITCHB, "counter scratch must fit the window");
    uint32_t *park = (uint32_t *)(lds + RUN_WIN_OFS);  // [word][plane][512 lanes]
    __syncthreads();  // every wave is done with the window of the last band
#pragma unroll
    for (int k = 0; k < RUN_K; k++) {
        park[(k * 3 + 0) * RUN_BAND + tid] = c0[k];
        park[(k * 3 + 1) * RUN_BAND + tid] = c1[k];
        park[(k * 3 + 2) * RUN_BAND + tid] = c2[k];
        c0[k] = c1[k] = c2[k] = 0;
    }
    __syncthreads();
    reduce_columns(lds, colacc, tid);
}

// Diagnostic build only (OMR_RUNS_DBG=8): per-block phase clocks of wave 0, summed into a global
// array [tables, commit+barriers, prefetch-issue, compute, flush, total]; read back with
// omr_debug_runs_stamps().  No stamp executes in a production run (dbg == 0).
__device__ unsigned long long g_run_stamps[8];
__device__ __forceinline__ unsigned long long run_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

struct RunGeom {  // source window of one band (wave-uniform)
    int wxw, wy0, nrows;
    int nq;  // aligned 4-word pieces per window row that hold samples (3..RUN_QUADS)
    bool fits;
};

#define RUN_FLUSH_BANDS 7  // 3-plane counters hold up to 7 rows per lane

template <bool STAMP>
__global__ __launch_bounds__(RUN_BAND, 4) void runs_kernel(const RunPass p, const int32_t *__restrict__ list,
                                                        int32_t *__restrict__ guard, uint32_t *__restrict__ vproj)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t *colacc = (uint32_t *)(lds + RUN_COL_OFS);
    constexpr bool stamp = STAMP;
    unsigned long long st_t0 = 0, st_prev = 0, st_acc[5] = {0, 0, 0, 0, 0};
    if (stamp) st_t0 = st_prev = run_clock();
#define RUN_STAMP(I)                                   \
    if (stamp) {                                       \
        const unsigned long long now_ = run_clock();   \
        st_acc[I] += now_ - st_prev;                   \
        st_prev = now_;                                \
    }
    // grid = (scans, word groups, candidates), scans fastest.  Workgroups go round-robin to the 8 XCDs, so
    // with 8 (4, 2) scans per launch XCD x only ever sweeps scan x (x mod 4, x mod 2): that scan's 1.09 MB
    // bit image stays in the XCD's 4 MB L2 and every window fetch hits it.  (Tried: giving an XCD all
    // scans of a (candidate, word group) pair so that the pair's run table is fetched once -- 3.3 k
    // instead of 3.7 k images/s, the eight bit images then evict each other.)
    const int a = __builtin_amdgcn_readfirstlane(list[blockIdx.z]);
    const int g = blockIdx.y;
    const int zscan = blockIdx.x;  // scan of the launch
    const int w0 = g * RUN_K;
    const int kw = min(RUN_K, p.NW - w0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the inner loop addresses LDS by integer: the dynamic segment must start at LDS address 0 (no
    // static LDS in this kernel).  If a toolchain ever places it elsewhere the candidate is handed to
    // the gather kernel instead of computing from wrong addresses.
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds != 0u) {
        if (tid == 0) guard[a] = 1;
        return;
    }

#ifdef OMR_RUNS_DEBUG
    {   // experiment: de-phase the two workgroups of a CU (they otherwise run fetch / compute / flush in lockstep)
        const int st = p.dbg >> 8;  // units of 1024 cycles
        const uint32_t tg = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (16 << 6) | 4);  // HW_ID.TG_ID
        if (st > 0 && (tg & 1)) for (int i = 0; i < st; i++) __builtin_amdgcn_s_sleep(16);
    }
#endif
    // block constants: one scalar load (the address is uniform)
    const RunBlk bk = p.blk[(int64_t)a * p.G + g];
    const int nlev_blk = bk.nlev, lagbits = bk.lagbits;
    const int ca_min = bk.ca_min, ca_max = bk.ca_max, cb_min = bk.cb_min, cb_max = bk.cb_max;
    const int2_t *__restrict__ RT = p.RT + (int64_t)a * p.NR;
    const int c_first = w0 * 32;
    uint16_t *__restrict__ out = p.part + (((int64_t)zscan * p.A + a) * p.G + g) * p.NR;

    // source bounding box of a band x word group: the map is monotone in r and in c, so the four
    // corner samples bound it.  The corner rows' (X0, Y0) are fetched one band ahead of their use.
    auto corners = [&](int yb, int2_t &t0, int2_t &t1) {
        const int yy = min(yb, p.NR - 1);
        t0 = RT[yy];
        t1 = RT[min(p.NR, yy + RUN_BAND) - 1];
    };
    auto geometry = [&](const int2_t t0, const int2_t t1) -> RunGeom {
        RunGeom q;
        const int t0x = __builtin_amdgcn_readfirstlane(t0.x), t0y = __builtin_amdgcn_readfirstlane(t0.y);
        const int t1x = __builtin_amdgcn_readfirstlane(t1.x), t1y = __builtin_amdgcn_readfirstlane(t1.y);
        const int minbit = (min(t0x, t1x) + ca_min) >> 10;  // min over the four corners
        const int maxbit = (max(t0x, t1x) + ca_max) >> 10;
        const int minrow = (min(t0y, t1y) + cb_min) >> 10;
        const int maxrow = (max(t0y, t1y) + cb_max) >> 10;
        q.wxw = (minbit >> 5) & ~3;  // first window word (multiple of 4: 16-byte loads)
        q.wy0 = minrow - 7;          // a word reads up to 7 rows beside its true samples
        q.nrows = maxrow - minrow + 15;
        // every selected sample lies in words wxw .. maxbit >> 5; the word after a sample's word is
        // read too but none of its bits is ever selected, so it may hold anything
        q.nq = max(3, (((maxbit >> 5) - q.wxw) >> 2) + 1);
        q.fits = q.nq <= RUN_QUADS && q.nrows <= RUN_WIN_ROWS;
        return q;
    };
    // The window (nrows x nq aligned 16-byte pieces, zero outside the image) is fetched into
    // registers one band ahead, so the fetch of band b+1 overlaps the compute of band b and every
    // wave carries the same share.  A round of the block covers rpr = 512 / nq whole rows: thread t
    // owns piece t % nq of row t / nq + n * rpr in round n, so both its global and its LDS address
    // advance by a wave-uniform stride.  The loads go through a buffer descriptor of the bit image:
    // rows above / below the image give byte offsets outside [0, num_records) (as unsigned), which
    // the hardware range check turns into zeros, and a thread whose words lie left / right of the
    // image (the same in every round) uses an offset that is out of range in every round -- no
    // compare, no branch and no zero fill per piece.
    constexpr int PIECES = (RUN_WIN_ROWS * RUN_QUADS + RUN_BAND - 1) / RUN_BAND;
    static_assert(RUN_QUADS == 5 && RUN_BAND == 512, "piece -> row uses 16-bit reciprocals of 3 and 5");
    static_assert(PIECES * (RUN_BAND / RUN_QUADS) >= RUN_WIN_ROWS, "rounds cover the window");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.src + (int64_t)zscan * p.src_rows * p.src_wpr), /*stride*/ 0,
        (int)((uint32_t)p.src_rows * (uint32_t)p.src_wpr * 4u), 0x00020000);
    // (row, first word) of the thread's round-0 piece.  Recomputed where it is used from an opaque
    // copy of tid: cached or loop-hoisted offsets cost more registers than the kernel has, and a
    // spill reload inside the fetch sequence would wait for the loads already in flight.
    auto piece0 = [&](const RunGeom &q, int &row0, int &w4, int &rpr) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const uint32_t magic = q.nq == 3 ? 21846u : (q.nq == 4 ? 16384u : 13108u);  // 65536 / nq, rounded up
        rpr = q.nq == 3 ? 170 : (q.nq == 4 ? 128 : 102);
        row0 = (int)(((uint32_t)t * magic) >> 16);
        w4 = 4 * (t - row0 * q.nq);
    };
    // The loads of a window are issued as soon as a wave has finished the previous band (before the
    // barrier: they do not touch LDS) and committed after it.  They are NOT held in registers across a
    // compute phase: the merge loop runs at the edge of the 128-VGPR budget of 4 waves per SIMD, and every
    // variant that kept 16-24 piece registers live through it ended with the allocator spilling them
    // (round 2: six variants, 16-78 spilled VGPRs); the second workgroup of the CU covers the fetch instead.
    // Rounds 0..3 (all of a 3-piece window: 69 % of the bands at C2) fly across the loop-top barrier; the
    // rounds of wider windows are loaded inside the commit, so that no more than 16 piece registers are
    // ever live at the loop's back edge (with 24 the allocator spilled two pieces around the whole loop).
    constexpr int EARLY = 4;
    static_assert(EARLY * 170 >= RUN_WIN_ROWS, "the early rounds cover a 3-piece window");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pre[EARLY];
    auto fetch = [&](const RunGeom &q) {
        if (!q.fits) return;
        int row0, w4, rpr;
        piece0(q, row0, w4, rpr);
        const int w = q.wxw + w4;
        const bool in_w = row0 < rpr && w >= 0 && w + 3 < p.src_wpr;
        // out-of-image threads: 2^31 stays out of range after adding any round's stride (< 2^31)
        const uint32_t voff = in_w ? (uint32_t)(((q.wy0 + row0) * p.src_wpr + w) * 4) : 0x80000000u;
        const uint32_t stride = (uint32_t)(rpr * p.src_wpr * 4);
#pragma unroll
        for (int n = 0; n < EARLY; n++) {
            if (n * rpr >= q.nrows) break;  // wave-uniform: this round has no rows
            pre[n] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + n * stride), 0, 0);
        }
    };
    auto commit = [&](const RunGeom &q) {
        int row0, w4, rpr;
        piece0(q, row0, w4, rpr);
        char *d0 = lds + RUN_WIN_OFS + row0 * RUN_PITCHB + w4 * 4;
        const int rows_left = row0 < rpr ? q.nrows - row0 : 0;
        const int dstride = rpr * RUN_PITCHB;
        u32x4 late[PIECES - EARLY];
        if (EARLY * rpr < q.nrows) {  // wave-uniform: a 4- or 5-piece window has late rounds
            const int w = q.wxw + w4;
            const bool in_w = row0 < rpr && w >= 0 && w + 3 < p.src_wpr;
            const uint32_t voff = in_w ? (uint32_t)(((q.wy0 + row0) * p.src_wpr + w) * 4) : 0x80000000u;
            const uint32_t stride = (uint32_t)(rpr * p.src_wpr * 4);
#pragma unroll
            for (int n = EARLY; n < PIECES; n++) {
                if (n * rpr >= q.nrows) break;
                late[n - EARLY] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + n * stride), 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < PIECES; n++) {
            if (n * rpr >= q.nrows) break;
            if (n * rpr < rows_left) {
                const u32x4 v = n < EARLY ? pre[n < EARLY ? n : 0] : late[n >= EARLY ? n - EARLY : 0];
                uint32_t *d = (uint32_t *)(d0 + n * dstride);
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
        }
    };

    // ---- prologue: the first band's fetch goes out before the run tables are staged
    int2_t cn0, cn1;
    corners(0, cn0, cn1);
    int2_t rt = RT[min(wave * 64 + lane, p.NR - 1)];
    // run tables of this block's words (16-byte copies; words past the end of the row repeat the
    // last real word so that every lane address stays meaningful)
    constexpr int W16 = RUN_TAB_BYTES / 16;
    constexpr int TP = (RUN_K * W16 + RUN_BAND - 1) / RUN_BAND;
    uint4 tv[TP];
    {
        const uint4 *src = (const uint4 *)(p.tabs + ((int64_t)a * p.NW + w0));
#pragma unroll
        for (int n = 0; n < TP; n++) {
            const int i = tid + n * RUN_BAND;
            const int k = i / W16, c = i - k * W16;
            // words past the end of the row: a copy of the last real word (addresses stay meaningful) whose
            // level masks are EMPTY, so they add nothing to any count
            tv[n] = (i < RUN_K * W16 && !(k >= kw && c < RUN_TUPX_OFS / 16)) ? src[min(k, kw - 1) * W16 + c]
                                                                            : make_uint4(0, 0, 0, 0);
        }
    }
    int2 mv = make_int2(0, 0);  // (ca0, cb0) of word tid: read back from LDS as a broadcast
    if (tid < RUN_K) {
        const RunMeta *__restrict__ mt = p.meta + ((int64_t)a * p.NW + w0);
        const int kk = min(tid, kw - 1);
        mv = make_int2(mt[kk].ca0, mt[kk].cb0);
    }
    RunGeom cur = geometry(cn0, cn1);
    if (!(RUN_DBG(p) & 2)) fetch(cur);
    corners(RUN_BAND, cn0, cn1);  // next band's corners: in flight until the end of the first iteration
    {
        uint4 *dst = (uint4 *)(lds + RUN_TABS_OFS);
#pragma unroll
        for (int n = 0; n < TP; n++) {
            const int i = tid + n * RUN_BAND;
            if (i < RUN_K * W16) dst[i] = tv[n];
        }
    }
    if (tid < RUN_K) *(int2 *)(lds + RUN_META_OFS + tid * 8) = mv;
    if (tid < RUN_K * 32) colacc[tid] = 0;

    uint32_t c0[RUN_K], c1[RUN_K], c2[RUN_K];
#pragma unroll
    for (int k = 0; k < RUN_K; k++) c0[k] = c1[k] = c2[k] = 0;

    int bands_pending = 0;
    RUN_STAMP(0)
    for (int yb = 0; yb < p.NR; yb += RUN_BAND) {
        const int r = yb + wave * 64 + lane;
        __syncthreads();  // previous band's readers are done (first pass: tables, meta, colacc staged)
        if (cur.fits) {
            if (!(RUN_DBG(p) & 2)) commit(cur);
        } else if (tid == 0) {
            guard[a] = 1;
        }
        __syncthreads();
        RUN_STAMP(1)
        const RunGeom now = cur;
        const int2_t rt_now = rt;
        const bool more = yb + RUN_BAND < p.NR;
        if (more) {
            cur = geometry(cn0, cn1);
            corners(yb + 2 * RUN_BAND, cn0, cn1);
            rt = RT[min(r + RUN_BAND, p.NR - 1)];
        }
        RUN_STAMP(2)
        // rows past the end of the image are masked off by EXEC for the whole compute phase (their
        // counters must not move); the band loop's barriers are outside this branch
        if (now.fits && r < p.NR && !(RUN_DBG(p) & 1)) {
            const int rx = rt_now.x - (now.wxw << 15);  // window-local fixed point
            const int ry = rt_now.y - (now.wy0 << 10);
            const uint32_t cnt = band_words_s(rx, ry, c0, c1, c2, nlev_blk, lagbits);
            out[r] = (uint16_t)cnt;
        }
        RUN_STAMP(3)
        // the 3-plane counters hold at most 7 rows per lane: reduce them across lanes in time
        if (++bands_pending == RUN_FLUSH_BANDS || !more) {
            bands_pending = 0;
            if (!(RUN_DBG(p) & 4)) flush_columns(lds, c0, c1, c2, colacc, tid);
            RUN_STAMP(4)
        }
        if (more && !(RUN_DBG(p) & 2)) fetch(cur);  // next window: in flight across the loop-top barrier
    }
    __syncthreads();
    // column counts of this block's (up to) 256 columns over ALL rows: one plain store each
    if (tid < RUN_K * 32 && c_first + tid < p.NC)
        vproj[((int64_t)zscan * p.A + a) * p.NC + c_first + tid] = colacc[tid];
    if (stamp && tid == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&g_run_stamps[i], st_acc[i]);
        atomicAdd(&g_run_stamps[5], run_clock() - st_t0);
        atomicAdd(&g_run_stamps[6], 1ull);
    }
#undef RUN_STAMP
}

hipError_t debug_runs_stamps(unsigned long long out[8], bool reset)
{
#ifndef OMR_RUNS_DEBUG
    (void)out;
    (void)reset;
    return hipErrorNotSupported;  // release build: no stamp variant is compiled in
#endif
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_run_stamps), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_run_stamps), z, sizeof z);
    }
    return e;
}

hipError_t launch_runs(const RunPass &p0, const int32_t *d_list, int n_list, int32_t *d_guard, uint32_t *d_vproj,
                       hipStream_t s)
{
    if (n_list <= 0) return hipSuccess;
    RunPass p = p0;
#ifdef OMR_RUNS_DEBUG
    {
        const char *e = getenv("OMR_RUNS_DBG");
        p.dbg = e ? atoi(e) : 0;
    }
    auto *kern = (p.dbg & 8) ? runs_kernel<true> : runs_kernel<false>;
#else
    p.dbg = 0;
    auto *kern = runs_kernel<false>;
#endif
    // per device and idempotent; cheap enough to repeat (the batch entry points use every device)
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RUN_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (p.scans < 1) p.scans = 1;
    hipLaunchKernelGGL(kern, dim3(p.scans, p.G, n_list), dim3(RUN_BAND), RUN_LDS_BYTES, s, p, d_list, d_guard, d_vproj);
    return hipGetLastError();
}

// proj[a][r] = sum_g part[a][g][r] for the listed candidates
__global__ __launch_bounds__(256) void fold_parts_kernel(const uint16_t *__restrict__ part, int G, int NR,
                                                         const int32_t *__restrict__ list,
                                                         uint32_t *__restrict__ proj, int A)
{
    part += (int64_t)blockIdx.z * A * G * NR;  // blockIdx.z = scan of the launch
    proj += (int64_t)blockIdx.z * A * NR;
    const int a = list[blockIdx.y];
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= NR) return;
    uint32_t s = 0;
    for (int g = 0; g < G; g++) s += part[((int64_t)a * G + g) * NR + r];
    proj[(int64_t)a * NR + r] = s;
}

hipError_t launch_fold_parts(const uint16_t *d_part, int G, int NR, const int32_t *d_list, int n_list,
                             uint32_t *d_proj, hipStream_t s, int scans, int A)
{
    if (n_list <= 0 || scans <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_parts_kernel, dim3((NR + 255) / 256, n_list, scans), dim3(256), 0, s, d_part, G, NR, d_list,
                       d_proj, A);
    return hipGetLastError();
}

}  // namespace omr
