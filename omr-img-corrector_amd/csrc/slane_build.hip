// slane_build.hip -- the scan-lane sweep's programs generated ON THE DEVICE (slane.hpp).
//
// The same enumeration as the host generator (slane_plan.cpp, which stays as the reference implementation behind
// omr_slane_strip_program and omr_batch_lanes_check_programs): every destination word's runs straight from the integer
// tables of warpAffine (the context's own device tables: adelta / bdelta / X0 / Y0, kernels.hip tables_kernel), the
// column base and the live range of every source row of a strip, the greedy fetch schedule of the register ring, the
// packed segment words.  15 600 strips of an A4 sweep are 7 GB of programs: two seconds on sixteen host threads plus
// the upload, a few tens of milliseconds here.
//
//   slane_scan_kernel    thread = destination word (task, row, k): its runs -> most segments of the strip,
//                        per source row of the strip: least / greatest word column, first / last destination row
//   slane_fill_kernel    every stream starts as the empty program (white words, nothing fetched, dummy commits)
//   slane_sched_kernel   thread = strip: the fetch schedule (sequential by construction: a row's loads take the latest
//                        free slots before its first use, bounded by the ring's reuse distance) and the commit lists
//   slane_words_kernel   thread = destination word: the runs again, now encoded against the strip's column bases
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "slane.hpp"

namespace omr {

namespace {

constexpr int SB_MAXSEG = 8;

__device__ __forceinline__ int sb_floor_div32(int v) { return v >= 0 ? v >> 5 : -((-v + 31) >> 5); }

// slane_plan.cpp word_segments(): the runs of destination word w of a row in increasing bit order, each handed to
// emit(j, s, src, len) when it closes (s = source row + gy, -1 = white; src = source column of its first bit).
// Returns their number, -1 for more than SB_MAXSEG.
template <class F>
__device__ __forceinline__ int sb_word_runs(const SlaneGeom &g, const int32_t *__restrict__ ad, const int32_t *__restrict__ bd,
                                            int32_t X0, int32_t Y0, int w, F emit)
{
    int n = 0, cur_s = -2, cur_base = 0, cur_src = 0, cur_len = 0;
    for (int i = 0; i < 32; i++) {
        const int x = 32 * w - g.off + i;
        int sy = -1, base = 0;
        if (w < g.NWd && x >= 0 && x < g.cols) {
            const int sx = (X0 + ad[x]) >> 10, yy = (Y0 + bd[x]) >> 10;
            if (sx >= -32 * (g.gx - 1) && sx < g.cols + 32 * (g.gx - 1) && yy >= -g.gy && yy < g.rows + g.gy) sy = yy + g.gy, base = sx - i;
        }
        if (cur_len > 0 && cur_s == sy && (sy < 0 || cur_base == base)) {
            cur_len++;
        } else {
            if (cur_len > 0) {
                if (n == SB_MAXSEG) return -1;
                emit(n, cur_s, cur_src, cur_len);
                n++;
            }
            cur_s = sy, cur_base = base, cur_src = sy < 0 ? 0 : base + i, cur_len = 1;
        }
    }
    if (cur_len > 0) {
        if (n == SB_MAXSEG) return -1;
        emit(n, cur_s, cur_src, cur_len);
        n++;
    }
    return n;
}

__global__ __launch_bounds__(256) void slane_init_kernel(int64_t n, int32_t *cmin, int32_t *cmax, int32_t *first, int32_t *last)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        cmin[i] = 32767;
        cmax[i] = -32768;
        first[i] = 0x7fffffff;
        last[i] = -1;
    }
}

// grid (tasks, ceil(rows / 128)), task = candidate * NS + strip
__global__ __launch_bounds__(256) void slane_scan_kernel(SlaneBuild b)
{
    const SlaneGeom &g = b.g;
    const int task = blockIdx.x, a = task / g.NS, strip = task - a * g.NS;
    const int r = blockIdx.y * 128 + (threadIdx.x >> 1), k = threadIdx.x & 1;
    if (r >= g.rows) return;
    const int32_t *ad = b.adelta + (int64_t)a * g.cols, *bd = b.bdelta + (int64_t)a * g.cols;
    const int2_t xy = b.xy0[(int64_t)a * g.rows + r];
    const int64_t tb = (int64_t)task * g.rowsG;
    const int n = sb_word_runs(g, ad, bd, xy.x, xy.y, strip * SL_K + k, [&](int, int s, int src, int len) {
        if (s < 0) return;
        const int c = sb_floor_div32(src), sh = src - 32 * c, chi = sh + len > 32 ? c + 1 : c;
        atomicMin(&b.cmin[tb + s], c);
        atomicMax(&b.cmax[tb + s], chi);
        atomicMin(&b.first[tb + s], r);
        atomicMax(&b.last[tb + s], r);
    });
    atomicMax(&b.most[task], n < 0 ? 99 : n);
}

// grid (tasks + 1, ceil(nrec / 256)): stream `tasks` is the null program (class 0)
__global__ __launch_bounds__(256) void slane_fill_kernel(SlaneBuild b, int ntasks)
{
    const int t = blockIdx.x, q = blockIdx.y * 256 + threadIdx.x;
    if (q >= b.nrec) return;
    const int cls = t < ntasks ? b.cls[t] : 0;
    uint32_t *seg = b.prog + (t < ntasks ? b.seg_off[t] : b.null_seg), *fet = b.prog + (t < ntasks ? b.fet_off[t] : b.null_fet);
    const int S = slane_slots(cls), RD = SL_K * S;
    const uint32_t white = ((uint32_t)(SL_ZERO + 1) << 5) | SL_PK_MODE | (1u << SL_NSHIFT), pad = ((uint32_t)SL_ZERO << 5) | SL_PK_MODE;
    for (int k = 0; k < SL_K; k++)
        for (int j = 0; j < S; j++) seg[(int64_t)q * RD + k * S + j] = j == 0 ? white : pad;
    const uint32_t nocommit = (uint32_t)SL_DUMMY | SL_COMMIT_MODE;
    uint32_t *f = fet + (int64_t)q * SL_FREC;
    f[0] = f[1] = 0u;
    f[2] = nocommit | (nocommit << 16);
    f[3] = 1u;  // turn header (slane_turns_kernel writes the real ones)
}

// thread = task.  used[task][nrec] starts 0, freg[task][nrec][4] starts SL_DUMMY (set by the host with hipMemset)
__global__ __launch_bounds__(64) void slane_sched_kernel(SlaneBuild b, int ntasks)
{
    const int task = blockIdx.x * 64 + threadIdx.x;
    if (task >= ntasks) return;
    const SlaneGeom &g = b.g;
    const int64_t tb = (int64_t)task * g.rowsG;
    uint8_t *used = b.used + (int64_t)task * b.nrec, *freg = b.freg + (int64_t)task * b.nrec * SL_FETCH;  // (freg: SL_PAIRS of the 4 bytes per record)
    uint32_t *fet = b.prog + b.fet_off[task];
    bool ok = true;
    for (int s = 0; s < g.rowsG && ok; s++) {
        const int la = b.last[tb + s];
        if (la < 0) continue;
        const int lo = b.cmin[tb + s], hi = b.cmax[tb + s], ncols = hi - lo + 1;
        if (ncols > SL_RING_COLS || lo < -g.gx || hi >= g.NW + g.gx) {
            ok = false;
            break;
        }
        int lb = -SL_PRE;
        if (s >= SL_RING_ROWS) {
            const int lp = b.last[tb + s - SL_RING_ROWS];
            if (lp >= 0) lb = max(lb, lp - SL_AHEAD + 1);
        }
        int rec = b.first[tb + s] - SL_AHEAD;
        for (int pr = (ncols + 1) / 2 - 1; pr >= 0; pr--) {  // word columns 2 pr, 2 pr + 1 of the row: one pair (slane_plan.cpp)
            while (rec >= lb && used[rec + SL_PRE] == SL_PAIRS) rec--;
            if (rec < lb) {
                ok = false;
                break;
            }
            const int q = rec + SL_PRE, u = used[q];
            freg[(int64_t)q * SL_FETCH + u] = (uint8_t)((s & (SL_RING_ROWS - 1)) * SL_RING_COLS + 2 * pr);
            // g.entry(s - gy, lo + 2 pr); a row of three columns loads its second pair as (column 1, column 2): slane_plan.cpp
            const int64_t e0 = 1 + (int64_t)s * g.colsG + (lo + 2 * pr - (ncols == 3 && pr == 1 ? 1 : 0) + g.gx);
            fet[(int64_t)q * SL_FREC + u] = (uint32_t)(e0 << 8);
            used[q] = (uint8_t)(u + 1);
        }
    }
    if (!ok) {
        atomicExch(b.bad, 1);
        return;
    }
    for (int q = SL_AHEAD; q < b.nrec; q++) {  // what row q commits = what row q - SL_AHEAD fetched
        const uint8_t *fr = &freg[(int64_t)(q - SL_AHEAD) * SL_FETCH];
        fet[(int64_t)q * SL_FREC + 2] = ((uint32_t)fr[0] | SL_COMMIT_MODE) | (((uint32_t)fr[1] | SL_COMMIT_MODE) << 16);
    }
}

// thread = (task, turn): the turn header -- the most segments any word of the turn's SL_TURN records needs -- read back from the
// encoded segment words (after slane_words_kernel)
__global__ __launch_bounds__(256) void slane_turns_kernel(SlaneBuild b, int ntasks)
{
    const int nturns = (b.nrec + SL_TURN - 1) / SL_TURN;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)ntasks * nturns) return;
    const int task = (int)(i / nturns), q0 = (int)(i - (int64_t)task * nturns) * SL_TURN;
    const int S = slane_slots(b.cls[task]), RD = SL_K * S;
    const uint32_t *seg = b.prog + b.seg_off[task];
    uint32_t *fet = b.prog + b.fet_off[task];
    uint32_t most = 1;
    for (int q = q0; q < q0 + SL_TURN && q < b.nrec; q++)
        for (int k = 0; k < SL_K; k++) most = max(most, (seg[(int64_t)q * RD + k * S] >> SL_NSHIFT) & 15u);
    for (int q = q0; q < q0 + SL_TURN && q < b.nrec; q++) fet[(int64_t)q * SL_FREC + 3] = most;
}

// grid (tasks, ceil(rows / 128))
__global__ __launch_bounds__(256) void slane_words_kernel(SlaneBuild b)
{
    const SlaneGeom &g = b.g;
    const int task = blockIdx.x, a = task / g.NS, strip = task - a * g.NS;
    const int r = blockIdx.y * 128 + (threadIdx.x >> 1), k = threadIdx.x & 1;
    if (r >= g.rows) return;
    const int32_t *ad = b.adelta + (int64_t)a * g.cols, *bd = b.bdelta + (int64_t)a * g.cols;
    const int2_t xy = b.xy0[(int64_t)a * g.rows + r];
    const int64_t tb = (int64_t)task * g.rowsG;
    const int S = slane_slots(b.cls[task]), RD = SL_K * S;
    uint32_t *w = b.prog + b.seg_off[task] + (int64_t)(r + SL_PRE) * RD + k * S;
    // (slane_plan.cpp: the first segment is read straight into the word, a funnel shift's amount rides in the slot before it;
    // the first dword is written last, with the count)
    uint32_t w0 = 0, prev = 0;
    const int n = sb_word_runs(g, ad, bd, xy.x, xy.y, strip * SL_K + k, [&](int j, int s, int src, int len) {
        uint32_t idx = SL_ZERO, sh = 0;
        if (s >= 0) {
            const int c = sb_floor_div32(src);
            sh = (uint32_t)(src - 32 * c);
            const int lo = b.cmin[tb + s];
            idx = (uint32_t)((s & (SL_RING_ROWS - 1)) * SL_RING_COLS +
                             slane_ring_register(c - lo, b.cmax[tb + s] - lo + 1, sh + (uint32_t)len > 32u));
        }
        if (j == 0) slane_first_segment(idx, sh, len);
        const uint32_t pk = sh | (idx << 5) | SL_PK_MODE;
        if (j > 0) {
            prev |= (uint32_t)len << SL_QSHIFT;
            if (j == 1) w0 = prev;
            else if (j - 1 < S) w[j - 1] = prev;
        }
        prev = pk;
        if (j == 0) w0 = pk;
    });
    if (n < 1 || n > S) {
        atomicExch(b.bad, 1);
        return;
    }
    if (n > 1) w[n - 1] = prev;
    w[0] = w0 | ((uint32_t)n << SL_NSHIFT);
}

}  // namespace

hipError_t launch_slane_build_scan(const SlaneBuild &b, int ntasks, hipStream_t s)
{
    const int64_t n = (int64_t)ntasks * b.g.rowsG;
    hipLaunchKernelGGL(slane_init_kernel, dim3(4096), dim3(256), 0, s, n, b.cmin, b.cmax, b.first, b.last);
    hipLaunchKernelGGL(slane_scan_kernel, dim3(ntasks, (b.g.rows + 127) / 128), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_slane_build_emit(const SlaneBuild &b, int ntasks, hipStream_t s)
{
    hipLaunchKernelGGL(slane_fill_kernel, dim3(ntasks + 1, (b.nrec + 255) / 256), dim3(256), 0, s, b, ntasks);
    hipLaunchKernelGGL(slane_sched_kernel, dim3((ntasks + 63) / 64), dim3(64), 0, s, b, ntasks);
    hipLaunchKernelGGL(slane_words_kernel, dim3(ntasks, (b.g.rows + 127) / 128), dim3(256), 0, s, b);
    {
        const int64_t n = (int64_t)ntasks * ((b.nrec + SL_TURN - 1) / SL_TURN);
        hipLaunchKernelGGL(slane_turns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b, ntasks);
    }
    return hipGetLastError();
}

}  // namespace omr
