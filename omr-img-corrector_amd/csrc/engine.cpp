// engine.cpp -- plans, scratch and the enqueue sequence behind the C ABI.
// Built with hipcc -ffp-contract=off: the host-side matrix arithmetic below must round exactly
// like OpenCV's (no FMA contraction).
#include "engine.hpp"

#include <algorithm>
#include <map>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/omrdeskew.h"

namespace omr {

static thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int fail_gpu(const char *what, hipError_t e)
{
    return fail(OMR_ERR_GPU, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}

const char *last_error() { return g_err.c_str(); }
void clear_error() { g_err.clear(); }

// ---- device buffer cache ---------------------------------------------------------------------
namespace {
thread_local hipStream_t t_pool_stream = nullptr;
thread_local bool t_pool_on = false;

struct BlockCache {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;  // size -> block
    size_t cached = 0;
};
BlockCache &cache_of(int dev)  // callers check 0 <= dev < 16
{
    static BlockCache caches[16];
    return caches[dev];
}
size_t cache_limit()
{
    static const size_t lim = [] {
        const char *e = getenv("OMR_POOL_MB");
        return (size_t)(e ? atoll(e) : 2048) << 20;
    }();
    return lim;
}
size_t round_block(size_t n)
{
    const size_t g = n >= ((size_t)1 << 20) ? ((size_t)1 << 20) : (size_t)4096;
    return (n + g - 1) / g * g;
}
}  // namespace

// ---- per-call stream + pinned staging buffer: leased from a bounded process-wide pool ------------
// The host's worker "pool" (thread_pool.rs:41-88) spawns a fresh OS thread per task, so nothing may be owned by
// a thread: a thread_local stream / staging buffer per caller would leak one HIP stream and one pinned block of
// the rotated sheet's size per file.  A call leases a slot (its stream keeps the slot's pinned block warm) and
// returns it; idle slots beyond the caps below are destroyed.
namespace {
const size_t kStageMax = (size_t)64 << 20;      // larger results go in pieces
const int kMaxDevices = 16;
const int kIdleSlotsPerDevice = 32;             // streams kept for reuse
const size_t kIdlePinnedBytes = (size_t)512 << 20;  // pinned memory kept for reuse, per device

struct SlotPool {
    std::mutex mu;
    std::vector<CallSlot *> idle;
    size_t idle_pinned = 0;
    int live = 0;  // slots that exist (leased + idle): what the leak test watches
};
SlotPool &slots_of(int dev)
{
    static SlotPool pools[kMaxDevices];
    return pools[dev];
}
thread_local CallSlot *t_slot = nullptr;  // innermost lease of the calling thread

void destroy_slot(CallSlot *c)
{
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int stage_reserve(CallSlot *c, size_t n)
{
    n = n > kStageMax ? kStageMax : n;
    if (c->cap >= n) return 0;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr;
    c->cap = 0;
    const size_t want = (n + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipHostMalloc(&c->pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        c->pinned = nullptr;
        return fail_gpu("hipHostMalloc (download staging)", e);
    }
    c->cap = want;
    return 0;
}
}  // namespace

int lease_call_slot(CallSlot **out)
{
    int dev = 0;
    OMR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(OMR_ERR_BADARG, "device %d: at most %d devices are supported", dev, kMaxDevices);
    SlotPool &pool = slots_of(dev);
    CallSlot *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        if (!pool.idle.empty()) {
            c = pool.idle.back();
            pool.idle.pop_back();
            pool.idle_pinned -= c->cap;
        }
    }
    if (!c) {
        c = new CallSlot;
        c->dev = dev;
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail_gpu("hipStreamCreateWithFlags", e);
        }
        std::lock_guard<std::mutex> lk(pool.mu);
        pool.live++;
    }
    c->outer = t_slot;
    t_slot = c;
    *out = c;
    return 0;
}

void return_call_slot(CallSlot *c)
{
    if (!c) return;
    if (t_slot == c) t_slot = c->outer;
    c->outer = nullptr;
    SlotPool &pool = slots_of(c->dev);
    bool keep;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        keep = (int)pool.idle.size() < kIdleSlotsPerDevice;
        if (keep && pool.idle_pinned + c->cap > kIdlePinnedBytes) {
            // keep the stream, give the pinned block back
            (void)hipHostFree(c->pinned);
            c->pinned = nullptr;
            c->cap = 0;
        }
        if (keep) {
            pool.idle.push_back(c);
            pool.idle_pinned += c->cap;
        } else {
            pool.live--;
        }
    }
    if (!keep) {
        (void)hipStreamSynchronize(c->stream);
        destroy_slot(c);
    }
}

void call_slot_stats(int dev, int *live, int *idle, size_t *idle_pinned)
{
    if (live) *live = 0;
    if (idle) *idle = 0;
    if (idle_pinned) *idle_pinned = 0;
    if (dev < 0 || dev >= kMaxDevices) return;  // no pool exists for such a device (lease_call_slot refuses it)
    SlotPool &pool = slots_of(dev);
    std::lock_guard<std::mutex> lk(pool.mu);
    if (live) *live = pool.live;
    if (idle) *idle = (int)pool.idle.size();
    if (idle_pinned) *idle_pinned = pool.idle_pinned;
}

// RAII lease for copies made outside any entry point's own lease
namespace {
struct SlotLease {
    CallSlot *c = nullptr;
    bool mine = false;
    int take(hipStream_t s)
    {
        if (t_slot && t_slot->stream == s) {
            c = t_slot;
            return 0;
        }
        int rc = lease_call_slot(&c);
        mine = rc == 0;
        return rc;
    }
    ~SlotLease()
    {
        if (mine) return_call_slot(c);
    }
};
}  // namespace

int staged_d2h(void *dst, const void *d_src, size_t bytes, hipStream_t s)
{
    if (bytes < ((size_t)256 << 10)) {  // small results: the direct copy is as fast
        OMR_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
        OMR_HIP(hipStreamSynchronize(s));
        return 0;
    }
    SlotLease L;
    int rc = L.take(s);
    if (rc) return rc;
    if ((rc = stage_reserve(L.c, bytes))) return rc;
    for (size_t off = 0; off < bytes; off += L.c->cap) {
        const size_t n = bytes - off < L.c->cap ? bytes - off : L.c->cap;
        OMR_HIP(hipMemcpyAsync(L.c->pinned, (const char *)d_src + off, n, hipMemcpyDeviceToHost, s));
        OMR_HIP(hipStreamSynchronize(s));
        memcpy((char *)dst + off, L.c->pinned, n);
    }
    return 0;
}

int staged_d2h_2d(void *dst, size_t dst_step, const void *d_src, size_t row_bytes, size_t rows, hipStream_t s)
{
    if (dst_step == row_bytes) return staged_d2h(dst, d_src, row_bytes * rows, s);
    if (row_bytes == 0 || rows == 0) return 0;
    SlotLease L;
    int rc = L.take(s);
    if (rc) return rc;
    if ((rc = stage_reserve(L.c, row_bytes * rows))) return rc;
    if (L.c->cap < row_bytes && (rc = stage_reserve(L.c, row_bytes))) return rc;
    if (L.c->cap < row_bytes) return fail(OMR_ERR_BADARG, "image row of %zu bytes exceeds the staging buffer", row_bytes);
    const size_t per = L.c->cap / row_bytes;  // rows per piece
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t nr = rows - r0 < per ? rows - r0 : per;
        OMR_HIP(hipMemcpyAsync(L.c->pinned, (const char *)d_src + r0 * row_bytes, nr * row_bytes, hipMemcpyDeviceToHost, s));
        OMR_HIP(hipStreamSynchronize(s));
        for (size_t r = 0; r < nr; r++) memcpy((char *)dst + (r0 + r) * dst_step, (const char *)L.c->pinned + r * row_bytes, row_bytes);
    }
    return 0;
}

PoolScope::PoolScope(hipStream_t stream) : prev_(t_pool_stream), prev_on_(t_pool_on)
{
    t_pool_stream = stream;
    t_pool_on = true;
}
PoolScope::~PoolScope()
{
    t_pool_stream = prev_;
    t_pool_on = prev_on_;
}

NoPoolScope::NoPoolScope() : prev_(t_pool_stream), prev_on_(t_pool_on)
{
    t_pool_stream = nullptr;
    t_pool_on = false;
}
NoPoolScope::~NoPoolScope()
{
    t_pool_stream = prev_;
    t_pool_on = prev_on_;
}

hipError_t DevBuf::alloc(size_t n)
{
    release();
    if (n == 0) n = 16;
    if (t_pool_on && cache_limit() > 0) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
            const size_t want = round_block(n);
            BlockCache &c = cache_of(dev);
            {
                std::lock_guard<std::mutex> lk(c.mu);
                auto it = c.free_blocks.lower_bound(want);
                if (it != c.free_blocks.end() && it->first <= 2 * want) {
                    p = it->second;
                    cap_ = it->first;
                    c.cached -= cap_;
                    c.free_blocks.erase(it);
                }
            }
            if (!p) {
                hipError_t e = hipMalloc(&p, want);
                if (e != hipSuccess) {
                    p = nullptr;
                    return e;
                }
                cap_ = want;
            }
            bytes = n;
            owner_ = t_pool_stream;
            dev_ = dev;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    else p = nullptr;
    return e;
}

void DevBuf::release()
{
    if (p && dev_ >= 0) {
        // the owner's queued work may still use the block: drain that one stream, not the device
        bool keep = hipStreamSynchronize(owner_) == hipSuccess;
        BlockCache &c = cache_of(dev_);
        if (keep) {
            std::lock_guard<std::mutex> lk(c.mu);
            if (c.cached + cap_ <= cache_limit()) {
                c.free_blocks.emplace(cap_, p);
                c.cached += cap_;
            } else {
                keep = false;
            }
        }
        if (!keep) (void)hipFree(p);
    } else if (p) {
        (void)hipFree(p);
    }
    p = nullptr;
    bytes = 0;
    owner_ = nullptr;
    cap_ = 0;
    dev_ = -1;
}

// ---- OpenCV 4.6.0 geometry on the host ------------------------------------------------------

// getRotationMatrix2D_; call sites transfer.rs:475, omr.rs:159-163.
void rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6])
{
    const double CV_PI_ = 3.1415926535897932384626433832795;
    double angle = angle_deg * (CV_PI_ / 180);
    double alpha = cos(angle) * scale;
    double beta = sin(angle) * scale;
    M[0] = alpha;
    M[1] = beta;
    M[2] = (1 - alpha) * (double)cx - beta * (double)cy;
    M[3] = -beta;
    M[4] = alpha;
    M[5] = beta * (double)cx + (1 - alpha) * (double)cy;
}

// cv::warpAffine's in-place inversion, same operation order.
void invert_affine(const double Min[6], double M[6])
{
    memcpy(M, Min, 6 * sizeof(double));
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11;
    M[1] *= -D;
    M[3] *= -D;
    M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1;
    M[5] = b2;
}

// `(max_angle as f64 / step) as u16` (projection.rs:36): truncating, saturating, NaN -> 0.
int candidate_count(uint16_t max_angle, double step, int *N_out)
{
    double q = (double)max_angle / step;
    int N;
    if (!(q == q) || q <= 0) N = 0;
    else if (q >= 65535.0) N = 65535;
    else N = (int)q;
    if (N_out) *N_out = N;
    return 2 * N;
}

// transfer.rs:473-475 per candidate `deg as f64 * step` for deg in -N..N (projection.rs:38,50).
void sweep_matrices(int rows, int cols, int N, double step, double scale, double *M_out)
{
    const float cx = (float)cols / 2.0f, cy = (float)rows / 2.0f;
    for (int i = 0; i < 2 * N; i++) rotation_matrix_2d(cx, cy, (double)(i - N) * step, scale, M_out + 6 * (size_t)i);
}

// ---- tables ---------------------------------------------------------------------------------

static const int kSlabWords = 1024;  // must match LDS_SLAB_WORDS in kernels.hip
static const int kMaxWinWords = 32;

// Largest destination-row count per staged window whose source bounding box fits the wave's
// LDS slab.  The fixed-point map deviates from the real affine map by < 1 px per axis (two
// roundings of <= 0.5/1024 px plus the floor), which the +2 margins cover; the kernel re-checks
// with the exact integer corners and gathers from global memory if a window were ever larger.
static LdsTile size_tile(const double Mi[6])
{
    static const int cand[] = {256, 192, 128, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2, 1};
    LdsTile t{0, 0, 0};
    for (int R : cand) {
        double xspan = 63.0 * fabs(Mi[0]) + (double)(R - 1) * fabs(Mi[1]);
        double yspan = 63.0 * fabs(Mi[3]) + (double)(R - 1) * fabs(Mi[4]);
        if (!(xspan < 1e6) || !(yspan < 1e6)) continue;
        int words = (int)((xspan + 2.0) / 32.0) + 2;
        int rows = (int)(yspan + 2.0) + 2;
        if (words <= kMaxWinWords && (long long)rows * (words | 1) <= kSlabWords) {
            t.rows_per_tile = R;
            t.win_words = words;
            t.win_rows = rows;
            return t;
        }
    }
    return t;
}

int SweepTables::create(int rows, int cols, const double *fwd_M, int A, int dev)
{
    // OpenCV's remap asserts every dimension < SHRT_MAX
    if (rows <= 0 || cols <= 0 || rows >= 32767 || cols >= 32767)
        return fail(OMR_ERR_ASSERT, "bad image size %dx%d (need 0 < dim < 32767)", cols, rows);
    if (A <= 0 || !fwd_M) return fail(OMR_ERR_BADARG, "need at least one candidate matrix");
    int ndev = 0;
    OMR_HIP(hipGetDeviceCount(&ndev));
    if (dev < 0 || dev >= ndev) return fail(OMR_ERR_BADARG, "device %d out of range (%d visible)", dev, ndev);
    OMR_HIP(hipSetDevice(dev));
    device = dev;
    dims.rows = rows;
    dims.cols = cols;
    dims.A = A;
    dims.wpr = ((cols + 31) / 32 + 3) & ~3;  // rows are 16-byte multiples (aligned 4-word loads)

    host_minv.resize((size_t)A * 6);
    std::vector<LdsTile> ht((size_t)A);
    lds_ok = true;
    max_rows_per_tile = 0;
    for (int a = 0; a < A; a++) {
        for (int k = 0; k < 6; k++)
            if (!isfinite(fwd_M[6 * (size_t)a + k])) return fail(OMR_ERR_BADARG, "matrix %d is not finite", a);
        invert_affine(fwd_M + 6 * (size_t)a, &host_minv[6 * (size_t)a]);
        ht[a] = size_tile(&host_minv[6 * (size_t)a]);
        if (ht[a].rows_per_tile == 0) lds_ok = false;
        if (ht[a].rows_per_tile > max_rows_per_tile) max_rows_per_tile = ht[a].rows_per_tile;
    }
    OMR_HIP(minv.alloc(sizeof(double) * 6 * (size_t)A));
    OMR_HIP(adelta.alloc(sizeof(int32_t) * (size_t)A * cols));
    OMR_HIP(bdelta.alloc(sizeof(int32_t) * (size_t)A * cols));
    OMR_HIP(xy0.alloc(sizeof(int2_t) * (size_t)A * rows));
    OMR_HIP(tiles.alloc(sizeof(LdsTile) * (size_t)A));
    DevBuf ovf;
    OMR_HIP(ovf.alloc(sizeof(int32_t)));
    OMR_HIP(hipMemcpy(minv.p, host_minv.data(), sizeof(double) * 6 * (size_t)A, hipMemcpyHostToDevice));
    OMR_HIP(hipMemcpy(tiles.p, ht.data(), sizeof(LdsTile) * (size_t)A, hipMemcpyHostToDevice));
    OMR_HIP(hipMemset(ovf.p, 0, sizeof(int32_t)));
    OMR_HIP(launch_tables(minv.as<double>(), dims, 512, adelta.as<int32_t>(), bdelta.as<int32_t>(), xy0.as<int2_t>(),
                          ovf.as<int32_t>(), nullptr));
    int32_t h_ovf = 0;
    OMR_HIP(hipMemcpy(&h_ovf, ovf.p, sizeof h_ovf, hipMemcpyDeviceToHost));
    if (h_ovf) return fail(OMR_ERR_BADARG, "affine map leaves the 32-bit fixed-point range of warpAffine");
    return build_runs();
}

// Run tables, then a dry run of the kernel on an all-white scan: window geometry does not depend
// on the pixels, so a candidate whose windows fit once always fits.
int SweepTables::build_runs()
{
    const int rows = dims.rows, cols = dims.cols, A = dims.A;
    runs_built = false;
    n_runs = 0;
    n_gather = A;
    host_mode.assign((size_t)A, 0);
    NWh = (cols + 31) / 32;
    Gh = (NWh + OMR_RUN_K - 1) / OMR_RUN_K;
    // chunks of word groups per workgroup: as few row-count partials as possible while the grid still has
    // several workgroups per CU slot (two chunks for an A4 sweep of 400 candidates)
    Ph = (Gh + OMR_RUN_GC - 1) / OMR_RUN_GC;
    if (Gh >= 8 && Ph < 2) Ph = 2;
    GCh = (Gh + Ph - 1) / Ph;
    Ph = (Gh + GCh - 1) / GCh;
    NRp = (rows + 7) & ~7;
    {
        const int NB = (rows + 511) / 512, RBmax = OMR_RUN_MAX_ROWS / 512;
        RCHh = (NB + RBmax - 1) / RBmax;
        RBh = (NB + RCHh - 1) / RCHh;
    }
    const int NWp = Gh * OMR_RUN_K;
    const size_t tab_bytes = (size_t)A * (size_t)NWp * sizeof(RunTab);
    if (tab_bytes > ((size_t)3 << 30)) return OMR_OK;  // gather kernels only
    OMR_HIP(tabsH.alloc(sizeof(RunTab) * (size_t)A * NWp));
    OMR_HIP(metaH.alloc(sizeof(RunMeta) * (size_t)A * NWh));
    OMR_HIP(metacH.alloc(sizeof(int2_t) * (size_t)A * NWp));
    OMR_HIP(blkH.alloc(sizeof(RunBlk) * (size_t)A * Gh));
    OMR_HIP(mode.alloc(sizeof(int32_t) * (size_t)A));
    OMR_HIP(list_runs.alloc(sizeof(int32_t) * (size_t)A));
    OMR_HIP(list_gather.alloc(sizeof(int32_t) * (size_t)A));
    OMR_HIP(hipMemset(tabsH.p, 0, tabsH.bytes));
    OMR_HIP(launch_runtab(adelta.as<int32_t>(), bdelta.as<int32_t>(), A, cols, NWh, tabsH.as<RunTab>(),
                          metaH.as<RunMeta>(), metacH.as<int2_t>(), blkH.as<RunBlk>(), nullptr));
    // window origins of every (candidate, word group, band, wave), and how far they reach beyond the image: that
    // sizes the zero guard of the transposed bit image (the sweep then fetches its windows without a range test)
    {
        const int NBt = (rows + 511) / 512;
        OMR_HIP(wgeoH.alloc(sizeof(int2_t) * (size_t)A * Gh * NBt * 8));
        DevBuf ext;
        OMR_HIP(ext.alloc(4 * sizeof(int32_t)));
        int32_t h_ext[4] = {INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN};
        OMR_HIP(hipMemcpy(ext.p, h_ext, sizeof h_ext, hipMemcpyHostToDevice));
        OMR_HIP(launch_rungeo(xy0.as<int2_t>(), blkH.as<RunBlk>(), A, Gh, rows, wgeoH.as<int2_t>(), ext.as<int32_t>(), nullptr));
        OMR_HIP(hipMemcpy(h_ext, ext.p, sizeof h_ext, hipMemcpyDeviceToHost));
        if (h_ext[0] > h_ext[1]) return OMR_OK;  // no window fits: gather kernels only
        GXh = std::max(0, std::max(-h_ext[0], h_ext[1] - NWh));
        GYh = (std::max(0, std::max(-h_ext[2], h_ext[3] - rows)) + 3) & ~3;
        NWt = NWh + 2 * GXh;
        rowsT = ((rows + 3) & ~3) + 2 * GYh;
        if ((size_t)NWt * rowsT * 4 > ((size_t)1 << 30)) return OMR_OK;  // absurd guard: gather kernels only
    }
    // dry run on an all-white scan: window geometry does not depend on the pixels, so a candidate whose
    // windows fit once always fits
    DevBuf z0, hp, vp, gd, all;
    OMR_HIP(z0.alloc(sizeof(uint32_t) * (size_t)NWt * rowsT));
    OMR_HIP(hp.alloc(sizeof(uint16_t) * (size_t)A * Ph * NRp));
    OMR_HIP(vp.alloc(sizeof(uint32_t) * (size_t)A * cols));
    OMR_HIP(gd.alloc(sizeof(int32_t) * (size_t)A));
    OMR_HIP(all.alloc(sizeof(int32_t) * (size_t)A));
    OMR_HIP(hipMemset(z0.p, 0, z0.bytes));
    OMR_HIP(hipMemset(gd.p, 0, gd.bytes));
    OMR_HIP(hipMemset(vp.p, 0, vp.bytes));
    std::vector<int32_t> idx((size_t)A);
    for (int a = 0; a < A; a++) idx[a] = a;
    OMR_HIP(hipMemcpy(all.p, idx.data(), sizeof(int32_t) * (size_t)A, hipMemcpyHostToDevice));
    const RunPass ph = run_pass(z0.as<uint32_t>(), hp.as<uint16_t>(), 1);
    OMR_HIP(launch_runs(ph, all.as<int32_t>(), A, gd.as<int32_t>(), vp.as<uint32_t>(), nullptr));
    std::vector<int32_t> g((size_t)A);
    std::vector<RunMeta> mh((size_t)A * NWh);
    OMR_HIP(hipMemcpy(g.data(), gd.p, sizeof(int32_t) * (size_t)A, hipMemcpyDeviceToHost));
    OMR_HIP(hipMemcpy(mh.data(), metaH.p, sizeof(RunMeta) * mh.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> lr, lg;
    for (int a = 0; a < A; a++) {
        bool ok = g[a] == 0;
        for (int w = 0; ok && w < NWh; w++) ok = mh[(size_t)a * NWh + w].ok != 0;
        host_mode[a] = ok ? 1 : 0;
        (ok ? lr : lg).push_back(a);
    }
    n_runs = (int)lr.size();
    n_gather = (int)lg.size();
    OMR_HIP(hipMemcpy(mode.p, host_mode.data(), sizeof(int32_t) * (size_t)A, hipMemcpyHostToDevice));
    if (n_runs) OMR_HIP(hipMemcpy(list_runs.p, lr.data(), sizeof(int32_t) * lr.size(), hipMemcpyHostToDevice));
    if (n_gather) OMR_HIP(hipMemcpy(list_gather.p, lg.data(), sizeof(int32_t) * lg.size(), hipMemcpyHostToDevice));
    runs_built = true;
    return OMR_OK;
}

RunPass SweepTables::run_pass(const uint32_t *d_bitsT, uint16_t *d_part, int scans) const
{
    RunPass p{};
    p.srcT = d_bitsT;
    p.NWt = NWt;
    p.rowsT = rowsT;
    p.GX = GXh;
    p.GY = GYh;
    p.wgeo = wgeoH.as<int2_t>();
    p.RT = xy0.as<int2_t>();
    p.NR = dims.rows;
    p.NC = dims.cols;
    p.NWp = Gh * OMR_RUN_K;
    p.tabs = tabsH.as<RunTab>();
    p.metac = metacH.as<int2_t>();
    p.blk = blkH.as<RunBlk>();
    p.part = d_part;
    p.G = Gh;
    p.GC = GCh;
    p.P = Ph;
    p.NRp = NRp;
    p.RB = RBh;
    p.RCH = RCHh;
    p.scans = scans;
    p.A = dims.A;
    return p;
}

int SweepScratch::create(const SweepTables &t, int scans_per_launch)
{
    const SweepDims &d = t.dims;
    const size_t Z = (size_t)(scans_per_launch > 0 ? scans_per_launch : 1);
    zmax = (int)Z;
    if (t.runs_built && t.n_runs > 0) {
        OMR_HIP(hpart.alloc(sizeof(uint16_t) * Z * (size_t)d.A * t.Ph * t.NRp));
        OMR_HIP(guard.alloc(sizeof(int32_t) * (size_t)d.A));
        OMR_HIP(hipMemset(guard.p, 0, guard.bytes));  // runs_kernel sets guard[a] when it could not sweep candidate a
        OMR_HIP(bitsT.alloc(sizeof(uint32_t) * Z * (size_t)t.NWt * t.rowsT));
        OMR_HIP(hipMemset(bitsT.p, 0, bitsT.bytes));  // the guard stays zero: the transpose writes the image only
    }
    OMR_HIP(bits.alloc(sizeof(uint32_t) * Z * (size_t)d.rows * d.wpr));
    OMR_HIP(vproj.alloc(sizeof(uint32_t) * Z * (size_t)d.A * d.cols));
    OMR_HIP(hproj.alloc(sizeof(uint32_t) * Z * (size_t)d.A * d.rows));
    OMR_HIP(vsd.alloc(sizeof(double) * Z * (size_t)d.A));
    OMR_HIP(hsd.alloc(sizeof(double) * Z * (size_t)d.A));
    OMR_HIP(best.alloc(sizeof(int32_t) * Z));
    return OMR_OK;
}

int enqueue_sweep(const SweepTables &t, SweepScratch &s, int kernel_sel, const uint8_t *d_img, int64_t step,
                  int black_max, hipStream_t stream, uint32_t *d_vproj, uint32_t *d_hproj, double *d_v_sd,
                  double *d_h_sd, int32_t *d_best, hipEvent_t ev0, hipEvent_t ev1, bool want_proj,
                  hipStream_t post_stream, hipEvent_t ev_mid, int scans, int64_t img_stride)
{
    const SweepDims &d = t.dims;
    if (!d_img) return fail(OMR_ERR_BADARG, "null image");
    if (scans < 1 || scans > s.zmax) return fail(OMR_ERR_BADARG, "%d scans per launch, scratch holds %d", scans, s.zmax);
    if (scans > 1 && (d_vproj || d_hproj)) return fail(OMR_ERR_BADARG, "projections are returned for single-scan launches");
    if (step < d.cols) return fail(OMR_ERR_BADARG, "step_bytes %lld < cols %d", (long long)step, d.cols);
    // which kernels sweep which candidates
    bool use_runs = false, gather_lds = t.lds_ok;
    const int32_t *glist = nullptr;  // nullptr = every candidate
    int n_g = d.A;
    if (kernel_sel == KERNEL_GENERIC) gather_lds = false;
    else if (kernel_sel == KERNEL_LDS) {
        if (!t.lds_ok) return fail(OMR_ERR_BADARG, "a candidate's source window does not fit the LDS slab");
    } else {
        use_runs = t.runs_built && t.n_runs > 0 && s.hpart.p != nullptr;
        if (kernel_sel == KERNEL_RUNS && !use_runs)
            return fail(OMR_ERR_BADARG, "no candidate of this plan qualifies for the run-merging kernel");
        if (use_runs) {
            glist = t.list_gather.as<int32_t>();
            n_g = t.n_gather;
        }
    }
    want_proj = want_proj || d_vproj || d_hproj;

    // integer projections accumulate with atomics: the caller's buffers double as accumulators
    uint32_t *vp = d_vproj ? d_vproj : s.vproj.as<uint32_t>();
    uint32_t *hp = d_hproj ? d_hproj : s.hproj.as<uint32_t>();
    double *vs = d_v_sd ? d_v_sd : s.vsd.as<double>();
    double *hs = d_h_sd ? d_h_sd : s.hsd.as<double>();
    // The gather kernels accumulate with integer atomics -> vproj / hproj start at 0 when any
    // candidate is gathered; the run-merging kernel overwrites its candidates' rows.
    // (the run-merging kernel adds its column counts with atomics too when it sweeps the image in row chunks)
    if (n_g > 0 || (use_runs && t.RCHh > 1))
        OMR_HIP(hipMemsetAsync(vp, 0, sizeof(uint32_t) * (size_t)scans * d.A * d.cols, stream));
    if (n_g > 0) OMR_HIP(hipMemsetAsync(hp, 0, sizeof(uint32_t) * (size_t)scans * d.A * d.rows, stream));
    OMR_HIP(launch_pack_bits(d_img, step, d.rows, d.cols, black_max, s.bits.as<uint32_t>(), d.wpr, stream, scans,
                             img_stride));
    if (ev0) OMR_HIP(hipEventRecord(ev0, stream));
    if (use_runs) {
        // the run-merging kernel reads word columns: the bit images once more, transposed
        OMR_HIP(launch_transpose_bits(s.bits.as<uint32_t>(), d.rows, d.wpr, s.bitsT.as<uint32_t>(), t.NWh, t.NWt, t.rowsT,
                                      t.GXh, t.GYh, stream, scans));
        const RunPass ph = t.run_pass(s.bitsT.as<uint32_t>(), s.hpart.as<uint16_t>(), scans);
        OMR_HIP(launch_runs(ph, t.list_runs.as<int32_t>(), t.n_runs, s.guard.as<int32_t>(), vp, stream));
        s.guard_pending = true;
    }
    if (n_g > 0) {  // the gathered candidates of ALL scans of the launch group in one launch
        if (gather_lds)
            OMR_HIP(launch_sweep_lds(s.bits.as<uint32_t>(), d, t.adelta.as<int32_t>(), t.bdelta.as<int32_t>(), t.xy0.as<int2_t>(),
                                     t.tiles.as<LdsTile>(), glist, n_g, vp, hp, stream, scans));
        else
            OMR_HIP(launch_sweep_generic(s.bits.as<uint32_t>(), d, t.adelta.as<int32_t>(), t.bdelta.as<int32_t>(), t.xy0.as<int2_t>(), glist,
                                         n_g, vp, hp, stream, scans));
    }
    if (ev1) OMR_HIP(hipEventRecord(ev1, stream));
    if (post_stream && ev_mid) {
        // the latency-bound tail runs on its own stream so the next scan's sweep can start
        OMR_HIP(hipEventRecord(ev_mid, stream));
        OMR_HIP(hipStreamWaitEvent(post_stream, ev_mid, 0));
        stream = post_stream;
    }
    if (use_runs)  // row counts of the run-merged candidates: u16 partials per word group -> hproj
        OMR_HIP(launch_fold_parts(s.hpart.as<uint16_t>(), t.Ph, d.rows, t.NRp, t.list_runs.as<int32_t>(), t.n_runs, hp,
                                  stream, scans, d.A));
    OMR_HIP(launch_stddev(vp, hp, d, vs, hs, stream, scans, /*latency=*/post_stream == nullptr));
    if (d_best) OMR_HIP(launch_argmax_path1(vs, hs, d.A, d_best, stream, scans));
    return OMR_OK;
}

int guard_verdict(const int32_t *flags, size_t n, const char *kernel)
{
    for (size_t i = 0; i < n; i++)
        if (flags[i]) return fail(OMR_ERR_GPU, "%s reported that it could not sweep (guard flag %zu set): no results were computed", kernel, i);
    return OMR_OK;
}

}  // namespace omr

using namespace omr;

omr_sweep_plan::~omr_sweep_plan()
{
    (void)hipSetDevice(tables.device);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
}

omr_batch_ctx::~omr_batch_ctx()
{
    (void)hipSetDevice(tables.device);
    for (auto &e : events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (auto e : ev_mid) (void)hipEventDestroy(e);
    for (auto e : ev_post) (void)hipEventDestroy(e);
    for (auto s : streams) (void)hipStreamDestroy(s);
    for (auto s : post_streams) (void)hipStreamDestroy(s);
}

extern "C" {

int omr_version(void) { return 100; }

int omr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *omr_last_error(void) { return last_error(); }

int omr_get_rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6])
{
    if (!M) return fail(OMR_ERR_BADARG, "null output");
    rotation_matrix_2d(cx, cy, angle_deg, scale, M);
    return OMR_OK;
}

int omr_candidate_count(uint16_t max_angle, double step, int32_t *N_out)
{
    int N;
    int A = candidate_count(max_angle, step, &N);
    if (N_out) *N_out = N;
    return A;
}

int omr_sweep_matrices(int32_t rows, int32_t cols, uint16_t max_angle, double step, double scale, double *M_out,
                       int32_t cap_A)
{
    int N, A = candidate_count(max_angle, step, &N);
    if (!M_out || cap_A < A) return fail(OMR_ERR_BADARG, "matrix buffer too small (%d < %d)", cap_A, A);
    sweep_matrices(rows, cols, N, step, scale, M_out);
    return OMR_OK;
}

int omr_sweep_plan_create(int32_t rows, int32_t cols, const double *fwd_M, int32_t A, int32_t device,
                          omr_sweep_plan **plan_out)
{
    NoPoolScope plan_owned;  // these buffers outlive the calling entry point
    if (!plan_out) return fail(OMR_ERR_BADARG, "null plan_out");
    *plan_out = nullptr;
    std::unique_ptr<omr_sweep_plan> p(new omr_sweep_plan);
    int rc = p->tables.create(rows, cols, fwd_M, A, device);
    if (rc) return rc;
    rc = p->scratch.create(p->tables);
    if (rc) return rc;
    OMR_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    OMR_HIP(hipEventCreate(&p->ev0));
    OMR_HIP(hipEventCreate(&p->ev1));
    *plan_out = p.release();
    return OMR_OK;
}

int omr_sweep_plan_create_angles(int32_t rows, int32_t cols, uint16_t max_angle, double step, double scale,
                                 int32_t device, omr_sweep_plan **plan_out)
{
    NoPoolScope plan_owned;  // these buffers outlive the calling entry point
    int N, A = candidate_count(max_angle, step, &N);
    if (A <= 0) return fail(OMR_ERR_BADARG, "empty candidate range (max_angle %u, step %g)", (unsigned)max_angle, step);
    std::vector<double> M((size_t)A * 6);
    sweep_matrices(rows, cols, N, step, scale, M.data());
    return omr_sweep_plan_create(rows, cols, M.data(), A, device, plan_out);
}

void omr_sweep_plan_destroy(omr_sweep_plan *plan) { delete plan; }

int omr_sweep_plan_candidates(const omr_sweep_plan *plan) { return plan ? plan->tables.dims.A : 0; }

int omr_sweep_plan_run_device(omr_sweep_plan *plan, const uint8_t *d_img, int64_t step_bytes, int32_t black_max,
                              void *stream, uint32_t *d_vproj, uint32_t *d_hproj, double *d_v_sd, double *d_h_sd,
                              int32_t *d_best_idx)
{
    if (!plan) return fail(OMR_ERR_BADARG, "null plan");
    std::lock_guard<std::mutex> lk(plan->mu);
    OMR_HIP(hipSetDevice(plan->tables.device));
    plan->timed = plan->timing;
    return enqueue_sweep(plan->tables, plan->scratch, plan->kernel_sel, d_img, step_bytes, black_max,
                         (hipStream_t)stream, d_vproj, d_hproj, d_v_sd, d_h_sd, d_best_idx,
                         plan->timing ? plan->ev0 : nullptr, plan->timing ? plan->ev1 : nullptr);
}

int omr_sweep_plan_run(omr_sweep_plan *plan, const omr_image *img, int32_t black_max, uint32_t *vproj,
                       uint32_t *hproj, double *v_sd, double *h_sd, int32_t *best_idx)
{
    if (!plan || !img || !img->data) return fail(OMR_ERR_BADARG, "null plan or image");
    const SweepDims &d = plan->tables.dims;
    if (img->channels != 1) return fail(OMR_ERR_ASSERT, "sweep needs a 1-channel image (got %d)", img->channels);
    if (img->rows != d.rows || img->cols != d.cols)
        return fail(OMR_ERR_ASSERT, "image %dx%d does not match the plan %dx%d", img->cols, img->rows, d.cols, d.rows);
    if (img->step_bytes < img->cols) return fail(OMR_ERR_BADARG, "step_bytes < cols");
    std::lock_guard<std::mutex> lk(plan->mu);
    OMR_HIP(hipSetDevice(plan->tables.device));
    const size_t need = (size_t)d.rows * d.cols;
    if (plan->img.bytes < need) {
        NoPoolScope plan_owned;  // the staging buffer stays with the plan
        OMR_HIP(plan->img.alloc(need));
    }
    hipStream_t s = plan->stream;
    if (img->step_bytes == d.cols)  // packed: one linear copy (the 2-D path is row-by-row DMA for odd widths)
        OMR_HIP(hipMemcpyAsync(plan->img.p, img->data, need, hipMemcpyHostToDevice, s));
    else
        OMR_HIP(hipMemcpy2DAsync(plan->img.p, (size_t)d.cols, img->data, (size_t)img->step_bytes, (size_t)d.cols,
                                 (size_t)d.rows, hipMemcpyHostToDevice, s));
    plan->timed = plan->timing;
    int rc = enqueue_sweep(plan->tables, plan->scratch, plan->kernel_sel, plan->img.as<uint8_t>(), d.cols, black_max, s,
                           nullptr, nullptr, nullptr, nullptr, plan->scratch.best.as<int32_t>(),
                           plan->timing ? plan->ev0 : nullptr, plan->timing ? plan->ev1 : nullptr,
                           vproj != nullptr || hproj != nullptr);
    if (rc) return rc;
    if (vproj)
        OMR_HIP(hipMemcpyAsync(vproj, plan->scratch.vproj.p, sizeof(uint32_t) * (size_t)d.A * d.cols,
                               hipMemcpyDeviceToHost, s));
    if (hproj)
        OMR_HIP(hipMemcpyAsync(hproj, plan->scratch.hproj.p, sizeof(uint32_t) * (size_t)d.A * d.rows,
                               hipMemcpyDeviceToHost, s));
    if (v_sd) OMR_HIP(hipMemcpyAsync(v_sd, plan->scratch.vsd.p, sizeof(double) * (size_t)d.A, hipMemcpyDeviceToHost, s));
    if (h_sd) OMR_HIP(hipMemcpyAsync(h_sd, plan->scratch.hsd.p, sizeof(double) * (size_t)d.A, hipMemcpyDeviceToHost, s));
    if (best_idx) OMR_HIP(hipMemcpyAsync(best_idx, plan->scratch.best.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OMR_HIP(hipStreamSynchronize(s));
    if (plan->scratch.guard_pending) {  // the run-merging kernel's guard flags (see check_guards)
        plan->scratch.guard_pending = false;
        std::vector<int32_t> g(plan->scratch.guard.bytes / sizeof(int32_t));
        OMR_HIP(hipMemcpy(g.data(), plan->scratch.guard.p, plan->scratch.guard.bytes, hipMemcpyDeviceToHost));
        if (int r = guard_verdict(g.data(), g.size(), "runs_kernel")) {
            OMR_HIP(hipMemset(plan->scratch.guard.p, 0, plan->scratch.guard.bytes));
            return r;
        }
    }
    return OMR_OK;
}

int omr_sweep_plan_set_timing(omr_sweep_plan *plan, int32_t enabled)
{
    if (!plan) return fail(OMR_ERR_BADARG, "null plan");
    plan->timing = enabled != 0;
    return OMR_OK;
}

int omr_sweep_plan_last_kernel_ms(omr_sweep_plan *plan, float *ms_out)
{
    if (!plan || !ms_out) return fail(OMR_ERR_BADARG, "null argument");
    if (!plan->timed) return fail(OMR_ERR_BADARG, "timing was not enabled for the last run");
    OMR_HIP(hipSetDevice(plan->tables.device));
    OMR_HIP(hipEventSynchronize(plan->ev1));
    OMR_HIP(hipEventElapsedTime(ms_out, plan->ev0, plan->ev1));
    return OMR_OK;
}

int omr_sweep_plan_set_kernel(omr_sweep_plan *plan, int32_t which)
{
    if (!plan || which < 0 || which > 3) return fail(OMR_ERR_BADARG, "bad kernel selector");
    if (which == KERNEL_RUNS && !(plan->tables.runs_built && plan->tables.n_runs > 0))
        return fail(OMR_ERR_BADARG, "no candidate of this plan qualifies for the run-merging kernel");
    if (which == KERNEL_LDS && !plan->tables.lds_ok)
        return fail(OMR_ERR_BADARG, "a candidate's source window does not fit the LDS slab");
    plan->kernel_sel = which;
    return OMR_OK;
}

int omr_sweep_plan_info(const omr_sweep_plan *plan, int32_t *n_runs, int32_t *n_gather)
{
    if (!plan) return fail(OMR_ERR_BADARG, "null plan");
    const bool runs = plan->tables.runs_built && plan->tables.n_runs > 0;
    if (n_runs) *n_runs = runs ? plan->tables.n_runs : 0;
    if (n_gather) *n_gather = runs ? plan->tables.n_gather : plan->tables.dims.A;
    return OMR_OK;
}

#ifdef OMR_RUNS_DEBUG
// development aid (make debug only, not in the public header): phase clocks of runs_kernel
int omr_debug_runs_stamps(unsigned long long *out8, int reset)
{
    if (!out8) return fail(OMR_ERR_BADARG, "null output");
    OMR_HIP(hipDeviceSynchronize());
    OMR_HIP(debug_runs_stamps(out8, reset != 0));
    return OMR_OK;
}
#endif

int omr_sweep_plan_tables(omr_sweep_plan *plan, int32_t a, int32_t *adelta, int32_t *bdelta, int32_t *X0, int32_t *Y0)
{
    if (!plan) return fail(OMR_ERR_BADARG, "null plan");
    const SweepDims &d = plan->tables.dims;
    if (a < 0 || a >= d.A) return fail(OMR_ERR_BADARG, "candidate %d out of range", a);
    OMR_HIP(hipSetDevice(plan->tables.device));
    if (adelta)
        OMR_HIP(hipMemcpy(adelta, plan->tables.adelta.as<int32_t>() + (size_t)a * d.cols, sizeof(int32_t) * d.cols,
                          hipMemcpyDeviceToHost));
    if (bdelta)
        OMR_HIP(hipMemcpy(bdelta, plan->tables.bdelta.as<int32_t>() + (size_t)a * d.cols, sizeof(int32_t) * d.cols,
                          hipMemcpyDeviceToHost));
    if (X0 || Y0) {
        std::vector<int2_t> t((size_t)d.rows);
        OMR_HIP(hipMemcpy(t.data(), plan->tables.xy0.as<int2_t>() + (size_t)a * d.rows, sizeof(int2_t) * d.rows,
                          hipMemcpyDeviceToHost));
        for (int y = 0; y < d.rows; y++) {
            if (X0) X0[y] = t[y].x;
            if (Y0) Y0[y] = t[y].y;
        }
    }
    return OMR_OK;
}

int omr_projection_sweep(const omr_image *bin, const double *fwd_M, int32_t A, uint32_t *vproj, uint32_t *hproj,
                         double *v_sd, double *h_sd)
{
    if (!bin || !bin->data) return fail(OMR_ERR_BADARG, "null image");
    if (A == 0) return OMR_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(OMR_ERR_GPU, "no usable HIP device (there is no CPU fallback)");
    omr_sweep_plan *plan = nullptr;
    int rc = omr_sweep_plan_create(bin->rows, bin->cols, fwd_M, A, dev, &plan);
    if (rc) return rc;
    rc = omr_sweep_plan_run(plan, bin, 0, vproj, hproj, v_sd, h_sd, nullptr);
    omr_sweep_plan_destroy(plan);
    return rc;
}

int omr_argmax_projection_device(const double *d_v_sd, const double *d_h_sd, int32_t n, int32_t *d_index_out,
                                 void *stream)
{
    if (!d_v_sd || !d_h_sd || !d_index_out || n <= 0) return fail(OMR_ERR_BADARG, "bad arguments");
    OMR_HIP(launch_argmax_path1(d_v_sd, d_h_sd, n, d_index_out, (hipStream_t)stream));
    return OMR_OK;
}

// ---- batch ----------------------------------------------------------------------------------

int omr_batch_create(int32_t rows, int32_t cols, uint16_t max_angle, double step, double scale, int32_t device,
                     int32_t n_streams, omr_batch_ctx **ctx_out)
{
    NoPoolScope plan_owned;  // these buffers outlive the calling entry point
    if (!ctx_out) return fail(OMR_ERR_BADARG, "null ctx_out");
    *ctx_out = nullptr;
    if (n_streams < 1 || n_streams > 16) return fail(OMR_ERR_BADARG, "n_streams must be in 1..16");
    int N, A = candidate_count(max_angle, step, &N);
    if (A <= 0) return fail(OMR_ERR_BADARG, "empty candidate range");
    std::vector<double> M((size_t)A * 6);
    sweep_matrices(rows, cols, N, step, scale, M.data());
    std::unique_ptr<omr_batch_ctx> c(new omr_batch_ctx);
    c->N = N;
    c->step = step;
    int rc = c->tables.create(rows, cols, M.data(), A, device);
    if (rc) return rc;
    for (int i = 0; i < n_streams; i++) {
        // the sweep stream outranks the post stream: std-dev / arg-max fill the gaps, never the reverse
        int prio_lo = 0, prio_hi = 0;
        OMR_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        hipStream_t s, ps;
        OMR_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_hi));
        c->streams.push_back(s);
        OMR_HIP(hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, prio_lo));
        c->post_streams.push_back(ps);
        c->issued.push_back(0);
        for (int h = 0; h < 2; h++) {
            c->scratch.emplace_back(new SweepScratch);
            rc = c->scratch.back()->create(c->tables);
            if (rc) return rc;
            hipEvent_t e0, e1;
            OMR_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
            OMR_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            c->ev_mid.push_back(e0);
            c->ev_post.push_back(e1);
            c->post_pending.push_back(0);
        }
    }
    *ctx_out = c.release();
    return OMR_OK;
}

void omr_batch_destroy(omr_batch_ctx *ctx) { delete ctx; }

int omr_batch_info(const omr_batch_ctx *ctx, int32_t *n_runs, int32_t *n_gather)
{
    if (!ctx) return fail(OMR_ERR_BADARG, "null ctx");
    const bool runs = ctx->tables.runs_built && ctx->tables.n_runs > 0;
    if (n_runs) *n_runs = runs ? ctx->tables.n_runs : 0;
    if (n_gather) *n_gather = runs ? ctx->tables.n_gather : ctx->tables.dims.A;
    return OMR_OK;
}

int omr_batch_set_timing(omr_batch_ctx *ctx, int32_t enabled)
{
    if (!ctx) return fail(OMR_ERR_BADARG, "null ctx");
    ctx->timing = enabled != 0;
    return OMR_OK;
}

namespace {
struct DeskewOut {  // omr_batch_deskew_device's extra stage
    int interp;
    int border;
    uint8_t *d_out;
    int64_t out_stride, out_step;
    int32_t *d_out_size;
};

// per candidate: CONTAIN canvas + warpAffine's tables of its rotation (transfer.rs:487-519), once per context
int build_deskew_tables(omr_batch_ctx *ctx)
{
    if (ctx->dk_built) return OMR_OK;
    NoPoolScope ctx_owned;
    const SweepDims &d = ctx->tables.dims;
    const int A = d.A;
    std::vector<double> minv((size_t)A * 6);
    std::vector<int32_t> size((size_t)A * 2);
    int DR = 0, DC = 0;
    for (int i = 0; i < A; i++) {
        double M[6];
        int dr, dc;
        int rc = rotate_geometry(d.rows, d.cols, (double)(i - ctx->N) * ctx->step, 1.0, OMR_CLIP_CONTAIN, M, &dr, &dc);
        if (rc) return rc;
        invert_affine(M, &minv[6 * (size_t)i]);
        size[2 * (size_t)i] = dr;
        size[2 * (size_t)i + 1] = dc;
        DR = std::max(DR, dr);
        DC = std::max(DC, dc);
    }
    DC = (DC + 3) & ~3;
    DevBuf d_minv, ovf;
    OMR_HIP(d_minv.alloc(sizeof(double) * minv.size()));
    OMR_HIP(ovf.alloc(sizeof(int32_t)));
    OMR_HIP(ctx->dk_size.alloc(sizeof(int32_t) * size.size()));
    OMR_HIP(ctx->dk_adelta.alloc(sizeof(int32_t) * (size_t)A * DC));
    OMR_HIP(ctx->dk_bdelta.alloc(sizeof(int32_t) * (size_t)A * DC));
    OMR_HIP(ctx->dk_xy0.alloc(sizeof(int2_t) * (size_t)A * DR));
    OMR_HIP(hipMemcpy(d_minv.p, minv.data(), sizeof(double) * minv.size(), hipMemcpyHostToDevice));
    OMR_HIP(hipMemcpy(ctx->dk_size.p, size.data(), sizeof(int32_t) * size.size(), hipMemcpyHostToDevice));
    OMR_HIP(hipMemset(ovf.p, 0, sizeof(int32_t)));
    SweepDims td{DR, DC, A, 0};
    OMR_HIP(launch_tables(d_minv.as<double>(), td, 0, ctx->dk_adelta.as<int32_t>(), ctx->dk_bdelta.as<int32_t>(),
                          ctx->dk_xy0.as<int2_t>(), ovf.as<int32_t>(), nullptr));
    int32_t h_ovf = 0;
    OMR_HIP(hipMemcpy(&h_ovf, ovf.p, sizeof h_ovf, hipMemcpyDeviceToHost));  // also: the tables are complete
    if (h_ovf) return fail(OMR_ERR_BADARG, "affine map leaves the 32-bit fixed-point range of warpAffine");
    ctx->dk_rows = DR;
    ctx->dk_cols = DC;
    ctx->dk_built = true;
    return OMR_OK;
}

int batch_run(omr_batch_ctx *ctx, const uint8_t *d_scans, int64_t scan_stride, int64_t step_bytes, int32_t n,
              int32_t black_max, int32_t *d_best_idx, double *d_v_sd, double *d_h_sd, const DeskewOut *dk)
{
    const int S = (int)ctx->streams.size();
    const int A = ctx->tables.dims.A;
    const bool lanes = ctx->lanes > 0;  // scan-lane sweep: 64 scans per wavefront, up to ctx->lanes scans per launch
    for (int i = 0, launch = 0; i < n; launch++) {
        const int z = std::min(lanes ? ctx->lanes : ctx->group, n - i);  // scans of this launch
        const int k = launch % S;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->timing) {
            if (ctx->events_used == ctx->events.size()) {
                hipEvent_t a, b;
                OMR_HIP(hipEventCreate(&a));
                OMR_HIP(hipEventCreate(&b));
                ctx->events.emplace_back(a, b);
            }
            e0 = ctx->events[ctx->events_used].first;
            e1 = ctx->events[ctx->events_used].second;
            ctx->events_used++;
        }
        const int set = 2 * k + (int)(ctx->issued[k] & 1);
        ctx->issued[k]++;
        // this scratch set's previous std-dev / arg-max (/ warp) must have read its projections (/ best indices)
        if (ctx->post_pending[set]) OMR_HIP(hipStreamWaitEvent(ctx->streams[k], ctx->ev_post[set], 0));
        // the warp reads the winners on the device: the caller's array, or the scratch set's
        int32_t *best = d_best_idx ? d_best_idx + i
                                   : (dk ? (lanes ? ctx->slane_scratch[set]->best.as<int32_t>() : ctx->scratch[set]->best.as<int32_t>())
                                         : nullptr);
        int rc;
        if (lanes)
            rc = slane_enqueue(ctx->slane, *ctx->slane_scratch[set], d_scans + (size_t)i * scan_stride, scan_stride, step_bytes, z,
                               black_max, ctx->streams[k], ctx->post_streams[k], ctx->ev_mid[set],
                               d_v_sd ? d_v_sd + (size_t)i * A : nullptr, d_h_sd ? d_h_sd + (size_t)i * A : nullptr, best, e0, e1);
        else
            rc = enqueue_sweep(ctx->tables, *ctx->scratch[set], KERNEL_AUTO, d_scans + (size_t)i * scan_stride,
                               step_bytes, black_max, ctx->streams[k], nullptr, nullptr,
                               d_v_sd ? d_v_sd + (size_t)i * A : nullptr, d_h_sd ? d_h_sd + (size_t)i * A : nullptr, best, e0,
                               e1, false, ctx->post_streams[k], ctx->ev_mid[set], z, scan_stride);
        if (rc) return rc;
        if (dk) {  // after the arg-max, on the post stream: overlaps the next group's sweep
            DeskewPass p{};
            p.src = d_scans + (size_t)i * scan_stride;
            p.scan_stride = scan_stride;
            p.sstep = step_bytes;
            p.srows = ctx->tables.dims.rows;
            p.scols = ctx->tables.dims.cols;
            p.dst = dk->d_out + (size_t)i * dk->out_stride;
            p.out_stride = dk->out_stride;
            p.dstep = dk->out_step;
            p.best = best;
            p.wsize = ctx->dk_size.as<int32_t>();
            p.adelta = ctx->dk_adelta.as<int32_t>();
            p.bdelta = ctx->dk_bdelta.as<int32_t>();
            p.xy0 = ctx->dk_xy0.as<int2_t>();
            p.DC = ctx->dk_cols;
            p.DR = ctx->dk_rows;
            p.border = dk->border;
            p.out_size = dk->d_out_size ? dk->d_out_size + 2 * (size_t)i : nullptr;
            // the per-tile records: one buffer per post stream (the warps of a stream run one after the other)
            if (ctx->dk_tiles.size() < ctx->post_streams.size()) ctx->dk_tiles.resize(ctx->post_streams.size());
            if (!ctx->dk_tiles[(size_t)k]) ctx->dk_tiles[(size_t)k].reset(new DevBuf);
            {
                const size_t need = deskew_tile_bytes(p, lanes ? ctx->lanes : ctx->group);
                if (ctx->dk_tiles[(size_t)k]->bytes < need) {
                    NoPoolScope ctx_owned;
                    OMR_HIP(hipStreamSynchronize(ctx->post_streams[k]));
                    ctx->dk_tiles[(size_t)k]->release();
                    OMR_HIP(ctx->dk_tiles[(size_t)k]->alloc(need));
                }
            }
            OMR_HIP(launch_deskew_warp(p, z, dk->interp, ctx->dk_tiles[(size_t)k]->p, ctx->post_streams[k]));
        }
        if (hipEventRecord(ctx->ev_post[set], ctx->post_streams[k]) != hipSuccess)
            return fail(OMR_ERR_GPU, "hipEventRecord failed");
        ctx->post_pending[set] = 1;
        i += z;
    }
    return OMR_OK;
}
}  // namespace

int omr_batch_run_device(omr_batch_ctx *ctx, const uint8_t *d_scans, int64_t scan_stride, int64_t step_bytes,
                         int32_t n, int32_t black_max, int32_t *d_best_idx, double *d_v_sd, double *d_h_sd)
{
    if (!ctx || !d_scans || n < 0) return fail(OMR_ERR_BADARG, "bad batch arguments");
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    return batch_run(ctx, d_scans, scan_stride, step_bytes, n, black_max, d_best_idx, d_v_sd, d_h_sd, nullptr);
}

}  // extern "C"

// Scans that are already packed to 1 bit per pixel on the device ([rows][ceil(cols / 32)] dwords each, bit i of word c =
// pixel 32 c + i is black): the host-memory batch's packed transfer mode (oics_hostbatch.cpp).  Scan-lane contexts only.
int omr::batch_run_device_bits(omr_batch_ctx *ctx, const uint32_t *d_bits, int64_t scan_stride_bytes, int32_t n,
                               int32_t *d_best_idx, double *d_v_sd, double *d_h_sd)
{
    if (!ctx || !d_bits || n < 0) return fail(OMR_ERR_BADARG, "bad batch arguments");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->lanes <= 0) return fail(OMR_ERR_BADARG, "packed scans need a context in scan-lane mode");
    OMR_HIP(hipSetDevice(ctx->tables.device));
    return batch_run(ctx, (const uint8_t *)d_bits, scan_stride_bytes, 0, n, /*black_max: packed*/ -1, d_best_idx, d_v_sd, d_h_sd, nullptr);
}

extern "C" {

int omr_batch_deskew_canvas(omr_batch_ctx *ctx, int32_t *max_rows, int32_t *max_cols)
{
    if (!ctx) return fail(OMR_ERR_BADARG, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    int rc = build_deskew_tables(ctx);
    if (rc) return rc;
    if (max_rows) *max_rows = ctx->dk_rows;
    if (max_cols) *max_cols = ctx->dk_cols;
    return OMR_OK;
}

int omr_batch_deskew_device(omr_batch_ctx *ctx, const uint8_t *d_scans, int64_t scan_stride, int64_t step_bytes, int32_t n,
                            int32_t black_max, int32_t interp, uint8_t border_value, uint8_t *d_out, int64_t out_stride,
                            int64_t out_step, int32_t *d_out_size, int32_t *d_best_idx)
{
    if (!ctx || !d_scans || !d_out || n < 0) return fail(OMR_ERR_BADARG, "bad batch arguments");
    if (interp != OMR_INTER_NEAREST && interp != OMR_INTER_LINEAR)
        return fail(OMR_ERR_NOTIMPL, "interpolation flag %d is not implemented", interp);
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    int rc = build_deskew_tables(ctx);
    if (rc) return rc;
    if (out_step < ctx->dk_cols || out_stride < (int64_t)ctx->dk_rows * out_step)
        return fail(OMR_ERR_BADARG, "every output slot must hold the largest canvas, %d x %d (omr_batch_deskew_canvas)",
                    ctx->dk_cols, ctx->dk_rows);
    DeskewOut dk{interp, (int)border_value, d_out, out_stride, out_step, d_out_size};
    return batch_run(ctx, d_scans, scan_stride, step_bytes, n, black_max, d_best_idx, nullptr, nullptr, &dk);
}

int omr_call_pool_stats(int32_t device, int32_t *live_slots, int32_t *idle_slots, int64_t *idle_pinned_bytes)
{
    int live = 0, idle = 0;
    size_t pinned = 0;
    call_slot_stats(device, &live, &idle, &pinned);
    if (live_slots) *live_slots = live;
    if (idle_slots) *idle_slots = idle;
    if (idle_pinned_bytes) *idle_pinned_bytes = (int64_t)pinned;
    return OMR_OK;
}

int omr_batch_set_group(omr_batch_ctx *ctx, int32_t scans_per_launch)
{
    NoPoolScope plan_owned;  // these buffers outlive the calling entry point
    if (!ctx || scans_per_launch < 1 || scans_per_launch > 64) return fail(OMR_ERR_BADARG, "scans per launch must be 1..64");
    int rc = omr_batch_sync(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    for (auto &sc : ctx->scratch) {
        std::unique_ptr<SweepScratch> fresh(new SweepScratch);
        sc.reset();  // release before allocating the larger set
        if ((rc = fresh->create(ctx->tables, scans_per_launch))) return rc;
        sc = std::move(fresh);
    }
    ctx->group = scans_per_launch;
    return OMR_OK;
}

// Scan-lane sweep for this context: up to max_scans_per_launch scans per launch (rounded up to whole groups of 64),
// 64 scans per wavefront.  Generates every strip's program ON THE DEVICE (slane_build.hip; once per context); 0 switches back
// to the run-merging path.  OMR_ERR_NOTIMPL (and the context unchanged) when a candidate does not fit the scheme.
int omr_batch_set_lanes(omr_batch_ctx *ctx, int32_t max_scans_per_launch)
{
    NoPoolScope plan_owned;
    if (!ctx || max_scans_per_launch < 0 || max_scans_per_launch > 64 * 64) return fail(OMR_ERR_BADARG, "scans per launch must be 0..4096");
    int rc = omr_batch_sync(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    if (max_scans_per_launch == 0) {
        ctx->lanes = 0;
        ctx->slane_scratch.clear();
        return OMR_OK;
    }
    if (!ctx->slane.built && (rc = ctx->slane.build(ctx->tables))) return rc;
    const int groups = (max_scans_per_launch + SL_LANES - 1) / SL_LANES;
    ctx->slane_scratch.clear();
    for (size_t i = 0; i < ctx->scratch.size(); i++) {
        std::unique_ptr<SlaneScratch> sc(new SlaneScratch);
        if ((rc = sc->create(ctx->slane, groups))) {
            ctx->slane_scratch.clear();
            ctx->lanes = 0;
            return rc;
        }
        ctx->slane_scratch.push_back(std::move(sc));
    }
    ctx->lanes = groups * SL_LANES;
    return OMR_OK;
}

// Inspection switch: launches leave their row counts in the scratch set (omr_batch_lanes_projections reads them)
// instead of clearing them behind the std-dev kernel; the next launch on that set then clears them first.
int omr_batch_lanes_keep(omr_batch_ctx *ctx, int32_t on)
{
    if (!ctx || ctx->lanes <= 0) return fail(OMR_ERR_BADARG, "the context is not in scan-lane mode");
    int rc = omr_batch_sync(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &sc : ctx->slane_scratch) sc->keep_rows = on != 0;
    return OMR_OK;
}

// What the scan-lane plan of a context holds: bytes of programs in HBM, (candidate, strip) tasks, scans per launch.
int omr_batch_lanes_info(omr_batch_ctx *ctx, int64_t *program_bytes, int32_t *tasks, int32_t *scans_per_launch)
{
    if (!ctx) return fail(OMR_ERR_BADARG, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (program_bytes) *program_bytes = ctx->slane.built ? ctx->slane.prog_dwords * 4 : 0;
    if (tasks) *tasks = ctx->slane.built ? (int32_t)ctx->slane.tasks.size() : 0;
    if (scans_per_launch) *scans_per_launch = ctx->lanes;
    return OMR_OK;
}

// The plan's programs were generated on the device (slane_build.hip).  This regenerates them with the host generator
// (slane_plan.cpp, the reference implementation that tests/test_slane_program.py runs through the CPU interpreter) and
// compares the two buffers dword by dword: *dwords = size of the program buffer, *differing = dwords that differ
// (layout differences count as "everything differs").  For tests and inspection; takes seconds at A4.
int omr_batch_lanes_check_programs(omr_batch_ctx *ctx, int64_t *dwords, int64_t *differing)
{
    if (!ctx || ctx->lanes <= 0 || !ctx->slane.built) return fail(OMR_ERR_BADARG, "the context is not in scan-lane mode");
    if (!dwords || !differing) return fail(OMR_ERR_BADARG, "null output");
    int rc = omr_batch_sync(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    SlanePlan ref;
    if ((rc = ref.build(ctx->tables, true))) return rc;
    const SlanePlan &p = ctx->slane;
    *dwords = p.prog_dwords;
    *differing = p.prog_dwords;
    if (ref.prog_dwords != p.prog_dwords || ref.tasks != p.tasks || ref.strips.size() != p.strips.size()) return OMR_OK;
    for (size_t i = 0; i < p.strips.size(); i++)
        if (ref.strips[i].cls != p.strips[i].cls || ref.strips[i].seg_offset != p.strips[i].seg_offset ||
            ref.strips[i].fet_offset != p.strips[i].fet_offset || ref.strips[i].nseg != p.strips[i].nseg)
            return OMR_OK;
    int64_t diff = 0;
    const size_t piece = (size_t)64 << 20;  // dwords per piece
    std::vector<uint32_t> x(std::min<size_t>(piece, (size_t)p.prog_dwords)), y(x.size());
    for (size_t o = 0; o < (size_t)p.prog_dwords; o += piece) {
        const size_t n = std::min(piece, (size_t)p.prog_dwords - o);
        OMR_HIP(hipMemcpy(x.data(), p.prog.as<uint32_t>() + o, 4 * n, hipMemcpyDeviceToHost));
        OMR_HIP(hipMemcpy(y.data(), ref.prog.as<uint32_t>() + o, 4 * n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) diff += x[i] != y[i];
    }
    *differing = diff;
    return OMR_OK;
}

// Integer projections of one scan and candidate as the LAST scan-lane launch of scratch set `set` left them (for
// tests and inspection): vproj cols u32, hproj rows u32.  Synchronises the context.
int omr_batch_lanes_projections(omr_batch_ctx *ctx, int32_t set, int32_t scan, int32_t a, uint32_t *vproj, uint32_t *hproj)
{
    if (!ctx || ctx->lanes <= 0) return fail(OMR_ERR_BADARG, "the context is not in scan-lane mode");
    int rc = omr_batch_sync(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    OMR_HIP(hipSetDevice(ctx->tables.device));
    if (set < 0 || set >= (int)ctx->slane_scratch.size() || scan < 0 || scan >= ctx->lanes || a < 0 || a >= ctx->slane.A)
        return fail(OMR_ERR_BADARG, "set / scan / candidate out of range");
    if (hproj && !ctx->slane_scratch[(size_t)set]->keep_rows)
        return fail(OMR_ERR_BADARG, "row counts are cleared after every launch: call omr_batch_lanes_keep(ctx, 1) first");
    const SlaneScratch &s = *ctx->slane_scratch[(size_t)set];
    const SlanePlan &p = ctx->slane;
    const size_t nscp = (size_t)s.nsg * SL_LANES;
    if (vproj) {  // column counts are u16 on the device
        std::vector<uint16_t> c16((size_t)p.g.cols);
        OMR_HIP(hipMemcpy2D(c16.data(), 2, s.vproj.as<uint16_t>() + (size_t)a * p.g.cols * nscp + scan, nscp * 2, 2, c16.size(),
                            hipMemcpyDeviceToHost));
        for (int c = 0; c < p.g.cols; c++) vproj[c] = c16[(size_t)c];
    }
    if (hproj) {  // rows travel packed in pairs: record 2 i in the low half of a dword, 2 i + 1 in the high half
        std::vector<uint32_t> pairs((size_t)p.nrec / 2);
        OMR_HIP(hipMemcpy2D(pairs.data(), 4, s.hrows.as<uint32_t>() + (size_t)a * (p.nrec / 2) * nscp + scan, nscp * 4, 4,
                            pairs.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < p.g.rows; r++) {
            const int q = r + p.hrow0;
            hproj[r] = (pairs[(size_t)q >> 1] >> ((q & 1) * 16)) & 0xffffu;
        }
    }
    return OMR_OK;  // (the kernel's guard flag was read by the omr_batch_sync above)
}

// The sweep kernels address LDS by integer and hand a candidate back when they cannot sweep it (slane.hip: the
// accumulators are not at LDS address 0; runs.hip: the same, or a window that left its slab): they set a flag and return
// WITHOUT results.  Plan creation makes both impossible for the shapes it accepts, so a set flag means a toolchain or
// plan defect -- and the scores of that launch are zeros or stale.  Every synchronisation point of the production path
// reads the flags of the scratch sets that were launched on since the last one: OMR_ERR_GPU, never silent wrong angles.
static int check_guards(omr_batch_ctx *ctx)
{
    int rc = OMR_OK;
    for (auto &sc : ctx->slane_scratch) {
        if (!sc || !sc->guard_pending) continue;
        sc->guard_pending = false;
        int32_t g = 0;
        OMR_HIP(hipMemcpy(&g, sc->guard.p, sizeof g, hipMemcpyDeviceToHost));
        if (g) {
            OMR_HIP(hipMemset(sc->guard.p, 0, sizeof g));
            rc = guard_verdict(&g, 1, "slane_kernel (its LDS accumulators are not at LDS address 0)");
        }
    }
    for (auto &sc : ctx->scratch) {
        if (!sc || !sc->guard_pending || !sc->guard.p) continue;
        sc->guard_pending = false;
        std::vector<int32_t> g(sc->guard.bytes / sizeof(int32_t));
        OMR_HIP(hipMemcpy(g.data(), sc->guard.p, sc->guard.bytes, hipMemcpyDeviceToHost));
        if (int r = guard_verdict(g.data(), g.size(), "runs_kernel")) {
            OMR_HIP(hipMemset(sc->guard.p, 0, sc->guard.bytes));
            rc = r;
        }
    }
    return rc;
}

int omr_batch_sync(omr_batch_ctx *ctx)
{
    if (!ctx) return fail(OMR_ERR_BADARG, "null ctx");
    OMR_HIP(hipSetDevice(ctx->tables.device));
    for (auto s : ctx->streams) OMR_HIP(hipStreamSynchronize(s));
    for (auto s : ctx->post_streams) OMR_HIP(hipStreamSynchronize(s));
    std::lock_guard<std::mutex> lk(ctx->guard_mu);
    return check_guards(ctx);
}

#ifdef OMR_RUNS_DEBUG
// development aid (make debug only, not in the public header): set a guard flag of scratch set `set` on the device, as a
// kernel that could not sweep would (tests/test_gpu_debuglib.py: the next omr_batch_sync must return OMR_ERR_GPU)
int omr_debug_poke_guard(omr_batch_ctx *ctx, int set)
{
    if (!ctx) return fail(OMR_ERR_BADARG, "null ctx");
    OMR_HIP(hipSetDevice(ctx->tables.device));
    const int32_t one = 1;
    if (ctx->lanes > 0) {
        if (set < 0 || set >= (int)ctx->slane_scratch.size()) return fail(OMR_ERR_BADARG, "no such scratch set");
        OMR_HIP(hipMemcpy(ctx->slane_scratch[(size_t)set]->guard.p, &one, sizeof one, hipMemcpyHostToDevice));
        ctx->slane_scratch[(size_t)set]->guard_pending = true;
    } else {
        if (set < 0 || set >= (int)ctx->scratch.size() || !ctx->scratch[(size_t)set]->guard.p) return fail(OMR_ERR_BADARG, "no such scratch set");
        OMR_HIP(hipMemcpy(ctx->scratch[(size_t)set]->guard.p, &one, sizeof one, hipMemcpyHostToDevice));
        ctx->scratch[(size_t)set]->guard_pending = true;
    }
    return OMR_OK;
}
#endif

int omr_batch_kernel_ms(omr_batch_ctx *ctx, double *sum_ms, int32_t *launches)
{
    if (!ctx || !sum_ms || !launches) return fail(OMR_ERR_BADARG, "null argument");
    int rc = omr_batch_sync(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    double sum = 0;
    for (size_t i = 0; i < ctx->events_used; i++) {
        float ms = 0;
        OMR_HIP(hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
        sum += ms;
    }
    *sum_ms = sum;
    *launches = (int32_t)ctx->events_used;
    ctx->events_used = 0;
    return OMR_OK;
}

}  // extern "C"
