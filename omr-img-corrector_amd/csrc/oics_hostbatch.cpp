// oics_hostbatch.cpp -- batches that start in HOST memory (SURVEY.md 8b `omr_sweep_batch`, 8d's end-to-end figure):
// a context that owns, per device, the sweep plan, a pinned ring and two device stages, so that the copy of one launch's
// scans overlaps the sweep of the previous one and nothing is allocated inside a run.
//
//   caller's scans (pageable: copier threads -> pinned ring slot of 16 scans; or already pinned: straight from the
//   caller's memory; or PACKED: the copier threads turn the binarised scan into 1 bit per pixel on their way into the
//   ring -- they touch every byte anyway -- and 1/8 of the bytes cross the link, 1.09 MB per A4 scan)
//   --async DMA on a copy stream--> device stage (one launch = 64 scans, two stages)
//   --omr_batch_run_device (scan-lane sweep when the candidates fit it, else the run-merging path)--> results, one
//   download at the end.  Scan i goes to device i % n_devices; the only "collective" is the host-side gather.
//   The copier threads are a pool that lives as long as the context (round 4 started 16 threads per 16-scan chunk).
#include <string.h>

#include <emmintrin.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/omrdeskew.h"
#include "engine.hpp"

using namespace omr;

namespace {
#ifdef OMR_RUNS_DEBUG
// development aid (make debug only): k LOGICAL devices mapped onto the visible ones (logical device d = physical device
// d % omr_device_count()), so that the multi-device worker loop of omr_host_batch_run -- interleaved scans, one worker
// thread, context, copy stream and pinned ring per device, results gathered in order -- can be exercised on a box with one
// GPU (tests/test_gpu_debuglib.py).  The release library has no such switch: there a device is a device.
std::atomic<int> g_logical_devices{0};
int visible_devices()
{
    const int k = g_logical_devices.load();
    return k > 0 ? k : omr_device_count();
}
int physical_device(int logical)
{
    const int n = omr_device_count();
    return n > 0 ? logical % n : logical;
}
#else
int visible_devices() { return omr_device_count(); }
int physical_device(int logical) { return logical; }
#endif
constexpr int HB_CHUNK = 16;    // scans per ring slot / DMA (139 MB at A4: DMAs of that size run at the link's rate; three slots are
                                // 0.4 GB of pinned memory to allocate, a quarter of what 64-scan slots cost a one-off call)
constexpr int HB_LANES_MIN = 64; // scans per device from which the scan-lane sweep is used (one full wavefront of scans)
constexpr int HB_LAUNCH = 64;   // scans per sweep launch in scan-lane mode: the pipeline is bound by the copies (10 ms per 64 A4
                                // scans at 56 GB/s against 6.6 ms of sweep), so what counts is the LAST launch's sweep, which
                                // nothing overlaps -- the smaller the launch, the shorter that tail
constexpr int HB_SLOTS = 3;

// ---- the copier threads: a pool owned by the context; run() hands out [0, n) in pieces and returns when all are done
class WorkerPool {
  public:
    explicit WorkerPool(int threads)
    {
        for (int t = 0; t < threads; t++) th_.emplace_back([this]() { loop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    // fn(i) for every i in [0, n), spread over the pool's threads; several callers (one per device) may run() at once
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        Job job{&fn, n, 0, 0};
        std::unique_lock<std::mutex> lk(mu_);
        jobs_.push_back(&job);
        cv_.notify_all();
        done_.wait(lk, [&]() { return job.finished == job.n; });
    }

  private:
    struct Job {
        const std::function<void(int)> *fn;
        int n, next, finished;
    };
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&]() { return stop_ || !jobs_.empty(); });
            if (stop_) return;
            Job *j = jobs_.front();
            const int i = j->next++;
            if (j->next == j->n) jobs_.erase(jobs_.begin());  // every index is handed out: later workers take the next job
            lk.unlock();
            (*j->fn)(i);
            lk.lock();
            if (++j->finished == j->n) done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<Job *> jobs_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

// ---- binarised u8 row -> 1 bit per pixel: bit i of word c = (pixel 32 c + i == 0), the bits past the last column zero
__attribute__((target("avx2"))) void pack_row_avx2(const uint8_t *p, int cols, uint32_t *out, int nw)
{
    const __m256i zero = _mm256_setzero_si256();
    int w = 0;
    for (; (w + 1) * 32 <= cols; w++)
        out[w] = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + 32 * w)), zero));
    for (; w < nw; w++) {
        uint32_t m = 0;
        for (int i = 0; i < 32 && 32 * w + i < cols; i++) m |= (uint32_t)(p[32 * w + i] == 0) << i;
        out[w] = m;
    }
}
void pack_row_sse2(const uint8_t *p, int cols, uint32_t *out, int nw)
{
    const __m128i zero = _mm_setzero_si128();
    int w = 0;
    for (; (w + 1) * 32 <= cols; w++) {
        const uint32_t lo = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + 32 * w)), zero));
        const uint32_t hi = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + 32 * w + 16)), zero));
        out[w] = lo | (hi << 16);
    }
    for (; w < nw; w++) {
        uint32_t m = 0;
        for (int i = 0; i < 32 && 32 * w + i < cols; i++) m |= (uint32_t)(p[32 * w + i] == 0) << i;
        out[w] = m;
    }
}
void pack_scan(const omr_image &im, int rows, int cols, uint32_t *out)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    const int nw = (cols + 31) / 32;
    for (int r = 0; r < rows; r++) {
        const uint8_t *p = im.data + (size_t)r * im.step_bytes;
        if (avx2) pack_row_avx2(p, cols, out + (size_t)r * nw, nw);
        else pack_row_sse2(p, cols, out + (size_t)r * nw, nw);
    }
}

struct HostBatchDev {
    int dev = 0, launch = 0, cap = 0;
    bool lanes = false;
    std::unique_ptr<omr_batch_ctx> ctx;
    uint8_t *ring[HB_SLOTS] = {nullptr, nullptr, nullptr};
    DevBuf stage[2], dbest, dvs, dhs;
    hipStream_t copy = nullptr;
    hipEvent_t ev_slot[HB_SLOTS] = {nullptr, nullptr, nullptr}, ev_ready[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    bool slot_busy[HB_SLOTS] = {false, false, false}, stage_used[2] = {false, false};
    ~HostBatchDev()
    {
        (void)hipSetDevice(dev);
        for (auto &r : ring)
            if (r) (void)hipHostFree(r);
        for (auto e : ev_slot)
            if (e) (void)hipEventDestroy(e);
        for (int k = 0; k < 2; k++) {
            if (ev_ready[k]) (void)hipEventDestroy(ev_ready[k]);
            if (ev_used[k]) (void)hipEventDestroy(ev_used[k]);
        }
        if (copy) (void)hipStreamDestroy(copy);
    }
};
}  // namespace

struct omr_host_batch {
    int rows = 0, cols = 0, N = 0, A = 0, n_devices = 0;
    double step = 0;
    std::vector<std::unique_ptr<HostBatchDev>> devs;
    std::unique_ptr<WorkerPool> pool;  // the copier / packer threads
    std::mutex mu;
};

extern "C" {

int omr_host_batch_create(int32_t rows, int32_t cols, uint16_t max_angle, double step, int32_t n_devices,
                          int32_t max_scans, omr_host_batch **out)
{
    if (!out || max_scans < 1) return fail(OMR_ERR_BADARG, "bad host-batch arguments");
    *out = nullptr;
    if (omr_device_count() <= 0) return fail(OMR_ERR_GPU, "no usable HIP device (there is no CPU fallback)");
    int ndev = visible_devices();
    // 0 (or less) = every visible device; more than are visible is an error, not a silent clamp: a caller that asked
    // for 8 devices must not believe it ran on 8
    if (n_devices > ndev) return fail(OMR_ERR_BADARG, "n_devices exceeds omr_device_count()");
    if (n_devices <= 0) n_devices = ndev;
    std::unique_ptr<omr_host_batch> hb(new omr_host_batch);
    hb->A = candidate_count(max_angle, step, &hb->N);
    if (hb->A <= 0) return fail(OMR_ERR_BADARG, "empty candidate range");
    hb->rows = rows, hb->cols = cols, hb->step = step, hb->n_devices = n_devices;
    {
        const unsigned hw = std::thread::hardware_concurrency();
        hb->pool.reset(new WorkerPool((int)std::max(2u, std::min(hw ? hw : 8u, 32u))));
    }
    const int per_dev = (max_scans + n_devices - 1) / n_devices;
    const size_t img = (size_t)rows * cols;
    NoPoolScope owned;
    for (int dv = 0; dv < n_devices; dv++) {
        std::unique_ptr<HostBatchDev> d(new HostBatchDev);
        d->dev = physical_device(dv);
        OMR_HIP(hipSetDevice(d->dev));
        omr_batch_ctx *raw = nullptr;
        int rc = omr_batch_create(rows, cols, max_angle, step, 1.0, d->dev, 1, &raw);
        if (rc) return rc;
        d->ctx.reset(raw);
        // the scan-lane sweep (its plan is built on the device in tens of milliseconds) from one full wavefront of scans on
        if (per_dev >= HB_LANES_MIN) {
            d->launch = HB_LAUNCH;
            rc = omr_batch_set_lanes(d->ctx.get(), d->launch);
            if (rc == OMR_OK) d->lanes = true;
            else if (rc != OMR_ERR_NOTIMPL) return rc;
        }
        if (!d->lanes) {
            d->launch = std::min(32, per_dev);
            if ((rc = omr_batch_set_group(d->ctx.get(), d->launch))) return rc;
        }
        d->cap = per_dev;
        OMR_HIP(hipStreamCreateWithFlags(&d->copy, hipStreamNonBlocking));
        const int chunk = std::min(HB_CHUNK, d->launch);
        for (int k = 0; k < HB_SLOTS; k++) {
            OMR_HIP(hipHostMalloc((void **)&d->ring[k], img * chunk, hipHostMallocDefault));
            OMR_HIP(hipEventCreateWithFlags(&d->ev_slot[k], hipEventDisableTiming));
        }
        for (int k = 0; k < 2; k++) {
            OMR_HIP(d->stage[k].alloc(img * d->launch));
            OMR_HIP(hipEventCreateWithFlags(&d->ev_ready[k], hipEventDisableTiming));
            OMR_HIP(hipEventCreateWithFlags(&d->ev_used[k], hipEventDisableTiming));
        }
        OMR_HIP(d->dbest.alloc(sizeof(int32_t) * (size_t)per_dev));
        OMR_HIP(d->dvs.alloc(sizeof(double) * (size_t)per_dev * hb->A));
        OMR_HIP(d->dhs.alloc(sizeof(double) * (size_t)per_dev * hb->A));
        hb->devs.push_back(std::move(d));
    }
    *out = hb.release();
    return OMR_OK;
}

void omr_host_batch_destroy(omr_host_batch *hb) { delete hb; }

int omr_host_batch_info(const omr_host_batch *hb, int32_t *n_devices, int32_t *scans_per_launch, int32_t *scan_lane)
{
    if (!hb || hb->devs.empty()) return fail(OMR_ERR_BADARG, "null host batch");
    if (n_devices) *n_devices = hb->n_devices;
    if (scans_per_launch) *scans_per_launch = hb->devs[0]->launch;
    if (scan_lane) *scan_lane = hb->devs[0]->lanes ? 1 : 0;
    return OMR_OK;
}

// Scans per sweep launch of the context's devices (scan-lane mode: a multiple of 64, at most 512).  The default, 64, suits the
// u8 transfer modes, whose pipeline is bound by the copies: what is left exposed is the LAST launch's sweep.  With packed
// transfers the sweep is the bottleneck and larger launches sweep faster per scan (64: 5.9 ms, 128: 10.4, 256: 19 at A4).
int omr_host_batch_set_launch(omr_host_batch *hb, int32_t scans_per_launch)
{
    if (!hb || hb->devs.empty()) return fail(OMR_ERR_BADARG, "null host batch");
    std::lock_guard<std::mutex> lk(hb->mu);
    if (scans_per_launch < 1 || scans_per_launch > 512) return fail(OMR_ERR_BADARG, "scans per launch must be 1..512");
    NoPoolScope owned;
    const size_t img = (size_t)hb->rows * hb->cols;
    for (auto &dp : hb->devs) {
        HostBatchDev &d = *dp;
        OMR_HIP(hipSetDevice(d.dev));
        int launch = scans_per_launch;
        if (d.lanes) launch = ((launch + 63) / 64) * 64;
        else launch = std::min(launch, 64);  // run-merging path: omr_batch_set_group takes 1..64
        if (launch == d.launch) continue;
        int rc = omr_batch_sync(d.ctx.get());
        if (rc) return rc;
        if ((rc = d.lanes ? omr_batch_set_lanes(d.ctx.get(), launch) : omr_batch_set_group(d.ctx.get(), launch))) return rc;
        for (int k = 0; k < 2; k++) {
            d.stage[k].release();
            OMR_HIP(d.stage[k].alloc(img * (size_t)launch));
            d.stage_used[k] = false;
        }
        if (std::min(HB_CHUNK, launch) > std::min(HB_CHUNK, d.launch)) {  // the ring's slots hold min(16, launch) u8 scans
            OMR_HIP(hipStreamSynchronize(d.copy));
            for (int k = 0; k < HB_SLOTS; k++) {
                if (d.ring[k]) OMR_HIP(hipHostFree(d.ring[k]));
                d.ring[k] = nullptr;
                OMR_HIP(hipHostMalloc((void **)&d.ring[k], img * (size_t)std::min(HB_CHUNK, launch), hipHostMallocDefault));
                d.slot_busy[k] = false;
            }
        }
        d.launch = launch;
    }
    return OMR_OK;
}

int omr_host_batch_run(omr_host_batch *hb, const omr_image *scans, int32_t n, int32_t transfer_mode, int32_t *best_idx,
                       double *best_angle, double *v_sd_opt, double *h_sd_opt)
{
    if (!hb || !scans || n < 0 || !best_idx) return fail(OMR_ERR_BADARG, "bad batch arguments");
    if (transfer_mode < OMR_HOST_PAGEABLE || transfer_mode > OMR_HOST_PACKED) return fail(OMR_ERR_BADARG, "transfer mode %d", transfer_mode);
    if (n == 0) return OMR_OK;
    std::lock_guard<std::mutex> lk(hb->mu);
    const int rows = hb->rows, cols = hb->cols, A = hb->A, ND = hb->n_devices;
    for (int i = 0; i < n; i++) {
        const omr_image &im = scans[i];
        if (!im.data || im.channels != 1 || im.rows != rows || im.cols != cols || im.step_bytes < cols)
            return fail(OMR_ERR_ASSERT, "scan %d: every scan of the batch must be a %dx%d 1-channel image", i, cols, rows);
    }
    if ((n + ND - 1) / ND > hb->devs[0]->cap) return fail(OMR_ERR_BADARG, "%d scans: the context was made for %d per device", n, hb->devs[0]->cap);
    std::vector<int> rcs((size_t)ND, OMR_OK);
    std::vector<std::string> errs((size_t)ND);
    const size_t img = (size_t)rows * cols;
    const int NW = (cols + 31) / 32;
    const size_t pimg = (size_t)rows * NW * 4;  // a scan packed to 1 bit per pixel
    WorkerPool &pool = *hb->pool;
    auto worker = [&](int dv) {
        auto run = [&]() -> int {
            HostBatchDev &d = *hb->devs[(size_t)dv];
            OMR_HIP(hipSetDevice(d.dev));
            std::vector<int> mine;
            for (int i = dv; i < n; i += ND) mine.push_back(i);
            const int m = (int)mine.size();
            if (m == 0) return OMR_OK;
            // packed transfers: only the scan-lane sweep reads packed scans; a context on the run-merging path (fewer than
            // 64 scans per device, or candidates the scheme refuses) copies the bytes as they are -- the same results
            const bool packed = transfer_mode == OMR_HOST_PACKED && d.lanes;
            const bool pinned_src = transfer_mode == OMR_HOST_PINNED;
            // scans per ring slot and DMA: 16 u8 scans (139 MB at A4), or as many packed ones as the slot holds
            const int chunk = packed ? std::min(d.launch, std::max(1, (int)((img * (size_t)std::min(HB_CHUNK, d.launch)) / pimg)))
                                     : std::min(HB_CHUNK, d.launch);
            const size_t unit = packed ? pimg : img;  // bytes of a scan in the ring and in the device stage
            hipStream_t main_s = d.ctx->streams[0];
            int slot_no = 0;
            for (int j0 = 0, li = 0; j0 < m; j0 += d.launch, li++) {
                const int g = std::min(d.launch, m - j0), b = li & 1;
                // the stage's previous launch must have packed it before new scans land on it
                if (d.stage_used[b]) OMR_HIP(hipStreamWaitEvent(d.copy, d.ev_used[b], 0));
                for (int c0 = 0; c0 < g; c0 += chunk) {
                    const int cg = std::min(chunk, g - c0);
                    uint8_t *dst = d.stage[b].as<uint8_t>() + (size_t)c0 * unit;
                    if (pinned_src) {  // the caller's memory is page-locked: DMA straight out of it
                        for (int z = 0; z < cg; z++) {
                            const omr_image &im = scans[mine[(size_t)(j0 + c0 + z)]];
                            if (im.step_bytes == cols)
                                OMR_HIP(hipMemcpyAsync(dst + (size_t)z * img, im.data, img, hipMemcpyHostToDevice, d.copy));
                            else
                                OMR_HIP(hipMemcpy2DAsync(dst + (size_t)z * img, (size_t)cols, im.data, (size_t)im.step_bytes, (size_t)cols,
                                                         (size_t)rows, hipMemcpyHostToDevice, d.copy));
                        }
                        continue;
                    }
                    const int k = slot_no++ % HB_SLOTS;
                    if (d.slot_busy[k]) OMR_HIP(hipEventSynchronize(d.ev_slot[k]));  // the slot's last DMA has read it
                    uint8_t *slot = d.ring[k];
                    pool.run(cg, [&](int z) {
                        const omr_image &im = scans[mine[(size_t)(j0 + c0 + z)]];
                        uint8_t *to = slot + (size_t)z * unit;
                        if (packed) pack_scan(im, rows, cols, (uint32_t *)to);
                        else if (im.step_bytes == cols) memcpy(to, im.data, img);
                        else
                            for (int r = 0; r < rows; r++) memcpy(to + (size_t)r * cols, im.data + (size_t)r * im.step_bytes, (size_t)cols);
                    });
                    OMR_HIP(hipMemcpyAsync(dst, slot, unit * (size_t)cg, hipMemcpyHostToDevice, d.copy));
                    OMR_HIP(hipEventRecord(d.ev_slot[k], d.copy));
                    d.slot_busy[k] = true;
                }
                OMR_HIP(hipEventRecord(d.ev_ready[b], d.copy));
                OMR_HIP(hipStreamWaitEvent(main_s, d.ev_ready[b], 0));
                int rc = packed ? batch_run_device_bits(d.ctx.get(), d.stage[b].as<uint32_t>(), (int64_t)pimg, g, d.dbest.as<int32_t>() + j0,
                                                        d.dvs.as<double>() + (size_t)j0 * A, d.dhs.as<double>() + (size_t)j0 * A)
                                : omr_batch_run_device(d.ctx.get(), d.stage[b].as<uint8_t>(), (int64_t)img, cols, g, 0,
                                                       d.dbest.as<int32_t>() + j0, d.dvs.as<double>() + (size_t)j0 * A,
                                                       d.dhs.as<double>() + (size_t)j0 * A);
                if (rc) return rc;
                OMR_HIP(hipEventRecord(d.ev_used[b], main_s));
                d.stage_used[b] = true;
            }
            int rc = omr_batch_sync(d.ctx.get());  // (also reads the sweep kernels' guard flags: OMR_ERR_GPU, never silent zeros)
            if (rc) return rc;
            OMR_HIP(hipStreamSynchronize(d.copy));
            std::vector<int32_t> hbest((size_t)m);
            OMR_HIP(hipMemcpy(hbest.data(), d.dbest.p, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
            std::vector<double> hv, hh;
            if (v_sd_opt) {
                hv.resize((size_t)m * A);
                OMR_HIP(hipMemcpy(hv.data(), d.dvs.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost));
            }
            if (h_sd_opt) {
                hh.resize((size_t)m * A);
                OMR_HIP(hipMemcpy(hh.data(), d.dhs.p, sizeof(double) * hh.size(), hipMemcpyDeviceToHost));
            }
            for (int j = 0; j < m; j++) {
                const int i = mine[(size_t)j];
                best_idx[i] = hbest[(size_t)j];
                if (v_sd_opt) memcpy(v_sd_opt + (size_t)i * A, hv.data() + (size_t)j * A, sizeof(double) * (size_t)A);
                if (h_sd_opt) memcpy(h_sd_opt + (size_t)i * A, hh.data() + (size_t)j * A, sizeof(double) * (size_t)A);
            }
            return OMR_OK;
        };
        rcs[(size_t)dv] = run();
        if (rcs[(size_t)dv]) errs[(size_t)dv] = last_error();
    };
    if (ND == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int dv = 0; dv < ND; dv++) th.emplace_back(worker, dv);
        for (auto &t : th) t.join();
    }
    for (int dv = 0; dv < ND; dv++)
        if (rcs[(size_t)dv]) return fail(rcs[(size_t)dv], "device %d: %s", dv, errs[(size_t)dv].c_str());
    if (best_angle)
        for (int i = 0; i < n; i++) best_angle[i] = ((double)best_idx[i] - (double)hb->N) * hb->step;  // projection.rs:189-190
    return OMR_OK;
}

#ifdef OMR_RUNS_DEBUG
int omr_debug_set_logical_devices(int k)
{
    if (k < 0 || k > 16) return fail(OMR_ERR_BADARG, "0..16 logical devices");
    g_logical_devices = k;
    return OMR_OK;
}
#endif

}  // extern "C"
