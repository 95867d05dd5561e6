// engine.hpp -- device-side engine objects behind the C ABI (include/omrdeskew.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <memory>
#include <mutex>
#include <string>
#include <map>
#include <vector>

#include "kernels.hpp"
#include "slane.hpp"

namespace omr {

// thread-local error channel (omr_last_error)
int fail(int code, const char *fmt, ...);
int fail_gpu(const char *what, hipError_t e);
const char *last_error();
void clear_error();

#define OMR_HIP(expr)                                          \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return ::omr::fail_gpu(#expr, e__); \
    } while (0)

// RAII device allocation.  Buffers allocated while a PoolScope is active on the calling thread (the
// per-call entry points open one around their private stream) come from a per-device cache and go
// back to it after that stream has drained: hipMalloc costs 0.1-0.5 ms and hipFree synchronises the
// whole device, which would serialise the host's worker threads (thread_pool.rs:41-54).  Long-lived
// owners (plans, batch contexts) allocate outside any scope and keep plain hipMalloc / hipFree.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n);
    void release();
    template <class T>
    T *as() const { return static_cast<T *>(p); }

  private:
    hipStream_t owner_ = nullptr;  // pooled: the stream whose work may still touch the buffer
    size_t cap_ = 0;               // pooled: size of the cached block
    int dev_ = -1;
};

// Marks the calling thread's allocations as short-lived work of `stream` (nests; restores on exit).
struct PoolScope {
    explicit PoolScope(hipStream_t stream);
    ~PoolScope();
    PoolScope(const PoolScope &) = delete;
    PoolScope &operator=(const PoolScope &) = delete;

  private:
    hipStream_t prev_;
    bool prev_on_;
};
// Suspends pooling on the calling thread: for allocations that outlive the call (plans, batch contexts).
struct NoPoolScope {
    NoPoolScope();
    ~NoPoolScope();

  private:
    hipStream_t prev_;
    bool prev_on_;
};

// A per-call stream with a pinned staging block, leased from a bounded per-device pool (engine.cpp).  The
// host's worker pool (thread_pool.rs:41-88) runs every task on a fresh OS thread, so neither may belong to a
// thread: creating a stream per call costs more than a 248x230 sweep and serialises in the runtime, keeping
// one per thread leaks it.  Leases nest (a driver that calls another entry point on its own stream).
struct CallSlot {
    hipStream_t stream = nullptr;  // non-blocking
    void *pinned = nullptr;        // grow-only download staging, at most 64 MB
    size_t cap = 0;
    int dev = 0;
    CallSlot *outer = nullptr;     // the calling thread's enclosing lease
};
int lease_call_slot(CallSlot **out);  // on the current device; fails for devices >= 16
void return_call_slot(CallSlot *c);   // the caller has drained c->stream (or only queued work that owns nothing)
void call_slot_stats(int dev, int *live, int *idle, size_t *idle_pinned);

// Device -> pageable host memory through the pinned staging block of the call's slot (the calling thread's
// innermost lease when it owns `s`, a temporary lease otherwise).  A direct hipMemcpy into a freshly
// malloc'ed result image makes the runtime pin the destination pages on the fly: 13 ms for the 5 MB rotated sheet
// of correct_default (72 files/s) against 0.6 ms through the staging buffer.  Synchronises `s`.
int staged_d2h(void *dst, const void *d_src, size_t bytes, hipStream_t s);
// same for a strided destination: `rows` rows of `row_bytes`, packed on the device, `dst_step` apart on the host
int staged_d2h_2d(void *dst, size_t dst_step, const void *d_src, size_t row_bytes, size_t rows, hipStream_t s);

// getRotationMatrix2D / warpAffine inversion on the host (fp64, built -ffp-contract=off, same
// libm as the caller's process) -- OpenCV 4.6.0 semantics, SURVEY.md A.1 / A.2 step 1.
void rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6]);
void invert_affine(const double M[6], double Minv[6]);
int candidate_count(uint16_t max_angle, double step, int *N_out);  // projection.rs:36-38
void sweep_matrices(int rows, int cols, int N, double step, double scale, double *M_out);
// rotate_mat's forward matrix and canvas (transfer.rs:459-523; oics_host.cpp)
int rotate_geometry(int rows, int cols, double angle_deg, double scale, int clip, double M[6], int *drows, int *dcols);

// Immutable per-(shape, matrices) state: inverse matrices, fixed-point tables, LDS tiling.
struct SweepTables {
    int device = 0;
    SweepDims dims{};
    DevBuf minv, adelta, bdelta, xy0, tiles;
    bool lds_ok = false;
    int max_rows_per_tile = 0;
    std::vector<double> host_minv;
    // run-merging ("S") kernel: per-(candidate, word) run tables and the split of the candidates
    // into run-merged ones and ones left to the gather kernels
    bool runs_built = false;
    int NWh = 0, Gh = 0;    // words per row, word groups (of OMR_RUN_K words)
    int GCh = 0, Ph = 0;    // word groups per workgroup chunk, chunks
    int RBh = 0, RCHh = 0;  // bands of 512 rows per row chunk, row chunks (1 unless the image is taller than 4608 rows)
    int NRp = 0;            // row pitch of the u16 row-count partials
    int GXh = 0, GYh = 0;   // zero guard of the transposed bit image: word columns left / right, rows above / below
    int NWt = 0, rowsT = 0; // its size with the guard: word columns, rows per word column
    DevBuf tabsH, metaH, metacH, blkH, wgeoH, list_runs, list_gather, mode;
    int n_runs = 0, n_gather = 0;
    std::vector<int32_t> host_mode;
    int create(int rows, int cols, const double *fwd_M, int A, int device);
    int build_runs();
    // arguments of the run-merging kernel for `scans` transposed bit images at d_bitsT
    RunPass run_pass(const uint32_t *d_bitsT, uint16_t *d_part, int scans) const;
};

// Mutable per-stream scratch: bit image, integer projections, scores.
struct SweepScratch {
    DevBuf bits, vproj, hproj, vsd, hsd, best;
    DevBuf hpart, guard;  // run-merging scratch: u16 row-count partials per chunk of word groups
    DevBuf bitsT;         // run-merging scratch: the bit images transposed (word columns contiguous)
    int zmax = 1;         // scans a launch may carry (every buffer above holds that many result sets)
    bool guard_pending = false;  // runs_kernel was launched on this set since its guard flags were last read
    int create(const SweepTables &t, int scans_per_launch = 1);
};

enum KernelSel { KERNEL_AUTO = 0, KERNEL_GENERIC = 1, KERNEL_LDS = 2, KERNEL_RUNS = 3 };

// Enqueue pack -> sweep -> std-dev -> arg-max for `scans` device-resident scans (img_stride bytes apart)
// in one launch of each kernel; scores / best index of scan z land at d_v_sd + z * A, d_best + z.
// host side of the kernels' guard flags: OMR_ERR_GPU when any of the n flags is set
int guard_verdict(const int32_t *flags, size_t n, const char *kernel);

int enqueue_sweep(const SweepTables &t, SweepScratch &s, int kernel_sel, const uint8_t *d_img, int64_t step,
                  int black_max, hipStream_t stream, uint32_t *d_vproj, uint32_t *d_hproj, double *d_v_sd,
                  double *d_h_sd, int32_t *d_best, hipEvent_t ev0, hipEvent_t ev1, bool want_proj = false,
                  hipStream_t post_stream = nullptr, hipEvent_t ev_mid = nullptr, int scans = 1, int64_t img_stride = 0);

// ---- scan-lane sweep (slane.hpp): lane = scan, for batches of same-shape scans
struct SlanePlan;
struct SlaneScratch;
struct SlanePlan {
    SlaneGeom g;
    int A = 0, nrec = 0, nexec = 0, hrow0 = 0;  // records numbered / in the streams; the number of image row 0's record
    bool built = false;
    int64_t prog_dwords = 0, null_seg = 0, null_fet = 0;  // the null program: empty words, nothing to fetch
    DevBuf prog, d_tasks;
    std::vector<SlaneStrip> strips;  // [A][NS]
    std::vector<int32_t> tasks;      // candidate * NS + strip, in launch order
    // the launch order in chunks of 32 candidates (slane_kernel's units): the work of one workgroup of a chunk and its size,
    // and the units dealt to the XCDs for every composition of a launch met so far (slane_deal_units, slane.hip)
    std::vector<double> chunk_weight;
    std::vector<int> chunk_size;
    struct UnitTab {
        DevBuf tab;
        int per_xcd = 0;
    };
    mutable std::map<int, UnitTab> unit_tabs;  // key = ncq (strip groups x groups of scan groups)
    mutable std::mutex unit_mu;
    int units_for(int ncq, const int32_t **d_tab, int *per_xcd) const;
    // OMR_ERR_NOTIMPL when a candidate does not fit the scheme.  on_host: slane_plan.cpp's generator (the reference
    // implementation, 16 host threads + upload) instead of slane_build.hip's
    int build(const SweepTables &t, bool on_host = false);
    void layout();                                 // seg_offset / fet_offset of every strip, the null program, prog_dwords
    int generate_on_device(const SweepTables &t);  // classes + programs by slane_build.hip
    int generate_on_host(const SweepTables &t);    // ... by slane_plan.cpp on the host's threads, then one upload
};
struct SlaneScratch {
    int nsg = 0;  // scan groups of 64 scans a launch may carry
    bool keep_rows = false, rows_dirty = false;  // inspection: leave the row counts in place after a launch
    bool guard_pending = false;                  // slane_kernel was launched on this set since its guard flag was last read
    size_t rows_bytes = 0;                       // of hrows: the row counts and, behind them, the totals [candidate][scan]
    uint32_t *bits_base = nullptr;  // the first scan group's bit image inside `bits` (aligned to SlaneGeom::group_stride())
    DevBuf bits, hrows, vproj, planes, descs[3], vsd, hsd, best, guard;  // descs[lg]: workgroups of (16 >> lg) strips x (1 << lg) scan groups
    int create(const SlanePlan &p, int groups);
};
int slane_enqueue(const SlanePlan &p, SlaneScratch &s, const uint8_t *d_img, int64_t scan_stride, int64_t step, int nscans,
                  int black_max, hipStream_t stream, hipStream_t post_stream, hipEvent_t ev_mid, double *d_v_sd, double *d_h_sd,
                  int32_t *d_best, hipEvent_t ev0, hipEvent_t ev1);

}  // namespace omr

struct omr_batch_ctx;
namespace omr {
int batch_run_device_bits(omr_batch_ctx *ctx, const uint32_t *d_bits, int64_t scan_stride_bytes, int32_t n, int32_t *d_best_idx,
                          double *d_v_sd, double *d_h_sd);
}

struct omr_sweep_plan {
    omr::SweepTables tables;
    omr::SweepScratch scratch;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false, timed = false;
    int kernel_sel = omr::KERNEL_AUTO;
    omr::DevBuf img;  // staging for the host-image entry point
    std::mutex mu;
    ~omr_sweep_plan();
};

struct omr_batch_ctx {
    omr::SweepTables tables;
    int N = 0;
    double step = 0;
    // per main stream: two scratch sets used alternately, a post stream for the latency-bound
    // std-dev / arg-max kernels (they overlap the next scan's sweep), and the events that order them
    std::vector<std::unique_ptr<omr::SweepScratch>> scratch;  // [2 * n_streams]
    std::vector<hipStream_t> streams, post_streams;
    std::vector<hipEvent_t> ev_mid, ev_post;                   // [2 * n_streams]
    std::vector<char> post_pending;                            // [2 * n_streams]
    std::vector<uint64_t> issued;                              // scans issued per main stream
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    bool timing = false;
    int group = 1;  // scans per kernel launch (omr_batch_set_group)
    // scan-lane sweep (omr_batch_set_lanes): 64 scans per wavefront, up to `lanes` scans per launch
    int lanes = 0;
    omr::SlanePlan slane;
    std::vector<std::unique_ptr<omr::SlaneScratch>> slane_scratch;  // [2 * n_streams]
    // final deskew (omr_batch_deskew_device): per candidate the CONTAIN canvas and warpAffine's fixed-point tables
    bool dk_built = false;
    int dk_rows = 0, dk_cols = 0;  // largest canvas (cols rounded up to 4)
    omr::DevBuf dk_size, dk_adelta, dk_bdelta, dk_xy0;
    std::vector<std::unique_ptr<omr::DevBuf>> dk_tiles;  // per post stream: the warp's per-tile records (deskew.hip)
    std::mutex mu;
    std::mutex guard_mu;  // omr_batch_sync is called with and without `mu` held: the guard flags have a lock of their own
    ~omr_batch_ctx();
};
