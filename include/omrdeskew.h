/*
 * omrdeskew.h -- C ABI of libomrdeskew.so, the MI355X-native OMR deskew engine.
 *
 * This is the drop-in boundary for the projection-std-dev angle sweep of
 * ch1ny/omr-img-corrector (crate `oics`, packages/lib).  The reference has no FFI layer of its
 * own (it calls OpenCV through the `opencv` crate); a Rust shim crate `oics` binds these entry
 * points 1:1 in an `extern "C"` block (INTEGRATION.md shows it).  Every entry point cites the
 * reference interface it replaces as file:line under /root/reference.
 *
 * Conventions
 *   - images are 8-bit, row-major, `channels` interleaved, `step_bytes >= cols*channels`;
 *     inputs are borrowed and never written; outputs are caller-allocated unless stated.
 *   - return value: 0 on success, otherwise a negative OpenCV-style code
 *     (-215 assertion / bad shape, -5 bad argument, -4 out of memory, -213 not implemented,
 *      -217 GPU API error).  omr_last_error() gives the thread-local message.
 *   - no exceptions and no aborts cross the ABI; all entry points are thread-safe and
 *     re-entrant (the Tauri host runs several corrections at once: thread_pool.rs:41-88).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point
 *     fails with -217.
 *   - `_device` entry points take device pointers (hipMalloc'ed, same device as the plan) and a
 *     hipStream_t passed as void*; they enqueue work and return without synchronising.
 */
#ifndef OMRDESKEW_H
#define OMRDESKEW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMR_OK 0
#define OMR_ERR_ASSERT (-215)
#define OMR_ERR_BADARG (-5)
#define OMR_ERR_NOMEM (-4)
#define OMR_ERR_NOTIMPL (-213)
#define OMR_ERR_GPU (-217)

/* Borrowed image view: stands for `&opencv::core::Mat` / `&TransformableMatrix`
 * (packages/lib/src/transfer.rs:16-18). */
typedef struct {
    const uint8_t *data;
    int32_t rows, cols, channels;
    int64_t step_bytes;
} omr_image;

/* Owned image returned by the library (free with omr_image_free): stands for the fresh
 * `TransformableMatrix` every reference helper returns (transfer.rs:55-59). */
typedef struct {
    uint8_t *data;
    int32_t rows, cols, channels;
    int64_t step_bytes;
} omr_image_owned;

/* types.rs:7-11 RotateClipStrategy */
#define OMR_CLIP_DEFAULT 0
#define OMR_CLIP_CONTAIN 1
/* omr.rs:41-45 ResultStatus */
#define OMR_STATUS_BELIEVED 0
#define OMR_STATUS_NEED_CHECK 1
#define OMR_STATUS_NOT_A_RESULT 2
/* interpolation flags as the reference passes them (0 = INTER_NEAREST = WARP_POLAR_LINEAR's
 * numeric value, transfer.rs / projection.rs:52; 1 = INTER_LINEAR, core/src/main.rs:76) */
#define OMR_INTER_NEAREST 0
#define OMR_INTER_LINEAR 1
#define OMR_INTER_AREA 3 /* imgproc::INTER_AREA: resize only */

int omr_version(void);
int omr_device_count(void);
const char *omr_last_error(void);
void omr_image_free(omr_image_owned *img);

/* ---- geometry helpers ---------------------------------------------------------------- */

/* imgproc::get_rotation_matrix_2d(center: Point2f, angle, scale) -> 2x3 f64;
 * call sites transfer.rs:475,501; omr.rs:159-163,425-426. */
int omr_get_rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6]);

/* `(max_angle as f64 / step) as u16` and the half-open candidate range -N..N
 * (projection.rs:36-38, omr.rs:140-145).  Returns A = 2N (>= 0); *N_out = N. */
int omr_candidate_count(uint16_t max_angle, double step, int32_t *N_out);

/* The A forward matrices of a sweep over an image of rows x cols: centre
 * (cols as f32 / 2, rows as f32 / 2), angle (i - N) * step, scale (transfer.rs:473-475;
 * omr.rs:157-163 passes scale = projection_resize_scale).  M_out: A x 6 doubles. */
int omr_sweep_matrices(int32_t rows, int32_t cols, uint16_t max_angle, double step, double scale,
                       double *M_out, int32_t cap_A);

/* ---- the hot path ---------------------------------------------------------------------- */

/* One scan, A candidate matrices, host buffers.  For every candidate: nearest-neighbour
 * warpAffine of the binarised image onto the same canvas with a white border, per-column and
 * per-row count of pixels == 0, population std-dev of each projection.
 * Replaces the body of projection.rs:47-65 / omr.rs:153-180:
 *   rotate_mat DEFAULT (transfer.rs:459-486) -> get_projection_standard_deviations
 *   (transfer.rs:527-536) or get_mat_projection_data (omr.rs:8-39) + calculate.rs:13-23.
 * bin_u8c1: 1 channel; a pixel is black iff its value == 0.
 * fwd_M: A x 6 row-major forward matrices as produced by getRotationMatrix2D.
 * vproj (A x cols) and hproj (A x rows) may be NULL; v_sd / h_sd: A doubles each. */
int omr_projection_sweep(const omr_image *bin_u8c1, const double *fwd_M, int32_t A,
                         uint32_t *vproj, uint32_t *hproj, double *v_sd, double *h_sd);

/* Resident form of the same path: device buffers and fixed-point tables are created once per
 * (rows, cols, matrices) and reused for every scan of that shape. */
typedef struct omr_sweep_plan omr_sweep_plan;

int omr_sweep_plan_create(int32_t rows, int32_t cols, const double *fwd_M, int32_t A,
                          int32_t device, omr_sweep_plan **plan_out);
int omr_sweep_plan_create_angles(int32_t rows, int32_t cols, uint16_t max_angle, double step,
                                 double scale, int32_t device, omr_sweep_plan **plan_out);
void omr_sweep_plan_destroy(omr_sweep_plan *plan);
int omr_sweep_plan_candidates(const omr_sweep_plan *plan); /* A */

/* Enqueue one scan on `stream`.  d_img: device u8, 1 channel, rows x cols, row pitch
 * step_bytes.  A pixel is black iff value <= black_max: pass 0 for an image binarised by
 * transfer_gray_image_to_thresh_binary (transfer.rs:294-301), 127 to fuse that threshold
 * (threshold(127,255,BINARY) then == 0  <=>  gray <= 127) into the load.
 * Outputs are device pointers; any of d_vproj (A x cols u32), d_hproj (A x rows u32),
 * d_best_idx (1 int32: projection.rs:125-190 arg-max, lowest index on exact ties) may be NULL.
 * d_v_sd / d_h_sd: A doubles each. */
int omr_sweep_plan_run_device(omr_sweep_plan *plan, const uint8_t *d_img, int64_t step_bytes,
                              int32_t black_max, void *stream, uint32_t *d_vproj,
                              uint32_t *d_hproj, double *d_v_sd, double *d_h_sd,
                              int32_t *d_best_idx);

/* Same, host image in / host results out (synchronous: H2D, sweep, D2H on `plan`'s stream). */
int omr_sweep_plan_run(omr_sweep_plan *plan, const omr_image *img_u8c1, int32_t black_max,
                       uint32_t *vproj, uint32_t *hproj, double *v_sd, double *h_sd,
                       int32_t *best_idx);

/* Duration in ms of the sweep kernel of the most recent omr_sweep_plan_run*(), measured with
 * HIP events on the stream it was launched on (bench.py's roofline line). Synchronises. */
int omr_sweep_plan_last_kernel_ms(omr_sweep_plan *plan, float *ms_out);
/* Toggle event recording around the sweep kernel (off by default). */
int omr_sweep_plan_set_timing(omr_sweep_plan *plan, int32_t enabled);
/* Select the sweep kernel: 0 = automatic (run-merging kernel for every candidate that qualifies,
 * LDS-staged or generic gather kernel for the rest), 1 = generic gather kernel for all, 2 = LDS-staged
 * gather kernel for all (fails with -5 when a candidate's source footprint does not fit), 3 = as 0 but
 * fails with -5 when no candidate qualifies for the run-merging kernel.  For tests and profiling. */
int omr_sweep_plan_set_kernel(omr_sweep_plan *plan, int32_t which);
/* How the plan splits its candidates: *n_runs are swept by the run-merging kernel (small-angle
 * rotations), *n_gather by the per-sample gather kernels.  Either pointer may be NULL. */
int omr_sweep_plan_info(const omr_sweep_plan *plan, int32_t *n_runs, int32_t *n_gather);
/* Debug/parity hook: copy the fixed-point tables of candidate `a` to host
 * (adelta, bdelta: cols ints; X0, Y0: rows ints) -- OpenCV hal::warpAffine's tables. */
int omr_sweep_plan_tables(omr_sweep_plan *plan, int32_t a, int32_t *adelta, int32_t *bdelta,
                          int32_t *X0, int32_t *Y0);

/* Batch of n scans (all rows x cols, 1 channel) resident on the plan's device, scan i at
 * d_scans + i * scan_stride_bytes.  Scans are independent (one correct_default per file,
 * app/src-tauri/src/task.rs:19-38): they are issued round-robin on `n_streams` internal
 * streams; no collective is involved.  Outputs are device pointers: d_best_idx (n int32),
 * d_v_sd / d_h_sd (n x A doubles, may be NULL).  Returns after enqueueing; call
 * omr_batch_sync() or synchronise the device before reading. */
typedef struct omr_batch_ctx omr_batch_ctx;
int omr_batch_create(int32_t rows, int32_t cols, uint16_t max_angle, double step, double scale,
                     int32_t device, int32_t n_streams, omr_batch_ctx **ctx_out);
void omr_batch_destroy(omr_batch_ctx *ctx);
int omr_batch_run_device(omr_batch_ctx *ctx, const uint8_t *d_scans, int64_t scan_stride_bytes,
                         int64_t step_bytes, int32_t n, int32_t black_max, int32_t *d_best_idx,
                         double *d_v_sd, double *d_h_sd);
int omr_batch_sync(omr_batch_ctx *ctx);
/* Scans carried by one launch of each kernel (default 1; 1..64).  Larger groups amortise the fixed
 * cost between dependent launches (about 30 us per sweep at 2480x3508) over more work; the scratch
 * grows accordingly (about 40 MB per scan of the group at 2480x3508).  Results do not change.
 * Prefer 1, 2, 4, 8 or a multiple of 8: the sweep's grid runs scans fastest and workgroups go
 * round-robin to the 8 XCDs, so each XCD then sees one scan (or two) and keeps its bit image in L2. */
int omr_batch_set_group(omr_batch_ctx *ctx, int32_t scans_per_launch);
/* Split of the candidates between the run-merging kernel and the gather kernels (see
 * omr_sweep_plan_info). */
int omr_batch_info(const omr_batch_ctx *ctx, int32_t *n_runs, int32_t *n_gather);
/* Sum over launch groups of the sweep-stage duration (ms; HIP events on the launch stream around
 * the sweep kernels of a group: one run-merging launch -- all scans of the group, both projections
 * -- plus one gather launch per scan when some candidates do not qualify) and the number of groups
 * timed since the last call (timing must be enabled with omr_batch_set_timing).  Synchronises. */
int omr_batch_set_timing(omr_batch_ctx *ctx, int32_t enabled);
int omr_batch_kernel_ms(omr_batch_ctx *ctx, double *sum_ms, int32_t *launches);
/* The reference's whole unit of work for a batch (omr.rs:339-452 correct_default; core/src/main.rs:69-92): detect
 * every scan's angle, then rotate the scan by it with CONTAIN geometry (transfer.rs:487-519), border_value
 * outside.  Sweep, arg-max and warp stay on the device: the warp reads the winning candidate's index there, and
 * its fixed-point tables exist per candidate (matrices from the host's libm, like omr_rotate*).  interp is
 * OMR_INTER_NEAREST (omr.rs:408-445) or OMR_INTER_LINEAR (core/src/main.rs:72-81).  Scan i's canvas fills the
 * top-left dst_rows x dst_cols pixels of d_out + i * out_stride_bytes (rows out_step_bytes apart) and its size
 * lands in d_out_size[2 i] (rows) and [2 i + 1] (cols); every slot must hold the largest canvas of the candidate
 * set, omr_batch_deskew_canvas().  The result is what omr_rotate_device(angle = (best_idx - N) * step, scale 1,
 * OMR_CLIP_CONTAIN) writes for the same scan, bit for bit.  d_best_idx (n int32) and d_out_size (2 n int32) may be
 * NULL.  Returns after enqueueing, like omr_batch_run_device.
 * Performance note: the warp stages every tile's source box in LDS with dword loads, which needs d_scans,
 * scan_stride_bytes, step_bytes and the scan width to be multiples of 4; otherwise (e.g. a 453-column scan, tightly
 * packed) every tile takes a byte-wise per-tap path straight from memory -- same result, several times slower. */
int omr_batch_deskew_canvas(omr_batch_ctx *ctx, int32_t *max_rows, int32_t *max_cols);
int omr_batch_deskew_device(omr_batch_ctx *ctx, const uint8_t *d_scans, int64_t scan_stride_bytes,
                            int64_t step_bytes, int32_t n, int32_t black_max, int32_t interp,
                            uint8_t border_value, uint8_t *d_out, int64_t out_stride_bytes,
                            int64_t out_step_bytes, int32_t *d_out_size, int32_t *d_best_idx);
/* Streams and pinned staging blocks the per-call entry points lease from a bounded per-device pool (a host
 * that runs every task on a fresh OS thread, thread_pool.rs:41-88, must not leak one of each per call):
 * slots that exist, slots idle in the pool, pinned bytes held by idle slots.  Pointers may be NULL. */
int omr_call_pool_stats(int32_t device, int32_t *live_slots, int32_t *idle_slots, int64_t *idle_pinned_bytes);

/* ---- scan-lane sweep (batches) ------------------------------------ */
/* The batch path sweeps 64 scans per wavefront (lane = scan): the geometry of a candidate -- which source row,
 * word column, shift and destination bits make up every destination word (projection.rs:47-65 ->
 * transfer.rs:459-486) -- is one wave-uniform PROGRAM per (candidate, strip of two word columns), enumerated
 * from warpAffine's integer tables.  This entry point builds one strip's program on the HOST (no GPU needed):
 * n_records rows (pre_rows virtual ones first) of seg_dwords_per_row dwords in the segment stream and of 4 dwords
 * in the fetch stream (layout: csrc/slane.hpp); guard_cols / guard_rows = the zero guard (word columns, rows) the
 * program's entry numbers assume around the interleaved bit image.  NULL output pointers query the sizes.
 * OMR_ERR_NOTIMPL when the strip does not fit the scheme (such candidates stay with the run-merging kernel). */
int omr_slane_strip_program(int32_t rows, int32_t cols, const double *fwd_M, int32_t strip, uint32_t *seg_out,
                            uint32_t *fetch_out, int32_t *seg_dwords_per_row, int32_t *n_records, int32_t *pre_rows,
                            int32_t *most_segments, int32_t *guard_cols, int32_t *guard_rows);

/* Switch a batch context to the scan-lane sweep: every launch then carries up to max_scans_per_launch scans (whole
 * groups of 64, at most 4096), 64 scans per wavefront; results are bit-identical to the run-merging path.  The
 * programs of all (candidate, strip) pairs are generated on the device, once (about 45 ms and 4.7 GB of HBM for
 * 2480x3508 with 400 candidates).  A launch of up to 64 / 128 / 256 scans takes about 5.9 / 10.4 / 19 ms at that size:
 * size launches in multiples of 64.  0 switches back.  OMR_ERR_NOTIMPL, context unchanged, when a candidate does not
 * fit the scheme (beyond about +-10 degrees at unit scale, or more than 8166 rows). */
int omr_batch_set_lanes(omr_batch_ctx *ctx, int32_t max_scans_per_launch);
/* Bytes of programs the context's scan-lane plan holds in HBM, its (candidate, strip) tasks, scans per launch
 * (0 = the context is on the run-merging path).  Pointers may be NULL. */
int omr_batch_lanes_info(omr_batch_ctx *ctx, int64_t *program_bytes, int32_t *tasks, int32_t *scans_per_launch);
/* Inspection: with on != 0 a launch leaves its row counts in the scratch set (needed by
 * omr_batch_lanes_projections); by default they are cleared behind the std-dev kernel, off the sweep's stream. */
int omr_batch_lanes_keep(omr_batch_ctx *ctx, int32_t on);
/* The scan-lane programs of a context are generated on the device; this regenerates them with the host generator (the
 * reference implementation behind omr_slane_strip_program) and compares: *dwords = dwords of programs, *differing = how
 * many differ (0 = identical).  For tests and inspection; seconds at A4. */
int omr_batch_lanes_check_programs(omr_batch_ctx *ctx, int64_t *dwords, int64_t *differing);

/* The integer projections one scan / candidate of the last scan-lane launch left in scratch set `set` (0 for the
 * first launch of a stream): vproj = cols counts, hproj = rows counts; either may be NULL.  Synchronises. */
int omr_batch_lanes_projections(omr_batch_ctx *ctx, int32_t set, int32_t scan, int32_t a, uint32_t *vproj,
                                uint32_t *hproj);

/* ---- batches that start in host memory ("wall-clock including H2D"; core/src/main.rs:68-95) --- */
/* A context for repeated host-memory batches of one shape and sweep: per device the sweep plan (the scan-lane sweep when
 * the candidates fit it and at least 64 scans per device are expected, else the run-merging path), a pinned ring of three
 * 16-scan slots and two device stages of one launch (64 scans) each.  n_devices <= 0 = every visible device,
 * n_devices > omr_device_count() is OMR_ERR_BADARG.  max_scans = the largest n a run may carry. */
typedef struct omr_host_batch omr_host_batch;
int omr_host_batch_create(int32_t rows, int32_t cols, uint16_t max_angle, double step, int32_t n_devices,
                          int32_t max_scans, omr_host_batch **out);
void omr_host_batch_destroy(omr_host_batch *hb);
int omr_host_batch_info(const omr_host_batch *hb, int32_t *n_devices, int32_t *scans_per_launch, int32_t *scan_lane);
/* scans[i] -> device i % n_devices: host memory -> device stage -> sweep; the transfer of one launch overlaps the sweep of
 * the previous one.  Pixels == 0 are black (binarised scans, as omr_sweep_batch).  transfer_mode:
 *   OMR_HOST_PAGEABLE  the context's copier threads move the scans into its pinned ring, DMA from there;
 *   OMR_HOST_PINNED    the caller's memory is page-locked: DMA straight out of it;
 *   OMR_HOST_PACKED    the copier threads -- which touch every byte anyway -- turn each binarised scan into 1 bit per pixel
 *                      on its way into the ring: 1/8 of the bytes cross the link (1.09 MB per A4 scan), a device kernel
 *                      interleaves the packed rows into the scan-lane image.  Contexts on the run-merging path (fewer than
 *                      64 scans per device) transfer the bytes as they are.
 * The three give the same results, bit for bit.  Outputs as omr_sweep_batch. */
#define OMR_HOST_PAGEABLE 0
#define OMR_HOST_PINNED 1
#define OMR_HOST_PACKED 2
int omr_host_batch_run(omr_host_batch *hb, const omr_image *scans, int32_t n, int32_t transfer_mode,
                       int32_t *best_idx, double *best_angle, double *v_sd_opt, double *h_sd_opt);
/* Scans per sweep launch (scan-lane contexts: rounded up to a multiple of 64, at most 512; default 64, which suits the
 * copy-bound u8 modes -- the last launch's sweep is what nothing overlaps; packed transfers are sweep-bound and larger
 * launches sweep faster per scan).  Re-sizes the device stages and the sweep's scratch. */
int omr_host_batch_set_launch(omr_host_batch *hb, int32_t scans_per_launch);

/* Host-buffer batch over the visible devices (SURVEY.md 8b `omr_sweep_batch`): scans[i] goes
 * to device i % n_devices (an omr_host_batch made for this one call: plan and ring creation are inside the call; packed transfers).
 * The scans may have DIFFERENT shapes (the reference corrects one file per call, any size, task.rs:19-38): they are bucketed by
 * (rows, cols), one context per shape, and the results land at the scans' own positions;
 * the only "collective" is the host-side gather of the results.  n_devices <= 0 = every visible device;
 * n_devices > omr_device_count() is OMR_ERR_BADARG (never a silent clamp).
 * best_angle[i] = (best_idx[i] - N) * step (projection.rs:189-190). */
int omr_sweep_batch(const omr_image *scans, int32_t n, uint16_t max_angle, double step,
                    int32_t n_devices, int32_t *best_idx, double *best_angle, double *v_sd_opt,
                    double *h_sd_opt);

/* ---- drivers with the reference's signatures ------------------------------------------ */

/* oics::projection::get_angle_with_projections(&TransformableMatrix, u16, f64, f64, usize) -> f64
 * (projection.rs:17-23).  src: 3 or 4 channels as the reference requires (cvtColor RGB2GRAY);
 * 1 channel is accepted and skips the conversion.  threads_hint is ignored (the reference's
 * multi-thread branch is buggy, projection.rs:94, and no caller uses it). */
int omr_get_angle_with_projections(const omr_image *src, uint16_t max_angle, double step,
                                   double resize_scale, size_t threads_hint, double *angle_out);

/* find_target_angle(max_angle, step, thresh, threads) -> f64 on an already binarised image
 * (app/src-tauri/src/test.rs:83-178, the in-app copy of the same driver). */
int omr_find_target_angle(uint16_t max_angle, double step, const omr_image *thresh_u8c1,
                          size_t threads_hint, double *angle_out);

/* oics::omr::get_result_from_projection(&Mat, u16, f64, i32, i32) -> Result<OmrResult>
 * (omr.rs:52-229).  candidates receives min(cand_len, cand_cap) angles. */
int omr_get_result_from_projection(const omr_image *src, uint16_t max_angle, double step,
                                   int32_t max_w, int32_t max_h, double *angle, int32_t *status,
                                   double *candidates, int32_t cand_cap, int32_t *cand_len);

/* projection.rs:125-190 on host arrays (also used by the drivers above). */
int omr_argmax_projection(const double *v_sd, const double *h_sd, int32_t n, int32_t *index_out);
/* Same policy for scores that live on the device (enqueue only; d_index_out: 1 int32). */
int omr_argmax_projection_device(const double *d_v_sd, const double *d_h_sd, int32_t n,
                                 int32_t *d_index_out, void *stream);
/* omr.rs:147-221 on host arrays. */
int omr_select_projection_result(const double *v_sd, const double *h_sd, int32_t n, int32_t N,
                                 double step, double *angle, int32_t *status, double *candidates,
                                 int32_t cand_cap, int32_t *cand_len);

/* ---- per-image helpers of crate `oics` (GPU-backed) ------------------------------------- */

/* transfer::transfer_gray_image_to_thresh_binary (transfer.rs:294-301): dst = src>127 ? 255 : 0 */
int omr_threshold_binary(const omr_image *gray_u8c1, uint8_t *dst, int64_t dst_step_bytes);
/* transfer::transfer_rgb_image_to_gray_image (transfer.rs:283-290): cvtColor RGB2GRAY on 3/4 ch */
int omr_rgb_to_gray(const omr_image *src, uint8_t *dst, int64_t dst_step_bytes);
/* transfer::rotate_mat (transfer.rs:459-523); border_value = Scalar(b, g, r, a) as u8. */
int omr_rotate(const omr_image *src, double angle_deg, double scale, int32_t interp,
               const uint8_t border_value[4], int32_t clip, omr_image_owned *dst);
/* transfer::get_vertical_projection / get_horizontal_projection (transfer.rs:380-405, :305-333) */
int omr_get_vertical_projection(const omr_image *bin_u8c1, double *out_cols);
int omr_get_horizontal_projection(const omr_image *bin_u8c1, double *out_rows);
/* omr::get_mat_projection_data (omr.rs:8-39): (horizontal[rows], vertical[cols]) */
int omr_get_mat_projection_data(const omr_image *bin_u8c1, double *h_rows, double *v_cols);
/* transfer::get_projection_standard_deviations (transfer.rs:527-536): (vertical, horizontal) */
int omr_get_projection_standard_deviations(const omr_image *bin_u8c1, double *v_sd, double *h_sd);
/* TransformableMatrix::scale_self (transfer.rs:66-91): new size = ((w * scale) as i32, (h * scale) as i32),
 * INTER_LINEAR when scale > 1, INTER_AREA otherwise; scale == 1.0 returns a copy.
 * shrink_to (transfer.rs:93-126; the Projection phase of the in-app benchmark, app/src-tauri/src/test.rs:313):
 * scale = min(max_width / w, max_height / h) (a bound <= 0 means "no bound"), applied only when < 1.
 * resize_self (transfer.rs:128-145): resize to (width, height) with INTER_AREA.
 * The reference mutates `self`; here the result is a new owned image (omr_image_free). */
int omr_scale(const omr_image *src, double scale, omr_image_owned *dst);
int omr_shrink_to(const omr_image *src, int32_t max_width, int32_t max_height, omr_image_owned *dst);
int omr_resize(const omr_image *src, int32_t width, int32_t height, omr_image_owned *dst);

/* ---- the same stages on device-resident images (enqueue on `stream`, no synchronisation) ----
 * These are the building blocks of a fully resident pipeline: front end of omr.rs:87-139,
 * sweep (omr_sweep_plan_run_device / omr_batch_run_device), final deskew of omr.rs:408-445 /
 * transfer.rs:487-519.  All pointers are device pointers; steps are row pitches in bytes. */
int omr_rgb_to_gray_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols,
                           int32_t channels, uint8_t *d_dst, int64_t dst_step, void *stream);
/* erode(3x3 MORPH_ELLIPSE = cross, iterations = 3, BORDER_CONSTANT, default border): omr.rs:98-112 */
int omr_erode3_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols,
                      uint8_t *d_dst, int64_t dst_step, void *stream);
/* resize(src, dst size, INTER_AREA): transfer.rs:128-145 / omr.rs:114-126.  Integer shrink factors (the
 * callers' 0.2, 1240x1150 -> 248x230) run resizeAreaFast_; fractional shrink factors build resizeArea_'s
 * tap tables on the host and synchronise `stream` before returning; when an axis ENLARGES OpenCV emulates
 * INTER_AREA with its bilinear kernel (omr.rs:60-82 has no clamp on the scale: quirk B7) -- same here. */
int omr_resize_area_device(const uint8_t *d_src, int64_t src_step, int32_t src_rows, int32_t src_cols,
                           int32_t channels, uint8_t *d_dst, int64_t dst_step, int32_t dst_rows,
                           int32_t dst_cols, void *stream);
int omr_threshold_binary_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols,
                                uint8_t *d_dst, int64_t dst_step, void *stream);
/* rotate_mat's canvas (transfer.rs:472-498): DEFAULT keeps the size, CONTAIN grows it. */
int omr_rotate_size(int32_t rows, int32_t cols, double angle_deg, int32_t clip, int32_t *dst_rows,
                    int32_t *dst_cols);
/* rotate_mat on device buffers; dst_rows/dst_cols must equal omr_rotate_size()'s answer. */
int omr_rotate_device(const uint8_t *d_src, int64_t src_step, int32_t rows, int32_t cols,
                      int32_t channels, double angle_deg, double scale, int32_t interp,
                      const uint8_t border_value[4], int32_t clip, uint8_t *d_dst, int64_t dst_step,
                      int32_t dst_rows, int32_t dst_cols, void *stream);

/* ---- Hough-line deskew path ---------------------------------------------
 * OpenCV 4.6.0 semantics restated on the GPU: Canny is exact (integer stencils + a set-valued
 * hysteresis); HoughLinesP keeps hough.cpp's point order (cv::RNG seed 2^64-1), float32 votes and
 * 16.16 line walks, so the segments are the reference's segments, not an approximation. */

/* imgproc::canny(src, &mut edges, low, high, 3, false): call sites hough.rs:27 (50, 150 on the gray
 * scan), omr.rs:239 (on the 3-channel scan: per pixel the channel with the largest |dx|+|dy|),
 * omr.rs:323-330.  edges: 1 channel, 0 / 255. */
int omr_canny(const omr_image *src, double low_thresh, double high_thresh, omr_image_owned *edges);

/* imgproc::hough_lines_p(&edges, &mut lines, rho, theta, threshold, min_line_length, max_line_gap):
 * hough.rs:31-43, omr.rs:245-253 (rho 1, theta pi/180, threshold 0).  lines: cap x (x0, y0, x1, y1);
 * *n_lines = segments found (call again with a larger buffer if it exceeds cap).  theta must give
 * at most 256 accumulator angles (-213 otherwise). */
int omr_hough_lines_p(const omr_image *edges_u8c1, double rho, double theta, int32_t threshold,
                      double min_line_length, double max_line_gap, int32_t *lines, int32_t cap,
                      int32_t *n_lines);

/* oics::hough::get_angle_with_hough(&TransformableMatrix, min_line_length, max_line_gap, file_name,
 * edge_image_output_dir) -> Result<f64> (hough.rs:17-100; called core/src/main.rs:103-110,
 * app/src-tauri/src/test.rs:383-390).  The debug picture the reference writes with imwrite
 * (hough.rs:46-63, :94-99) is host-side codec work and stays with the caller, hence no file
 * arguments.  No segment found: -215 (the reference panics on angles[0], hough.rs:74). */
int omr_get_angle_with_hough(const omr_image *gray, double min_line_length, double max_line_gap,
                             double *angle_out);

/* oics::omr::get_result_from_edges_detection(&Mat, f64, f64) -> Result<OmrResult> (omr.rs:231-302). */
int omr_get_result_from_edges_detection(const omr_image *src, double edges_min_line_length,
                                        double edges_max_line_gap, double *angle, int32_t *status,
                                        double *candidates, int32_t cand_cap, int32_t *cand_len);

/* The same on n device-resident scans of one shape (one workgroup per scan in the sequential Hough
 * stage): BASELINE config 4.  angles / status / n_lines: host arrays of n; a scan without any
 * segment reports status NotAResult and angle 0 instead of the reference's panic. */
int omr_edges_detection_batch_device(const uint8_t *d_scans, int32_t n, int64_t scan_stride_bytes,
                                     int32_t rows, int32_t cols, int32_t channels, int64_t step_bytes,
                                     double min_line_length, double max_line_gap, double *angles,
                                     int32_t *status, int32_t *n_lines, void *stream);

/* Tuning knob of the batch above, process-wide: how many scans the sequential Hough stage works on at once
 * (= workgroups of its kernel; each finished workgroup takes the next scan of the batch).  Every scan in
 * flight keeps an accumulator (2.8 MB at A4) and a point mask (1.1 MB) hot, so the count trades cache
 * footprint against occupied compute units.  0 = the library's default (profiles/r03_hough.md).  Returns
 * the previous setting.  No counterpart in the reference (its OpenCV call is one scan per thread). */
int32_t omr_hough_set_scans_in_flight(int32_t scans);

/* The decision of correct_default (omr.rs:351-399): which angle to rotate by and whether the sheet
 * needs a manual check, from the projection result and the edges result. */
void omr_correct_default_decision(double proj_angle, int32_t proj_status, const double *proj_candidates,
                                  int32_t n_cand, double edges_angle, double *rotate_angle,
                                  int32_t *need_check);

/* oics::omr::correct_default(input_file, output_file, u16, f64, i32, i32, f64, f64) ->
 * Result<(f64, bool)> (omr.rs:339-448; called app/src-tauri/src/task.rs:38-47) on a decoded BGR
 * image: imread / imwrite stay on the host side of the shim.  rotated may be NULL. */
int omr_correct_default(const omr_image *src_bgr, uint16_t projection_max_angle,
                        double projection_angle_step, int32_t projection_max_width,
                        int32_t projection_max_height, double hough_min_line_length,
                        double hough_max_line_gap, double *rotate_angle, int32_t *need_check,
                        omr_image_owned *rotated);

/* ---- FFT deskew path ------------------------------------------------------
 * The 2-D DFT is float32 like the reference's (dft on CV_32F); it is a different factorisation than
 * OpenCV's (radix-2 Stockham / Bluestein chirp-z in LDS), so the 8-bit spectrum pictures agree with
 * the CPU restatement to about one grey level, not bit for bit.  Everything after the picture
 * (Canny, HoughLinesP, votes) is the exact chain of the Hough-line path.  Any axis length up to the image
 * limit (32766) is transformed: lengths <= 8192 and the power of two 16384 inside LDS, longer lines by a chirp-z
 * through global memory (slower: a 600-dpi A3 scan, 9921 x 14032, takes about 12 ms). */

/* oics::fft::get_fft_image(&TransformableMatrix) -> Result<(Mat, Mat)> (fft.rs:124-141):
 * (magnitude_image, magnitude_log_image), both 8-bit single channel.  Either output may be NULL. */
int omr_get_fft_image(const omr_image *gray_u8c1, omr_image_owned *magnitude_image,
                      omr_image_owned *magnitude_log_image);

/* The magnitude_log pictures of n device-resident scans of one shape (BASELINE config 5):
 * d_magnitude_log: n x rows x cols bytes, packed. */
int omr_fft_image_batch_device(const uint8_t *d_scans, int32_t n, int64_t scan_stride_bytes, int32_t rows,
                               int32_t cols, int64_t step_bytes, uint8_t *d_magnitude_log, void *stream);

/* oics::fft::get_angle_with_fft(&TransformableMatrix, canny_threshold_1, canny_threshold_2,
 * min_line_length, max_line_gap, file_name, edge_image_output_dir) -> Result<f64> (fft.rs:145-256;
 * called core/src/main.rs:141-150, app/src-tauri/src/test.rs:450-459).  The debug picture stays with
 * the caller (as for omr_get_angle_with_hough).  Keeps the vote's quirk (fft.rs:231 re-reads line i). */
int omr_get_angle_with_fft(const omr_image *gray_u8c1, double canny_threshold_1, double canny_threshold_2,
                           double min_line_length, double max_line_gap, double *angle_out);

/* oics::omr::get_result_from_fourier_transform(&Mat, weak, strong, min_line_length, max_line_gap) ->
 * Result<OmrResult> (omr.rs:304-337) on the 3/4-channel scan. */
int omr_get_result_from_fourier_transform(const omr_image *src, double canny_threshold_weak,
                                          double canny_threshold_strong, double fourier_min_line_length,
                                          double fourier_max_line_gap, double *angle, int32_t *status,
                                          double *candidates, int32_t cand_cap, int32_t *cand_len);

/* calculate::get_arithmetic_mean / get_standard_deviation (calculate.rs:2-10, :13-23) */
int omr_get_arithmetic_mean(const double *v, size_t n, double *out);
int omr_get_standard_deviation(const double *v, size_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif
