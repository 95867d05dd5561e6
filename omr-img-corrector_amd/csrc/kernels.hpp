// kernels.hpp -- launch wrappers of the gfx950 kernels (kernels.hip). Host-callable, no torch types.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace omr {

struct int2_t {
    int32_t x, y;
};

// Sweep geometry shared by host and device.
struct SweepDims {
    int32_t rows, cols;  // image (and destination canvas) size
    int32_t A;           // candidate count
    int32_t wpr;         // 32-bit words per row of the bit image
};

// Per-candidate tiling parameters of the LDS-staged kernel, computed on the host from the
// inverse matrix (plan creation).
struct LdsTile {
    int32_t rows_per_tile;  // destination rows a wave covers per staged window (0: does not fit)
    int32_t win_words;      // window row pitch in 32-bit words
    int32_t win_rows;       // window height
};

// u8 image -> bit image: bit (x & 31) of word [y][x >> 5] is 1 iff px <= black_max.
// `scans` images img_stride bytes apart -> `scans` bit images, packed one after the other.
hipError_t launch_pack_bits(const uint8_t *d_img, int64_t step, int rows, int cols, int black_max,
                            uint32_t *d_bits, int wpr, hipStream_t s, int scans = 1, int64_t img_stride = 0);

// OpenCV hal::warpAffine fixed-point tables for A inverse matrices.
// adelta/bdelta: [A][cols]; xy0: [A][rows] (X0, Y0 incl. round_delta). *d_overflow != 0 when a
// table entry left the int32-safe range.
hipError_t launch_tables(const double *d_Minv, SweepDims d, int round_delta, int32_t *d_adelta,
                         int32_t *d_bdelta, int2_t *d_xy0, int32_t *d_overflow, hipStream_t s);

// Generic sweep: every (candidate, dst pixel) gathers its source bit from global memory.
// vproj [A][cols], hproj [A][rows] must be zero on entry (integer atomics).
hipError_t launch_sweep_generic(const uint32_t *d_bits, SweepDims d, const int32_t *d_adelta,
                                const int32_t *d_bdelta, const int2_t *d_xy0, const int32_t *d_list,
                                int n_list, uint32_t *d_vproj, uint32_t *d_hproj, hipStream_t s, int scans = 1);

// LDS-staged sweep (rotation-like matrices whose per-wave source window fits the LDS budget).
// d_list (may be NULL = all candidates): the candidates to sweep, n_list of them.
hipError_t launch_sweep_lds(const uint32_t *d_bits, SweepDims d, const int32_t *d_adelta,
                            const int32_t *d_bdelta, const int2_t *d_xy0, const LdsTile *d_tiles,
                            const int32_t *d_list, int n_list, uint32_t *d_vproj, uint32_t *d_hproj,
                            hipStream_t s, int scans = 1);  // scans: bit images rows x wpr apart, projections A x cols / A x rows apart

// ---- run-merging ("S") sweep ---------------------------------------------------------------
// For every listed candidate the kernel builds 32 destination pixels at a time,
//     bit(r, c) = src[(RT[r].y + CB[c]) >> 10][(RT[r].x + CA[c]) >> 10],   RT = (X0, Y0), CA = adelta, CB = bdelta,
// and counts them per destination row and per destination column.
#define OMR_RUN_TUPLES 40
#define OMR_RUN_K 4          // destination words per word group of the run-merging kernel
#define OMR_RUN_GC 16        // most word groups one workgroup walks (a "chunk"; row-count partials are per chunk)
#define OMR_RUN_MAX_ROWS 4608  // rows the kernel's LDS row counters hold; taller images are swept in row chunks
struct RunTab {                        // per (candidate, 32-column word); 3648 B, 16-B aligned
    // [id][level]: destination bits that read source row `level`; levels 0-3 and 4-7 are kept in two planes of
    // 16-byte entries: a wave's lanes use up to ~14 consecutive ids at once, and 32-byte entries put ids 8
    // apart on the same LDS banks (ds_read_b128 conflicts); at 16 bytes only ids 16 apart collide
    uint32_t tupYlo[OMR_RUN_TUPLES][4];
    uint32_t tupYhi[OMR_RUN_TUPLES][4];
    uint32_t tupX[OMR_RUN_TUPLES][2];  // [id][s-1]:   destination bits whose source column lags by s
    uint8_t idxY[1024];                // row-coordinate fraction -> tupY id
    uint8_t idxX[1024];                // bit-coordinate fraction -> tupX id
};
struct RunMeta {  // per (candidate, word)
    int32_t ca0;    // CA[c0]
    int32_t cb0;    // CB[c0] + (base_off << 10)
    int32_t nlev;   // source rows a word can touch (1..8)
    int32_t smax;   // largest column lag (0..2)
    uint32_t valid; // destination bits that exist (last word of a row)
    int32_t ok;     // 0: this word cannot be run-merged (candidate falls back to the gather kernel)
    int32_t pad0, pad1;
};
struct RunBlk {  // per (candidate, word group): what a sweep block needs before it can fetch
    int32_t nlev, smax;   // maxima over the group's words
    uint32_t valid_last;  // RunMeta::valid of the group's last word
    int32_t ca_min, ca_max;  // min / max of CA at the group's first and last column (CA is monotone)
    int32_t cb_min, cb_max;  // CB likewise
    int32_t lagbits;      // 2 bits per word pair: the pair's largest column lag
};
struct RunPass {
    const uint32_t *srcT;  // transposed bit images of the launch's scans, with their zero guard: [scan][NWt][rowsT]
    int32_t NWt, rowsT;    // word columns, rows per word column (a multiple of 4), guard included
    int32_t GX, GY;        // guard: word columns left of the image, rows above it
    const int2_t *wgeo;    // [A][G][bands][8 waves] window origins (word column, row); x = INT_MAX: does not fit
    const int2_t *RT;      // [A][NR] (X0, Y0)
    int32_t NR, NC;        // destination rows, columns
    int32_t NWp;           // words per candidate in tabs / metac: G * OMR_RUN_K (the last group is padded)
    const RunTab *tabs;    // [A][NWp]
    const int2_t *metac;   // [A][NWp] (ca0, cb0)
    const RunBlk *blk;     // [A][G]
    uint16_t *part;        // [scan][A][P][NRp] row counts per chunk of word groups
    int32_t G;             // word groups: ceil(ceil(NC / 32) / OMR_RUN_K)
    int32_t GC, P;         // word groups per chunk, chunks (P = ceil(G / GC))
    int32_t NRp;           // row pitch of part (even)
    int32_t RB, RCH;       // bands of 512 rows per row chunk (RB * 512 <= OMR_RUN_MAX_ROWS), row chunks
    int32_t scans;         // scans per launch (blockIdx.x)
    int32_t A;             // candidates of the plan (stride of part / vproj between scans)
};
// run tables of every (candidate, word): d_tabs [A][G * OMR_RUN_K], d_meta [A][NW], d_metac [A][G * OMR_RUN_K], d_blk [A][G]
hipError_t launch_runtab(const int32_t *d_CA, const int32_t *d_CB, int A, int NC, int NW, RunTab *d_tabs,
                         RunMeta *d_meta, int2_t *d_metac, RunBlk *d_blk, hipStream_t s);
// window origins of the plan; d_ext[4] = {min column, max column + 1, min row, max row + 1} over the windows that fit
// (the caller presets it to {INT_MAX, INT_MIN, INT_MAX, INT_MIN})
hipError_t launch_rungeo(const int2_t *d_RT, const RunBlk *d_blk, int A, int G, int NR, int2_t *d_wgeo, int32_t *d_ext,
                         hipStream_t s);
// bit images [scan][rows][wpr] -> transposed, inside their zero guard: [scan][NWt][rowsT], image at (GX, GY)
hipError_t launch_transpose_bits(const uint32_t *d_bits, int rows, int wpr, uint32_t *d_T, int NW, int NWt, int rowsT, int GX,
                                 int GY, hipStream_t s, int scans = 1);
// d_list: n_list candidate indices; d_guard[a] != 0 when a window did not fit for candidate a.
// Row counts go to p.part (u16 partials per chunk of word groups), column counts to d_vproj[scan][a][NC]
// (complete, plain stores).
hipError_t launch_runs(const RunPass &p, const int32_t *d_list, int n_list, int32_t *d_guard, uint32_t *d_vproj,
                       hipStream_t s);  // p.scans launches' worth of blocks in one grid
#ifdef OMR_RUNS_DEBUG
// development aid (make debug only): phase clocks summed by runs_kernel
hipError_t debug_runs_stamps(unsigned long long out[8], bool reset);
#endif
// hproj[a][r] = sum of the P partial row counts, for the listed (run-merged) candidates
hipError_t launch_fold_parts(const uint16_t *d_part, int P, int NR, int NRp, const int32_t *d_list, int n_list,
                             uint32_t *d_proj, hipStream_t s, int scans = 1, int A = 0);

// calculate.rs:13-23 on the integer projections: one block per (candidate, axis).
// `scans` result sets back to back: vproj [scans][A][cols], hproj [scans][A][rows], sd [scans][A].
// latency: one block per chain (a single scan that is waited for) instead of one lane per chain
// (batches, where the kernel runs beside the next sweep and must stay out of its way)
hipError_t launch_stddev(const uint32_t *d_vproj, const uint32_t *d_hproj, SweepDims d, double *d_v_sd,
                         double *d_h_sd, hipStream_t s, int scans = 1, bool latency = false);

// projection.rs:125-190 arg-max (lowest index on exact ties).
hipError_t launch_argmax_path1(const double *d_v_sd, const double *d_h_sd, int A, int32_t *d_best,
                               hipStream_t s, int scans = 1);

// ---- per-image helpers -------------------------------------------------------------------
hipError_t launch_threshold(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst,
                            int64_t dstep, int thresh, int maxval, hipStream_t s);
hipError_t launch_rgb2gray(const uint8_t *d_src, int64_t sstep, int rows, int cols, int cn,
                           uint8_t *d_dst, int64_t dstep, hipStream_t s);
hipError_t launch_erode_cross3(const uint8_t *d_src, int64_t sstep, int rows, int cols,
                               uint8_t *d_dst, int64_t dstep, hipStream_t s);
hipError_t launch_resize_area_int(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn,
                                  uint8_t *d_dst, int64_t dstep, int drows, int dcols, int kx, int ky,
                                  hipStream_t s);
struct AreaTap {
    int32_t si, di;
    float alpha;
};
hipError_t launch_resize_area_general(const uint8_t *d_src, int64_t sstep, int cn, uint8_t *d_dst,
                                      int64_t dstep, int drows, int dcols, const AreaTap *d_xtab,
                                      const int32_t *d_xofs, const AreaTap *d_ytab,
                                      const int32_t *d_yofs, hipStream_t s);
hipError_t launch_warp_nn(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn,
                          uint8_t *d_dst, int64_t dstep, int drows, int dcols, const double *d_Minv,
                          uint32_t border_rgba, hipStream_t s);
hipError_t launch_warp_linear(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn,
                              uint8_t *d_dst, int64_t dstep, int drows, int dcols,
                              const double *d_Minv, uint32_t border_rgba, hipStream_t s);

// ---- batched final deskew (deskew.hip): scan z is rotated by the candidate best[z]'s angle, CONTAIN geometry
struct DeskewPass {
    const uint8_t *src;       // scans, 1 channel: scan_stride bytes apart, sstep bytes per row
    int64_t scan_stride, sstep;
    int32_t srows, scols;
    uint8_t *dst;             // canvases: out_stride bytes apart, dstep bytes per row, each holds DR x DC pixels
    int64_t out_stride, dstep;
    const int32_t *best;      // [scans] winning candidate per scan (device)
    const int32_t *wsize;     // [A][2] canvas rows, cols per candidate
    const int32_t *adelta, *bdelta;  // [A][DC] warpAffine's column tables of the candidate's CONTAIN matrix
    const int2_t *xy0;        // [A][DR] its row tables, without the round delta
    int32_t DC, DR;           // largest canvas over the candidates (DC a multiple of 4)
    int32_t border;           // border value (0..255)
    int32_t *out_size;        // [scans][2] canvas rows, cols of every scan (device), or null
    int32_t order, ntx, nty;  // (set by launch_deskew_warp) the workgroup order and the tiles across / down the largest canvas
};
// d_tiles: deskew_tile_bytes(p, scans) bytes of scratch (the per-tile records made by the launch's first kernel)
size_t deskew_tile_bytes(const DeskewPass &p, int scans);
hipError_t launch_deskew_warp(const DeskewPass &p, int scans, int interp, void *d_tiles, hipStream_t s);

// ---- tuned single-channel stage kernels (stages.hip); each falls back to the generic form -------
hipError_t launch_rgb2gray_fast(const uint8_t *d_src, int64_t sstep, int rows, int cols, int cn, uint8_t *d_dst,
                                int64_t dstep, hipStream_t s);
// erode(3x3 cross) x 3 iterations fused (omr.rs:98-112)
hipError_t launch_erode3x_cross(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                                hipStream_t s);
hipError_t launch_resize_area_int_fast(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn,
                                       uint8_t *d_dst, int64_t dstep, int drows, int dcols, int kx, int ky,
                                       hipStream_t s);
// 1- / 3-channel warpAffine (NEAREST, LINEAR) with the source box of a 64x16 tile staged in LDS; Minv is a
// HOST pointer (passed by value to the kernel).  Returns hipErrorInvalidValue for other channel counts (use the
// generic kernel).
hipError_t launch_warp_fast(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                            int64_t dstep, int drows, int dcols, const double Minv[6], int interp, uint32_t border_rgba,
                            hipStream_t s);
// resize(INTER_LINEAR) (area_mode false) / INTER_AREA's bilinear emulation when an axis enlarges (true)
hipError_t launch_resize_linear(const uint8_t *d_src, int64_t sstep, int srows, int scols, int cn, uint8_t *d_dst,
                                int64_t dstep, int drows, int dcols, bool area_mode, hipStream_t s);
hipError_t launch_threshold_fast(const uint8_t *d_src, int64_t sstep, int rows, int cols, uint8_t *d_dst, int64_t dstep,
                                 int thresh, int maxval, hipStream_t s);

}  // namespace omr
