// hough.hpp -- device kernels of the Hough-line deskew path:
// Canny(50, 150, 3) -> HoughLinesP(rho 1, theta pi/180, threshold, minLineLength, maxLineGap) as
// OpenCV 4.6.0 defines them (call sites packages/lib/src/hough.rs:27-43, omr.rs:236-253).
// All kernels take a batch: blockIdx.z (or .x for the per-scan kernels) = scan.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace omr {

// map values of the Canny stages (canny.cpp): 0 = may become an edge, 1 = no edge, 2 = edge
// Non-maximum suppression of the L1 Sobel gradient; cn = 1, 3 or 4 (per pixel the channel with the
// largest |dx| + |dy|).  src: n scans, scan_stride bytes apart.  map: n x rows x cols, packed.
hipError_t launch_canny_nms(const uint8_t *d_src, int64_t scan_stride, int64_t row_step, int rows, int cols, int cn,
                            int n, int low, int high, uint8_t *d_map, hipStream_t s);
// One hysteresis pass (tile-local fixed point); *d_changed |= 1 when any tile changed.
hipError_t launch_canny_hysteresis(uint8_t *d_map, int rows, int cols, int n, int *d_changed, hipStream_t s);
// map -> edges (0 / 255, in place) and the number of edge pixels of every row.
hipError_t launch_edges_rowcount(uint8_t *d_map, int rows, int cols, int n, int from_map, int32_t *d_rowcnt,
                                 hipStream_t s);
// exclusive scan of the row counts of every scan; d_total[scan] = number of edge pixels
hipError_t launch_edges_rowscan(const int32_t *d_rowcnt, int rows, int n, int32_t *d_rowoff, int32_t *d_total,
                                hipStream_t s);
// raster-order list of the non-zero pixels: nz[scan_off[scan] + k] = y << 16 | x, and the mask of the
// Hough stage, one bit per pixel in 8x8-pixel tiles of one 64-bit word each (a line walk of 128 steps
// touches a few dozen words; an A4 scan's mask is 1.1 MB): bit (y & 7) * 8 + (x & 7) of word
// (y >> 3) * tiles_x + (x >> 3).  d_mask_tiled must be zeroed before the call.
hipError_t launch_edges_compact(const uint8_t *d_edges, int rows, int cols, int n, const int32_t *d_rowoff,
                                const int64_t *d_scan_off, uint32_t *d_nz, uint8_t *d_mask_tiled, hipStream_t s);
__host__ __device__ inline int ppht_tiles_x(int cols) { return (cols + 7) / 8; }
__host__ __device__ inline int64_t ppht_mask_bytes(int rows, int cols) { return (int64_t)((rows + 7) / 8) * ppht_tiles_x(cols) * 8; }

struct PphtWalk {  // per accumulator angle: the line walk of hough.cpp (16.16 fixed point)
    int32_t xflag, dx0, dy0, pad;
};
struct PphtArgs {
    uint8_t *mask;            // n x ppht_mask_bytes(): 64-bit words of 8x8 pixels, bit set = point still available
    int32_t width, height;
    uint32_t *nz;             // point lists (destroyed)
    uint32_t *order;          // same size and offsets as nz: the points in the order they are drawn (written by the kernel)
    const int64_t *scan_off;  // [n] offset of a scan's list in nz / order
    const int32_t *count;     // [n] points per scan
    void *accum;              // n x accum_stride bins; row k of a scan starts at row_base[k] - (its lowest rho).
                              // acc_u16 = 0: int32 bins, zeroed.  acc_u16 = 1: uint16 bins holding count + 0x8080 (fill
                              // the buffer with the byte 0x80), accum_stride even; for rows + cols <= OMR_PPHT_U16_MAX_EXTENT
    int64_t accum_stride;
    int32_t acc_u16;
    const int32_t *row_base;  // numangle: offset of the bin rho = 0 of row k (rows hold only the reachable rho range)
    int32_t numangle;
    const float *ttab;        // numangle x (cos / rho, sin / rho) as float
    const PphtWalk *walk;     // numangle
    int32_t threshold, line_length, line_gap;
    int32_t *lines;           // n x cap x 4
    int32_t cap;
    int32_t *n_lines;         // [n]
    int32_t n_scans;
    int32_t *queue;           // one zeroed counter: scans handed out beyond the first grid-full
};
#define OMR_PPHT_THREADS 320      // wave 0 serves the points, wave 1 draws them, waves 2-4 help with the un-votes
#define OMR_PPHT_MAX_ANGLES 256   // up to four accumulator angles per lane
// rows + cols up to which the accumulator is kept in 16-bit bins (hough.hip).  0 = never: measured side by side on one
// GPU (profiles/r03_hough.md) the 16-bit accumulator LOST to int32 bins, 587 against 650 scans/s at 256 A4 scans -- the
// batch is bound by the latency of each scan's dependent chain, not by the footprint, and 16-bit stores are partial
// writes of a 32-bit word.  -DOMR_PPHT_U16_MAX_EXTENT=16000 rebuilds the 16-bit library of that comparison (a bin's
// count stays within +-(image diagonal), far inside the 16-bit range for rows + cols <= 16000).
#ifndef OMR_PPHT_U16_MAX_EXTENT
#define OMR_PPHT_U16_MAX_EXTENT 0
#endif
// in_flight: workgroups launched = scans worked on at once (<= 0 or >= n_scans: one workgroup per scan)
hipError_t launch_ppht(const PphtArgs &a, int in_flight, hipStream_t s);
#ifdef OMR_RUNS_DEBUG
hipError_t debug_ppht_stamps(unsigned long long out[12], bool reset);
#endif

// counts[i] = #{ j : |a[i] - a[j]| < 0.1 } in f32 (hough.rs:77-83) or in f64 on widened values
// (omr.rs:278-284)
hipError_t launch_angle_votes(const float *d_angles, int n, int as_f64, int32_t *d_counts, hipStream_t s);
// the same for a batch: scan k owns angles / counts [off[k], off[k + 1]); max_n = the longest list
hipError_t launch_angle_votes_batch(const float *d_angles, const int64_t *d_off, int n_scans, int max_n, int as_f64,
                                    int32_t *d_counts, hipStream_t s);

}  // namespace omr
