// fft.hpp -- device kernels of the FFT deskew path: the spectrum pictures of
// packages/lib/src/fft.rs:42-141 (2-D complex float32 DFT with DFT_SCALE, fft_shift, magnitude,
// min-max "correction", log, 8-bit conversion).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace omr {

struct cfloat {
    float x, y;
};

// One pass of 1-D transforms over `lines` lines of length n, a whole line per workgroup in LDS:
// a radix-4 / radix-2 Stockham FFT when n is a power of two (m = n), else Bluestein's chirp-z through two
// FFTs of length m = the power of two >= 2n - 1.  m <= 8192 ping-pongs between two LDS buffers (radix 8);
// m = 16384 (lines of 4097 .. 8192 points, e.g. a 600-dpi A4 scan) runs in place in one buffer (radix 2).
struct FftPass {
    const uint8_t *src_u8;  // real 8-bit input (im = 0), x = (float)v * in_scale; or NULL
    int64_t src_step;       // bytes between lines of src_u8
    float in_scale;
    const cfloat *src_c;    // complex input; used when src_u8 == NULL
    cfloat *dst;            // complex output
    int64_t line_stride;    // elements between consecutive lines of src_c and dst (n for packed rows, 1 for columns)
    int64_t elem_stride;    // elements between consecutive points of a line (1 for rows, the row pitch for columns)
    int32_t n, m, log2m, lines;
    const cfloat *W;        // per-stage twiddle tables of the radix-8 stages, concatenated: stage Ns holds
                            // exp(-2 pi i k / (8 Ns)), k < Ns, at offset (Ns - Ns0) / 7, Ns0 = 2^(log2m % 3)
    const cfloat *chirp;    // n: exp(-i pi k^2 / n); NULL for the direct transform
    const cfloat *Bf;       // m: FFT_m of the padded conjugate chirp (Bluestein)
    float out_scale;
    // where the complex output goes: point k of line l at dst[l * dst_line_stride + k * dst_elem_stride]
    // (0, 0 = the input's strides).  The row pass of the pictures writes TRANSPOSED (line stride 1): the column
    // pass then reads whole lines of consecutive addresses instead of one 8-byte element per cache line.
    int64_t dst_line_stride, dst_elem_stride;
    int32_t xcd_blocked;    // workgroups of one XCD take consecutive lines (their partial cache lines of a
                            // transposed / strided array then meet in that XCD's L2)
    // Column pass of the spectrum pictures: when mag_dst is set the pass writes |F| (float) instead of F,
    // already quadrant-swapped (fft_shift, fft.rs:67-86: an odd last row / column stays put), and the
    // per-workgroup extrema of |F| to part[2 * line .. + 1] -- no complex spectrum is ever written.
    float *mag_dst;         // TRANSPOSED |F| of the half spectrum: lines (= columns 0 .. C/2) of mag_pitch floats, |F(k, c)| at
                            // [c * mag_pitch + k], unshifted (the picture kernel shifts and mirrors), or NULL
    int32_t mag_pitch, img_rows, img_cols;
    float *part;
    // The input is REAL (an 8-bit scan), which halves both passes:
    // real_pairs (row pass, src_u8): a workgroup transforms rows 2b and 2b + 1 together as the real and the
    //   imaginary part of one complex line and separates the two spectra by Hermitian symmetry,
    //   A[k] = (Z[k] + conj Z[n-k]) / 2, B[k] = (Z[k] - conj Z[n-k]) / 2i; only columns 0 .. n/2 are written
    //   (the rest is their mirror image).  src_rows = number of rows (an odd last row goes alone).
    // half_mirror (column pass in picture mode): lines = img_cols / 2 + 1; |F(k, c)| is also stored at the
    //   mirrored point ((R - k) % R, (C - c) % C), F(-k, -c) = conj F(k, c), unless the column is its own mirror.
    int32_t real_pairs, src_rows, half_mirror;
    // several scans per launch (blockIdx.y = scan): strides of the per-scan arrays; 0 / 1 scan by default
    const cfloat *Wfull;    // in-place path (m > OMR_FFT_MAX_PINGPONG): m / 2 twiddles exp(-2 pi i t / m)
    int32_t scans;
    int64_t src_u8_scan_stride;                  // bytes
    int64_t c_scan_stride;                       // elements of src_c / dst
    int64_t mag_scan_stride, part_scan_stride;   // floats
    int32_t sub;            // fft_mixed.hip's Bluestein on sub-lines: their number (0: every other kind of pass)
};
#define OMR_FFT_MAX_M 16384    // transforms of 16384 points run in place (one 128 KiB LDS buffer, radix 2)
#define OMR_FFT_MAX_PINGPONG 8192
hipError_t launch_fft_pass(const FftPass &p, hipStream_t s);

// fft_mixed.hip: the same pass for the line lengths that have a mixed-radix kernel (an A4 scan's short side: 1240,
// 2480, 4960 = 2^a * 5 * 31): in-place Stockham stages in LDS instead of Bluestein's two 8192-point transforms.
// fft_mixed_radices(): the stages of length n, first stage first (returns their number, 0 = not a mixed-radix length);
// p.W must hold fft_mixed_table_size(n) twiddles laid out as fft_mixed.hip describes, p.m = p.n, p.chirp = NULL.
int fft_mixed_radices(int n, int radices[4]);
int fft_mixed_table_size(int n);
// ... and for lines of n = SUB * P points, 2048 < n <= 8192, whose plain chirp-z would run 8192 or 16384 points:
// Bluestein on the SUB interleaved sub-lines (M = 1792 or 2048 points each) and one radix-SUB stage.
// fft_bluesub_lines(): SUB (0: not taken); fft_bluesub_plan(): M and the stages of an M-point transform; p.sub = SUB,
// p.m = M, p.chirp / p.Bf = the chirp tables of P = n / SUB points for M-point transforms, p.W = the stages' twiddles
// (laid out like a mixed-radix plan of M points, fft_bluesub_stage_table_size(n) entries) followed by
// G[r][k] = chirp[k] exp(-2 pi i r k / n) / M, r < SUB, k < P.
int fft_bluesub_lines(int n);
int fft_bluesub_plan(int n, int *m, int radices[3]);
int fft_bluesub_stage_table_size(int n);
hipError_t launch_fft_mixed(const FftPass &p, hipStream_t s);

// fft_big.hip: lines whose chirp-z needs more than OMR_FFT_MAX_M points (n > 8192, not a power of two): Bluestein with
// M = 32768 or 65536 points through GLOBAL memory (four-step transforms, M = M1 x 256), `chunk` lines at a time in
// `buf`.  Two uses, the two passes of the pictures:
//   src_u8 set  : row pass -- pair line l = rows 2 l, 2 l + 1 of the 8-bit scan; the two spectra's points 0 .. n / 2
//                 go to dst[k * dst_pitch + row] (the transposed half spectrum), times out_scale
//   src_c set   : column pass -- line l = src_c + l * line_stride (n contiguous points); |F| * out_scale goes to
//                 mag_dst[l * mag_pitch + k], the line's extrema to part[2 l], part[2 l + 1]
struct BigLines {
    const uint8_t *src_u8;
    int64_t src_step;
    float in_scale;
    int32_t src_rows;
    const cfloat *src_c;
    int64_t line_stride;
    cfloat *dst;
    int64_t dst_pitch;
    float *mag_dst;
    int32_t mag_pitch;
    float *part;
    float out_scale, inv_m;  // inv_m = 1 / M
    int32_t n, M, lines, chunk;
    cfloat *buf;             // chunk * M points
    const cfloat *chirp;     // n: exp(-i pi k^2 / n)
    const cfloat *BfT;       // M: FFT_M of the padded conjugate chirp, point k1 + M1 k2 at [k1 * 256 + k2]
    const cfloat *wM;        // M: exp(-2 pi i t / M)
};
#define OMR_FFT_BIG_MAX_M 65536
hipError_t launch_big_lines(const BigLines &p, hipStream_t s);

// d_minmax: 4 ordered-uint words {min |F|, max |F|, min log, max log}; d_part: scratch for per-block
// extrema, spec_part_floats(rows, cols) floats
inline size_t spec_part_floats(int rows, int cols)
{
    const size_t a = (size_t)((cols + 1023) / 1024) * (size_t)rows, b = 8192;
    return 2 * (a > b ? a : b);
}
// fold n per-workgroup (min, max) pairs into d_minmax[0..1] (ordered-uint keys)
hipError_t launch_minmax_final(const float *d_part, int n, uint32_t *d_minmax, hipStream_t s, int scans = 1,
                               int64_t part_scan_stride = 0);  // scan z: d_part + z * stride -> d_minmax + 4 z
// Both 8-bit pictures (row-major, packed) from the transposed |F| of the half spectrum (cols / 2 + 1 lines of mag_pitch floats) in one pass: correction(|F|) * 255 -> "magnitude_image" (x 255 again,
// fft.rs:134) and log(. + 1/255) -> correction -> "magnitude_log_image" (fft.rs:113-119, :136-138).  The
// extrema of the log picture are the images of the extrema of |F| under the same float expressions (every
// step is monotone), so no second reduction pass and no float log array are needed.
hipError_t launch_spec_pictures(const float *d_mag, int rows, int cols, int mag_pitch, const uint32_t *d_minmax,
                                uint8_t *d_mag_u8, uint8_t *d_log_u8, hipStream_t s, int scans = 1,
                                int64_t mag_scan_stride = 0);  // pictures of scan z: + z * rows * cols; minmax + 4 z

}  // namespace omr
