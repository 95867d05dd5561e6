// fft_mixed.hip -- line transforms of the FFT deskew path (packages/lib/src/fft.rs:42-65: the reference transforms the
// scan at its own size) as IN-PLACE Stockham stages in LDS, for the lengths that matter most: an A4 scan's sides and
// BASELINE config 5's 4096.
//   * mixed radix (31, 5, 2^a) for the SHORT side of an A4 scan: 150 k dpi gives 1240 k pixels, 1240 = 2^3 * 5 * 31.
//     fft.hip's general answer for a length that is not a power of two is Bluestein's chirp-z -- two 8192-point
//     transforms for a 2480-point line, 86 butterfly-levels per point; here the same line is three stages of radix 31,
//     5 and 16 (about 13), the line never leaving LDS;
//   * chirp-z on interleaved sub-lines for the LONG side (1754 k = 2^b * 877, 877 prime) and every other 4 P / 8 P
//     (fft_bluesub_kernel below);
//   * 4096 points as three radix-16 stages.
//
// Its own translation unit and its own kernels on purpose: the code generated for fft.hip's fft_pass_kernel is
// sensitive to what else is compiled into it (an odd-radix stage that was merely present slowed every power-of-two
// transform by 15-25 %, fft.hip).  oics_fft.cpp hands a pass to launch_fft_mixed() when fft_mixed_radices() or
// fft_bluesub_lines() knows the length; every other length keeps fft.hip's paths.
//
// One workgroup transforms LINES lines at once so that every stage has at least one butterfly per thread (radix 31
// has only n / 31 per line).  A stage works IN PLACE: every thread reads the inputs of its butterflies into registers,
// the workgroup meets, the outputs go back to the Stockham positions -- half the LDS of a ping-pong pair, so two to
// four workgroups share a CU, and that occupancy is what these latency-bound passes are short of (the same 4096-point
// line: 0.135 ms per picture as four ping-pong radix-8 stages, 0.102 here).  The first stage reads the pass's input
// (two 8-bit rows as the real and the imaginary part, copied into LDS in whole rows first, or a complex line straight
// from global memory); an odd radix comes first (Ns = 1: no twiddles on the most expensive butterfly).  An odd
// butterfly uses the real symmetry of its matrix (x[k] +- x[R - k]): R = 31 costs 900 fused multiply-adds, its cosines
// and sines are instruction literals (fft_dft_tables.hpp).
//
// float32 throughout like fft.hip; this file contracts (fmaf) -- the pictures' tolerance test covers it
// (tests/test_gpu_fft.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft.hpp"
#include "fft_dft_tables.hpp"

namespace omr {
namespace {

__device__ __forceinline__ cfloat cadd(const cfloat a, const cfloat b) { return cfloat{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cfloat csub(const cfloat a, const cfloat b) { return cfloat{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cfloat cmul_mi(const cfloat a) { return cfloat{a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ cfloat cmul(const cfloat a, const cfloat b)
{
    return cfloat{__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)};
}

// ---- butterflies: x[0 .. R) in, X[0 .. R) out, natural order, in place
template <int R>
__device__ __forceinline__ void dft_odd(cfloat (&x)[R])
{
    constexpr int H = (R - 1) / 2;
    cfloat s[H], d[H];
#pragma unroll
    for (int k = 1; k <= H; k++) {
        s[k - 1] = cadd(x[k], x[R - k]);
        d[k - 1] = csub(x[k], x[R - k]);
    }
    const cfloat x0 = x[0];
    cfloat t = x0;
#pragma unroll
    for (int k = 0; k < H; k++) t = cadd(t, s[k]);
    x[0] = t;
    // X[m] = x0 + sum_k (s_k cos(2 pi m k / R) - i d_k sin(2 pi m k / R)), X[R - m] its mirror image (+ i d_k sin)
#pragma unroll
    for (int m = 1; m <= H; m++) {
        float ax = x0.x, ay = x0.y, bx = 0.f, by = 0.f;
#pragma unroll
        for (int k = 0; k < H; k++) {
            const float c = DftOdd<R>::C[m - 1][k], sn = DftOdd<R>::S[m - 1][k];
            ax = __builtin_fmaf(c, s[k].x, ax);
            ay = __builtin_fmaf(c, s[k].y, ay);
            bx = __builtin_fmaf(sn, d[k].x, bx);
            by = __builtin_fmaf(sn, d[k].y, by);
        }
        x[m] = cfloat{ax + by, ay - bx};
        x[R - m] = cfloat{ax - by, ay + bx};
    }
}
__device__ __forceinline__ void dft4(cfloat &a0, cfloat &a1, cfloat &a2, cfloat &a3)
{
    const cfloat t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_mi(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}
template <int R>
__device__ __forceinline__ void dft(cfloat (&x)[R])
{
    if constexpr (R == 2) {
        const cfloat a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
    } else if constexpr (R == 4) {
        dft4(x[0], x[1], x[2], x[3]);
    } else if constexpr (R == 8) {
        const float r2 = 0.70710678118654752440f;
        const cfloat a0 = cadd(x[0], x[4]), a1 = csub(x[0], x[4]);
        const cfloat a2 = cadd(x[2], x[6]), a3 = cmul_mi(csub(x[2], x[6]));
        const cfloat a4 = cadd(x[1], x[5]), a5 = csub(x[1], x[5]);
        const cfloat a6 = cadd(x[3], x[7]), a7 = cmul_mi(csub(x[3], x[7]));
        const cfloat b0 = cadd(a0, a2), b2 = csub(a0, a2);
        const cfloat b1 = cadd(a1, a3), b3 = csub(a1, a3);
        const cfloat b4 = cadd(a4, a6), b6 = cmul_mi(csub(a4, a6));
        const cfloat t5 = cadd(a5, a7), t7 = csub(a5, a7);
        const cfloat b5 = cfloat{(t5.x + t5.y) * r2, (t5.y - t5.x) * r2};    // t5 * exp(-i pi / 4)
        const cfloat b7 = cfloat{(-t7.x + t7.y) * r2, (-t7.y - t7.x) * r2};  // t7 * exp(-3 i pi / 4)
        x[0] = cadd(b0, b4);
        x[1] = cadd(b1, b5);
        x[2] = cadd(b2, b6);
        x[3] = cadd(b3, b7);
        x[4] = csub(b0, b4);
        x[5] = csub(b1, b5);
        x[6] = csub(b2, b6);
        x[7] = csub(b3, b7);
    } else if constexpr (R == 16) {  // 4 x 4: radix 4 over n1, W16^(n2 k1), radix 4 over n2
        const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
#pragma unroll
        for (int n2 = 0; n2 < 4; n2++) dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);  // -> y[k1][n2] at x[4 k1 + n2]
        auto rot = [](cfloat v, float c, float s) {  // v * (c - i s)
            return cfloat{__builtin_fmaf(v.x, c, v.y * s), __builtin_fmaf(v.y, c, -(v.x * s))};
        };
        x[5] = rot(x[5], c1, s1);
        x[6] = rot(x[6], r2, r2);
        x[7] = rot(x[7], s1, c1);
        x[9] = rot(x[9], r2, r2);
        x[10] = cmul_mi(x[10]);
        x[11] = rot(x[11], -r2, r2);
        x[13] = rot(x[13], s1, c1);
        x[14] = rot(x[14], -r2, r2);
        x[15] = rot(x[15], -c1, -s1);
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) dft4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);  // X[k1 + 4 k2] at x[4 k1 + k2]
        cfloat y[16];
#pragma unroll
        for (int q = 0; q < 16; q++) y[q] = x[4 * (q & 3) + (q >> 2)];
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = y[q];
    } else {
        dft_odd<R>(x);
    }
}

// twiddle table of a stage (radix R after Ns points are done): R = 16 keeps exp(-2 pi i k / (16 Ns)), k < Ns, and takes
// its powers by multiplication (a tree four deep); the other radices keep every power, [q - 1][k], q = 1 .. R - 1
__host__ __device__ constexpr int mixed_tw_size(int R, int Ns) { return Ns == 1 ? 0 : (R == 16 ? Ns : (R - 1) * Ns); }

// One Stockham stage on LINES lines of N points in place.  first_in(l, i): point i of line l of the transform's input
// (only the first stage reads it; SYNC0: that input lives in the LDS lines themselves, so the first stage needs the
// two barriers of every other stage).
// PAD: a line is skewed by one element every 32 (index i lives at i + i / 32, lines N + N / 32 apart).  Power-of-two
// stages need it: radix 16 with Ns = 1 stores to addresses 16 j + q -- a 128-byte lane stride, 32 lanes on one pair
// of banks; with the skew the 64 lanes cover every bank twice.  (An odd first radix spreads its stores by itself.)
template <bool PAD>
__host__ __device__ __forceinline__ constexpr int lpad(int i) { return PAD ? i + (i >> 5) : i; }
template <int NT, int LINES, int N, int R, int Ns, bool FIRST, bool SYNC0, bool PAD, class F>
__device__ __forceinline__ void mixed_stage(cfloat *lds, const cfloat *__restrict__ tw, const int tid, F first_in)
{
    constexpr int NB = N / R, B = LINES * NB, BPT = (B + NT - 1) / NT, PITCH = lpad<PAD>(N);
    constexpr bool SYNC = !FIRST || SYNC0;
    cfloat u[BPT][R];
    if (SYNC) __syncthreads();  // the previous stage's outputs are in place
#pragma unroll
    for (int b = 0; b < BPT; b++) {
        const int idx = tid + b * NT;
        if (B % NT == 0 || idx < B) {
            const int l = idx / NB, j = idx - l * NB;
#pragma unroll
            for (int q = 0; q < R; q++) u[b][q] = FIRST ? first_in(l, j + q * NB) : lds[l * PITCH + lpad<PAD>(j + q * NB)];
        }
    }
    if (SYNC) __syncthreads();  // every input of this stage is in registers
#pragma unroll
    for (int b = 0; b < BPT; b++) {
        const int idx = tid + b * NT;
        if (B % NT == 0 || idx < B) {
            const int l = idx / NB, j = idx - l * NB;
            const int k = Ns == 1 ? 0 : j % Ns;
            if constexpr (Ns > 1) {
                if constexpr (R == 16) {
                    const cfloat w1 = tw[k];
                    const cfloat w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3);
                    const cfloat w7 = cmul(w4, w3), w8 = cmul(w4, w4), w9 = cmul(w8, w1), w10 = cmul(w5, w5), w11 = cmul(w8, w3);
                    const cfloat w12 = cmul(w6, w6), w13 = cmul(w8, w5), w14 = cmul(w7, w7), w15 = cmul(w8, w7);
                    const cfloat w[16] = {cfloat{1.f, 0.f}, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15};
#pragma unroll
                    for (int q = 1; q < 16; q++) u[b][q] = cmul(u[b][q], w[q]);
                } else {
#pragma unroll
                    for (int q = 1; q < R; q++) u[b][q] = cmul(u[b][q], tw[(q - 1) * Ns + k]);
                }
            }
            dft<R>(u[b]);
            cfloat *o = lds + l * PITCH;
            const int o0 = (j - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; q++) o[lpad<PAD>(o0 + q * Ns)] = u[b][q];
        }
    }
}
template <int NT, int LINES, int N, int Ns, bool FIRST, bool SYNC0, bool PAD, int R, int... REST>
struct MixedStages {
    template <class F>
    static __device__ __forceinline__ void run(cfloat *lds, const cfloat *__restrict__ tw, const int tid, F first_in)
    {
        mixed_stage<NT, LINES, N, R, Ns, FIRST, SYNC0, PAD>(lds, tw, tid, first_in);
        if constexpr (sizeof...(REST) > 0)
            MixedStages<NT, LINES, N, Ns * R, false, false, PAD, REST...>::run(lds, tw + mixed_tw_size(R, Ns), tid, first_in);
    }
};
template <int... RS>
struct Product {
    static constexpr int value = (1 * ... * RS);
};

// ---- what the passes of fft.hpp's FftPass read and write (shared by the kernels below)
__device__ __forceinline__ void pass_scan_offsets(FftPass &p)  // scan of the launch
{
    const int64_t z = blockIdx.y;
    if (p.src_u8) p.src_u8 += z * p.src_u8_scan_stride;
    if (p.src_c) p.src_c += z * p.c_scan_stride;
    if (p.dst) p.dst += z * p.c_scan_stride;
    if (p.mag_dst) p.mag_dst += z * p.mag_scan_stride;
    if (p.part) p.part += z * p.part_scan_stride;
}
// workgroups of one XCD take consecutive groups of lines (fft.hip: the partial cache lines of the transposed array then
// meet in that XCD's L2)
__device__ __forceinline__ int pass_group(const FftPass &p)
{
    int grp = blockIdx.x;
    if (p.xcd_blocked) {
        const int per = gridDim.x / 8, body = per * 8;
        if ((int)blockIdx.x < body) grp = (int)(blockIdx.x % 8) * per + (int)blockIdx.x / 8;
    }
    return grp;
}
// point i of complex line `line` of the pass's input: PAIRS = rows 2 line and 2 line + 1 of the 8-bit scan as real and
// imaginary part (an odd last row goes alone), else the complex array
template <bool PAIRS>
__device__ __forceinline__ cfloat pass_source(const FftPass &p, const int line, const int total, const int i)
{
    cfloat v{0.f, 0.f};
    if (line < total) {
        if (PAIRS) {
            const int64_t r = 2 * (int64_t)line;
            v.x = (float)p.src_u8[r * p.src_step + i] * p.in_scale + 0.0f;
            if (r + 1 < p.src_rows) v.y = (float)p.src_u8[(r + 1) * p.src_step + i] * p.in_scale + 0.0f;
        } else {
            v = p.src_c[(int64_t)line * p.line_stride + (int64_t)i * p.elem_stride];
        }
    }
    return v;
}
// The spectra of LINES lines of n points, complete in LDS (line l at lds + l * pitch, natural order), go out: PAIRS =
// the row pass of the pictures (Z = FFT(a + i b) -> FFT(a), FFT(b) by Hermitian symmetry, columns 0 .. n / 2, fft.hip's
// emit_pair; lane order: the LINES pairs of one column index are neighbours -- with the transposed output their 16-byte
// pieces are too), else the column pass (|F(k, line)| at its own place -- the picture kernel shifts and mirrors -- and
// the workgroup's extrema, written once per line slot: minmax_final_kernel reads one pair per line).
template <int NT, int LINES, bool PAIRS, bool PAD = false>
__device__ __forceinline__ void pass_emit(const FftPass &p, const cfloat *lds, const int n, const int pitch, const int line0,
                                          const int total, const int tid, float *red)
{
    const int64_t dls = p.dst_line_stride ? p.dst_line_stride : p.line_stride;
    const int64_t des = p.dst_elem_stride ? p.dst_elem_stride : p.elem_stride;
    if (PAIRS) {
        const bool wide16 = dls == 1 && (des & 1) == 0 && (((uintptr_t)p.dst) & 15) == 0;
        for (int idx = tid; idx < LINES * (n / 2 + 1); idx += NT) {
            const int k = idx / LINES, l = idx - k * LINES;
            const int line = line0 + l;
            if (line >= total) continue;
            const int64_t r = 2 * (int64_t)line;
            const bool second = r + 1 < p.src_rows;
            const cfloat zk = lds[l * pitch + lpad<PAD>(k)], zn = lds[l * pitch + lpad<PAD>(k == 0 ? 0 : n - k)];
            const cfloat a = cfloat{(0.5f * (zk.x + zn.x)) * p.out_scale, (0.5f * (zk.y - zn.y)) * p.out_scale};
            const cfloat b = cfloat{(0.5f * (zk.y + zn.y)) * p.out_scale, (-0.5f * (zk.x - zn.x)) * p.out_scale};
            cfloat *d0 = p.dst + r * dls + (int64_t)k * des;
            if (wide16 && second) {
                *(float4 *)d0 = make_float4(a.x, a.y, b.x, b.y);
            } else {
                *d0 = a;
                if (second) d0[dls] = b;
            }
        }
    } else {
        float lo = __builtin_inff(), hi = -__builtin_inff();
        for (int idx = tid; idx < LINES * n; idx += NT) {
            const int l = idx / n, k = idx - l * n;
            const int line = line0 + l;
            if (line >= total) continue;
            const cfloat z = lds[l * pitch + lpad<PAD>(k)];
            const cfloat v = cfloat{z.x * p.out_scale, z.y * p.out_scale};
            if (p.mag_dst) {
                const float mg = sqrtf(v.x * v.x + v.y * v.y);
                p.mag_dst[(int64_t)line * p.mag_pitch + k] = mg;
                lo = fminf(lo, mg);
                hi = fmaxf(hi, mg);
            } else {
                p.dst[(int64_t)line * dls + (int64_t)k * des] = v;
            }
        }
        if (!p.mag_dst) return;
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_down(lo, off));
            hi = fmaxf(hi, __shfl_down(hi, off));
        }
        __syncthreads();  // every wave is done with the lines
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = lo;
            red[2 * (tid >> 6) + 1] = hi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (NT + 63) / 64; w++) {
                lo = fminf(lo, red[2 * w]);
                hi = fmaxf(hi, red[2 * w + 1]);
            }
            for (int l = 0; l < LINES && line0 + l < total; l++) {
                p.part[2 * (line0 + l)] = lo;
                p.part[2 * (line0 + l) + 1] = hi;
            }
        }
    }
}

// The pass on lines of N = prod(RS) points: PAIRS = the row pass of the pictures (two 8-bit rows per complex line, half
// spectrum out, transposed), else the column pass (complex lines in, |F| and its extrema out).
template <int NT, int LINES, bool PAIRS, int... RS>
__global__ __launch_bounds__(NT) void fft_mixed_kernel(FftPass p)
{
    constexpr int N = Product<RS...>::value;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    cfloat *lds = (cfloat *)lds_raw;
    pass_scan_offsets(p);
    const int tid = threadIdx.x;
    const int total = PAIRS ? (p.src_rows + 1) / 2 : p.lines;  // complex lines of the pass, LINES per workgroup
    const int line0 = pass_group(p) * LINES;
    if constexpr (PAIRS) {
        // The radix-31 stage wants pixels 80 apart: 62 single-byte loads per thread from global memory, each a round trip
        // a workgroup this small cannot hide.  The 2 LINES rows are copied into the front of the LDS lines first (8 bytes
        // per lane, whole rows) and the stage reads its bytes there; it writes only after the workgroup has met, like
        // every later stage.
        static_assert(N % 8 == 0 && 2 * N <= (int)sizeof(cfloat) * N, "rows are staged in 8-byte pieces inside their line");
        uint8_t *rows8 = (uint8_t *)lds_raw;  // row pair l: real row at l * 2 N, imaginary row at l * 2 N + N
        const bool aligned = ((((uintptr_t)p.src_u8) | (uint64_t)p.src_step) & 7) == 0;
        for (int idx = tid; idx < LINES * 2 * (N / 8); idx += NT) {
            const int row = idx / (N / 8), piece = idx - row * (N / 8);
            const int64_t r = 2 * (int64_t)line0 + row;
            uint2 v = make_uint2(0u, 0u);
            if (r < p.src_rows) {
                const uint8_t *g = p.src_u8 + r * p.src_step + 8 * piece;
                if (aligned) {
                    v = *(const uint2 *)g;
                } else {
                    v.x = g[0] | (g[1] << 8) | (g[2] << 16) | ((uint32_t)g[3] << 24);
                    v.y = g[4] | (g[5] << 8) | (g[6] << 16) | ((uint32_t)g[7] << 24);
                }
            }
            *(uint2 *)(rows8 + (size_t)row * N + 8 * piece) = v;
        }
        MixedStages<NT, LINES, N, 1, true, true, false, RS...>::run(lds, p.W, tid, [&](int l, int i) {
            const uint8_t *b = rows8 + (size_t)l * 2 * N + i;
            return cfloat{(float)b[0] * p.in_scale + 0.0f, (float)b[N] * p.in_scale + 0.0f};
        });
    } else {
        MixedStages<NT, LINES, N, 1, true, false, false, RS...>::run(lds, p.W, tid,
                                                               [&](int l, int i) { return pass_source<PAIRS>(p, line0 + l, total, i); });
    }
    __syncthreads();
    pass_emit<NT, LINES, PAIRS>(p, lds, N, N, line0, total, tid, (float *)lds_raw);
}

// ---- Bluestein on SUB interleaved sub-lines.  A line of n = SUB * P points whose P has no small factors (the LONG side
// of an A4 scan: 150 k dpi gives 1754 k pixels, 1754 = 2 * 877 and 877 is prime) used to pay two transforms of
// m >= 2 n - 1 points (8192 for n = 3508).  Decimated in time by SUB first, x_r[j] = x[SUB j + r], it pays SUB chirp-z
// transforms of P points with M >= 2 P - 1 points each -- M = 1792 = 7 * 16 * 16 when that holds (P <= 896: one eighth
// fewer points than 2048, and the odd first radix spreads its stores over the LDS banks by itself), else M = 2048 =
// 16 * 16 * 8.  Either way three stages instead of the four of 8192 points, the SUB sub-lines are
// the LINES of the in-place stages above (57 KB of LDS for SUB = 4, M = 1792: two workgroups per CU where the
// 8192-point ping-pong pair held the CU alone), and one radix-SUB stage with the twiddles W_n^(r k) puts the line
// together: X[k + P q] = sum_r W_SUB^(r q) W_n^(r k) X_r[k].
// Tables: p.chirp = exp(-i pi k^2 / P) (P), p.Bf = FFT_M of the padded conjugate chirp, p.W = the twiddles of the
// stages, then G[r][k] = chirp[k] W_n^(r k) / M (SUB x P: the last chirp product, the inverse transform's 1 / m and
// the combining twiddle in one factor).
template <int... RS>
struct StageTable;  // twiddle entries of a plan's stages
template <int R, int... REST>
struct StageTable<R, REST...> {
    static constexpr int size(int Ns = 1) { return mixed_tw_size(R, Ns) + StageTable<REST...>::size(Ns * R); }
};
template <>
struct StageTable<> {
    static constexpr int size(int = 1) { return 0; }
};
template <int NT, int SUB, bool PAIRS, bool PAD, int... RS>
__global__ __launch_bounds__(NT) void fft_bluesub_kernel(FftPass p)
{
    constexpr int M = Product<RS...>::value, MP = lpad<PAD>(M);  // sub-line r at lds + r * MP
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    cfloat *lds = (cfloat *)lds_raw;
    pass_scan_offsets(p);
    const int tid = threadIdx.x, n = p.n, P = n / SUB;
    const int total = PAIRS ? (p.src_rows + 1) / 2 : p.lines;
    const int line = pass_group(p);
    const cfloat *__restrict__ chirp = p.chirp;
    const cfloat *__restrict__ Bf = p.Bf;
    // A_r = FFT_M(x_r * chirp, zero beyond P)
    MixedStages<NT, SUB, M, 1, true, false, PAD, RS...>::run(lds, p.W, tid, [&](int r, int i) {
        cfloat v{0.f, 0.f};
        if (i < P) v = cmul(pass_source<PAIRS>(p, line, total, SUB * i + r), chirp[i]);
        return v;
    });
    // conj(IFFT_M(A_r Bf)) * M = FFT_M(conj(A_r Bf)): the product rides on the first stage's loads
    MixedStages<NT, SUB, M, 1, true, true, PAD, RS...>::run(lds, p.W, tid, [&](int r, int i) {
        const cfloat c = cmul(lds[r * MP + lpad<PAD>(i)], Bf[i]);
        return cfloat{c.x, -c.y};
    });
    // X_r[k] W_n^(r k) = conj(.) G[r][k], then the radix-SUB stage; the line ends up in natural order at lds[0 .. n)
    const cfloat *__restrict__ G = p.W + StageTable<RS...>::size();
    constexpr int KPT = (1024 + NT - 1) / NT;  // P <= 1024
    cfloat y[KPT][SUB];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < KPT; b++) {
        const int k = tid + b * NT;
        if (k < P) {
#pragma unroll
            for (int r = 0; r < SUB; r++) {
                const cfloat v = lds[r * MP + lpad<PAD>(k)];
                y[b][r] = cmul(cfloat{v.x, -v.y}, G[r * P + k]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < KPT; b++) {
        const int k = tid + b * NT;
        if (k < P) {
            dft<SUB>(y[b]);
#pragma unroll
            for (int q = 0; q < SUB; q++) lds[lpad<PAD>(k + P * q)] = y[b][q];
        }
    }
    __syncthreads();
    pass_emit<NT, 1, PAIRS, PAD>(p, lds, n, 0, line, total, tid, (float *)lds_raw);
}

template <int NT, int LINES, bool PAIRS, int... RS>
hipError_t launch_one(const FftPass &p, hipStream_t s)
{
    constexpr int N = Product<RS...>::value;
    const size_t lds = sizeof(cfloat) * (size_t)LINES * N;
    const int total = PAIRS ? (p.src_rows + 1) / 2 : p.lines;
    const dim3 grid((total + LINES - 1) / LINES, p.scans > 0 ? p.scans : 1);
    auto kern = fft_mixed_kernel<NT, LINES, PAIRS, RS...>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
    return hipGetLastError();
}

template <int NT, int SUB, bool PAIRS, bool PAD, int... RS>
hipError_t launch_sub(const FftPass &p, hipStream_t s)
{
    const size_t lds = sizeof(cfloat) * (size_t)SUB * lpad<PAD>(Product<RS...>::value);
    const int total = PAIRS ? (p.src_rows + 1) / 2 : p.lines;
    const dim3 grid(total, p.scans > 0 ? p.scans : 1);
    auto kern = fft_bluesub_kernel<NT, SUB, PAIRS, PAD, RS...>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
    return hipGetLastError();
}

}  // namespace

// The lengths this file transforms and their stages, first stage first.  Returns the number of stages (0: not ours).
int fft_mixed_radices(int n, int radices[4])
{
    switch (n) {
    case 1240: radices[0] = 31, radices[1] = 5, radices[2] = 8; return 3;
    case 2480: radices[0] = 31, radices[1] = 5, radices[2] = 16; return 3;
    case 4960: radices[0] = 31, radices[1] = 5, radices[2] = 4, radices[3] = 8; return 4;
    case 4096: radices[0] = 16, radices[1] = 16, radices[2] = 16; return 3;
    default: return 0;
    }
}

// The twiddle tables p.W must point to (the host builds them in double precision): the stages' tables one after the
// other, a stage of radix R after Ns points: R == 16 -> exp(-2 pi i k / (16 Ns)), k < Ns; else [q - 1][k] =
// exp(-2 pi i q k / (R Ns)), q = 1 .. R - 1, k < Ns; nothing for the first stage.
int fft_mixed_table_size(int n)
{
    int r[4];
    const int ns = fft_mixed_radices(n, r);
    int size = 0, Ns = 1;
    for (int i = 0; i < ns; i++) {
        size += mixed_tw_size(r[i], Ns);
        Ns *= r[i];
    }
    return size;
}

// Bluestein on sub-lines: the number of sub-lines for a line of n points (0: not ours -- fft.hip's plain chirp-z).  Taken
// where it halves the points per transform against the plain path's m: 2048 < n <= 4096 in four sub-lines, 4096 < n <=
// 8192 in eight (where the plain path runs 16384 points in place, radix 2).
int fft_bluesub_lines(int n)
{
    int r[4];
    if (fft_mixed_radices(n, r) > 0 || (n & (n - 1)) == 0) return 0;
    if (n > 2048 && n <= 4096 && n % 4 == 0) return 4;
    if (n > 4096 && n <= 8192 && n % 8 == 0) return 8;
    return 0;
}
// points per sub-line transform and its stages, first stage first (returns their number)
int fft_bluesub_plan(int n, int *m, int radices[3])
{
    const int sub = fft_bluesub_lines(n);
    if (!sub) return 0;
    if (2 * (n / sub) - 1 <= 1792) {
        *m = 1792, radices[0] = 7, radices[1] = 16, radices[2] = 16;
    } else {
        *m = 2048, radices[0] = 16, radices[1] = 16, radices[2] = 8;
    }
    return 3;
}
int fft_bluesub_stage_table_size(int n)
{
    int m, r[3];
    if (!fft_bluesub_plan(n, &m, r)) return 0;
    return m == 1792 ? StageTable<7, 16, 16>::size() : StageTable<16, 16, 8>::size();
}

hipError_t launch_fft_mixed(const FftPass &p, hipStream_t s)
{
    if (p.lines <= 0) return hipSuccess;
    const bool rowpass = p.real_pairs != 0 && p.src_u8 != nullptr && p.dst != nullptr;
    const bool colpass = p.real_pairs == 0 && p.src_c != nullptr && (p.mag_dst != nullptr || p.dst != nullptr);
    if (!rowpass && !colpass) return hipErrorInvalidValue;
    if (p.sub) {
        int m, r[3];
        if (p.sub != fft_bluesub_lines(p.n) || !fft_bluesub_plan(p.n, &m, r) || !p.chirp || !p.Bf || !p.W || p.m != m)
            return hipErrorInvalidValue;
        if (m == 1792) {
            if (p.sub == 4) return rowpass ? launch_sub<512, 4, true, false, 7, 16, 16>(p, s) : launch_sub<512, 4, false, false, 7, 16, 16>(p, s);
            return rowpass ? launch_sub<1024, 8, true, false, 7, 16, 16>(p, s) : launch_sub<1024, 8, false, false, 7, 16, 16>(p, s);
        }
        // (2048 points, measured on 3508-point lines: with skewed lines the bank conflicts go from 2.0e8 to 1.1e7 cycles per
        // launch, but the index arithmetic costs 50 VGPRs -- one workgroup per CU instead of two, the 1024-thread form
        // spills -- and the pass takes the same 0.57 ms per 8 scans: unskewed)
        if (p.sub == 4) return rowpass ? launch_sub<512, 4, true, false, 16, 16, 8>(p, s) : launch_sub<512, 4, false, false, 16, 16, 8>(p, s);
        return rowpass ? launch_sub<1024, 8, true, false, 16, 16, 8>(p, s) : launch_sub<1024, 8, false, false, 16, 16, 8>(p, s);
    }
    if (p.chirp || !p.W) return hipErrorInvalidValue;
    switch (p.n) {
    case 1240: return rowpass ? launch_one<192, 4, true, 31, 5, 8>(p, s) : launch_one<192, 4, false, 31, 5, 8>(p, s);
    // (threads, lines per workgroup) measured side by side on one GPU at 2480: (256, 3) 0.1257, (192, 2) 0.1229,
    // (128, 1) 0.1250, (96, 1) 0.1274 ms per A4 scan -- 160 radix-31 butterflies on three waves, four workgroups per CU
    case 2480: return rowpass ? launch_one<192, 2, true, 31, 5, 16>(p, s) : launch_one<192, 2, false, 31, 5, 16>(p, s);
    case 4960: return rowpass ? launch_one<192, 1, true, 31, 5, 4, 8>(p, s) : launch_one<192, 1, false, 31, 5, 4, 8>(p, s);
    // 4096 points (BASELINE config 5's side) as three in-place radix-16 stages, two lines per 512-thread workgroup, two
    // workgroups per CU: 0.102 ms per 4096 x 4096 picture against 0.135 for fft.hip's four ping-pong radix-8 stages
    // ((256 threads, 1 line): 0.106), measured side by side
    case 4096: return rowpass ? launch_one<512, 2, true, 16, 16, 16>(p, s) : launch_one<512, 2, false, 16, 16, 16>(p, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace omr
