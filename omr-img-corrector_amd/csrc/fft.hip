// fft.hip -- 2-D complex float32 DFT and the spectrum pictures of the reference's FFT deskew path
// (packages/lib/src/fft.rs:42-141) for gfx950.
//
// The reference transforms the scan at its own size (no padding to a fast size), so lengths like
// 2480 = 2^4 * 5 * 31 or 3508 = 2^2 * 877 must work: a line is transformed entirely inside LDS, by a
// radix-8 Stockham FFT when its length is a power of two and by Bluestein's chirp-z (two power-of-two
// FFTs of length m >= 2n - 1 and three pointwise products) otherwise.  Twiddles and chirps are
// tabulated by the host in double precision.  The 2-D transform is a pass along the rows that stores the
// half spectrum TRANSPOSED (the input is real: two rows per workgroup, columns 0 .. C/2), then a pass along
// the lines of that array (= the picture's columns) that writes |F|, transposed too, and a picture kernel
// that turns it back through LDS tiles: every global access of the two passes runs along consecutive
// addresses (or 16-byte pieces that meet in one XCD's L2).  float32 throughout (the reference's dft is
// CV_32F), built without FMA contraction like the rest of the library (contracting the butterflies was
// measured: +1 %, not worth a second arithmetic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft.hpp"

namespace omr {

__device__ __forceinline__ cfloat cmul(const cfloat a, const cfloat b)
{
    return cfloat{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

#define FFT_THREADS NT  // threads of the workgroup: template parameter of the kernel and its device functions
// LDS lines are skewed by one element every 32: the Stockham stores of the first stages go to addresses
// 8 j + q (stride 64 bytes between lanes: a 16-way bank conflict on ds_write_b64 without the skew).
#define FPAD(i) ((i) + ((i) >> 5))
#define FFT_LDS_ELEMS(m) ((m) + ((m) >> 5))

// forward FFT of length m = 2^log2m, Stockham autosort.  Stages are radix 8 (three radix-2 levels in
// registers per LDS round trip) with one leading radix-2 or radix-4 stage when log2m is not a multiple of
// three; ping-pong between `in` and `out`, returns the buffer that holds the result (natural order).
// W: m twiddles exp(-2 pi i t / m).
__device__ __forceinline__ cfloat cadd(const cfloat a, const cfloat b) { return cfloat{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cfloat csub(const cfloat a, const cfloat b) { return cfloat{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cfloat cmul_mi(const cfloat a) { return cfloat{a.y, -a.x}; }  // a * (-i)

// 16-point DFT in registers: 4 x 4 decimation (radix-4 over n1, twiddles W16^(n2 k1), radix-4 over n2); x[4 n1 + n2]
// in, X[k1 + 4 k2] out, in place.
__device__ __forceinline__ void fft_radix4(cfloat &a0, cfloat &a1, cfloat &a2, cfloat &a3)
{
    const cfloat t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_mi(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}
__device__ __forceinline__ void fft_dft16(cfloat (&x)[16])
{
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) fft_radix4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);  // -> y[k1][n2] at x[4 k1 + n2]
    // y[k1][n2] *= W16^(n2 k1), W16 = exp(-2 pi i / 16)
    auto rot = [](cfloat v, float c, float s) { return cfloat{v.x * c + v.y * s, v.y * c - v.x * s}; };  // v * (c - i s)
    x[5] = rot(x[5], c1, s1);     // k1 = 1, n2 = 1: W^1
    x[6] = rot(x[6], r2, r2);     // n2 = 2: W^2
    x[7] = rot(x[7], s1, c1);     // n2 = 3: W^3
    x[9] = rot(x[9], r2, r2);     // k1 = 2, n2 = 1: W^2
    x[10] = cmul_mi(x[10]);       // W^4 = -i
    x[11] = rot(x[11], -r2, r2);  // W^6
    x[13] = rot(x[13], s1, c1);   // k1 = 3, n2 = 1: W^3
    x[14] = rot(x[14], -r2, r2);  // W^6
    x[15] = rot(x[15], -c1, -s1); // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fft_radix4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);  // -> X[k1 + 4 k2] at x[4 k1 + k2]
}

// first_in(i): point i of the transform's input.  The first stage reads its input through it, so whatever precedes
// a transform rides on that stage's loads instead of a pass of its own over an LDS buffer: the scan's pixels or the
// half spectrum's points from global memory (times the chirp, zero beyond n), or Bluestein's product with the chirp's
// spectrum and the conjugation of the inverse transform, conj(in[i] * Bf[i]).  The later stages read `in`.
// keep: only points 0 .. keep - 1 of the result are read by the caller (Bluestein's second transform: n of m); the
// radix-16 tail does not store the others.
template <int NT, class F>
__device__ cfloat *fft_forward_lds(cfloat *in, cfloat *out, const int m, const int log2m,
                                   const cfloat *__restrict__ Wst, const int tid, F first_in, const int keep = 1 << 30)
{
    constexpr bool PRE = true;
    int s = 0;
    // log2m = 3 a + 1 (8192, 1024, 128 points): a - 1 radix-8 stages and ONE radix-16 stage at the end instead of a
    // leading radix-2 stage and a radix-8 stages -- one pass over LDS and one barrier less for the same arithmetic
    const bool tail16 = log2m % 3 == 1 && log2m >= 7;
    const int lead = tail16 ? 0 : log2m % 3;
    const int Ns0 = 1 << lead;  // Ns of the first radix-8 stage
    const int log2m8 = tail16 ? log2m - 4 : log2m;  // bits handled by the leading and the radix-8 stages
    if (lead == 1) {  // radix 2, Ns = 1: no twiddles
        const int half = m >> 1;
        __syncthreads();
        for (int j = tid; j < half; j += FFT_THREADS) {
            const cfloat u0 = first_in(j), u1 = first_in(j + half);
            out[FPAD(2 * j)] = cadd(u0, u1);
            out[FPAD(2 * j + 1)] = csub(u0, u1);
        }
        cfloat *t = in;
        in = out;
        out = t;
        s = 1;
    } else if (lead == 2) {  // radix 4, Ns = 1: no twiddles
        const int quarter = m >> 2;
        __syncthreads();
        for (int j = tid; j < quarter; j += FFT_THREADS) {
            const cfloat u0 = first_in(j), u1 = first_in(j + quarter), u2 = first_in(j + 2 * quarter), u3 = first_in(j + 3 * quarter);
            const cfloat a = cadd(u0, u2), b = csub(u0, u2), c = cadd(u1, u3), d = cmul_mi(csub(u1, u3));
            out[FPAD(4 * j)] = cadd(a, c);
            out[FPAD(4 * j + 1)] = cadd(b, d);
            out[FPAD(4 * j + 2)] = csub(a, c);
            out[FPAD(4 * j + 3)] = csub(b, d);
        }
        cfloat *t = in;
        in = out;
        out = t;
        s = 2;
    }
    const int eighth = m >> 3;
    const float r2 = 0.70710678118654752440f;
    for (; s < log2m8; s += 3) {
        const int Ns = 1 << s;
        // one twiddle load per butterfly, from the stage's own contiguous table (lanes read neighbouring entries),
        // its powers by complex multiplication: the seven scattered loads W[q * tw] of the plain form touched up to
        // 64 cache lines per wave-instruction.  Requested BEFORE the barrier: its latency passes while the wave waits.
        const cfloat w1 = Wst[(Ns - Ns0) / 7 + (min(tid, eighth - 1) & (Ns - 1))];  // exp(-2 pi i k / (8 Ns))
        __syncthreads();
        // (launch_fft_pass gives a ping-pong transform at least m / 8 threads: one butterfly per thread and stage)
        if (const int j = tid; j < eighth) {
            const int k = j & (Ns - 1);
            cfloat u[8];
            if (PRE && s == 0) {  // (no leading radix-2 / radix-4 stage: this is the transform's first stage)
#pragma unroll
                for (int q = 0; q < 8; q++) u[q] = first_in(j + q * eighth);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) u[q] = in[FPAD(j + q * eighth)];
            }
            if (Ns > 1) {  // (Ns = 1: every twiddle is 1 -- 13 complex products less in a transform's first stage)
                const cfloat w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
                const cfloat w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
                u[1] = cmul(u[1], w1);
                u[2] = cmul(u[2], w2);
                u[3] = cmul(u[3], w3);
                u[4] = cmul(u[4], w4);
                u[5] = cmul(u[5], w5);
                u[6] = cmul(u[6], w6);
                u[7] = cmul(u[7], w7);
            }
            // radix-8 butterfly: y[q] = sum_p u[p] exp(-2 pi i p q / 8)
            const cfloat a0 = cadd(u[0], u[4]), a1 = csub(u[0], u[4]);
            const cfloat a2 = cadd(u[2], u[6]), a3 = cmul_mi(csub(u[2], u[6]));
            const cfloat a4 = cadd(u[1], u[5]), a5 = csub(u[1], u[5]);
            const cfloat a6 = cadd(u[3], u[7]), a7 = cmul_mi(csub(u[3], u[7]));
            const cfloat b0 = cadd(a0, a2), b2 = csub(a0, a2);  // even outputs of the first half
            const cfloat b1 = cadd(a1, a3), b3 = csub(a1, a3);
            const cfloat b4 = cadd(a4, a6), b6 = cmul_mi(csub(a4, a6));
            const cfloat t5 = cadd(a5, a7), t7 = csub(a5, a7);
            const cfloat b5 = cfloat{(t5.x + t5.y) * r2, (t5.y - t5.x) * r2};    // t5 * exp(-i pi / 4)
            const cfloat b7 = cfloat{(-t7.x + t7.y) * r2, (-t7.y - t7.x) * r2};  // t7 * exp(-3 i pi / 4)
            const int j0 = ((j - k) << 3) + k;
            out[FPAD(j0)] = cadd(b0, b4);
            out[FPAD(j0 + Ns)] = cadd(b1, b5);
            out[FPAD(j0 + 2 * Ns)] = cadd(b2, b6);
            out[FPAD(j0 + 3 * Ns)] = cadd(b3, b7);
            out[FPAD(j0 + 4 * Ns)] = csub(b0, b4);
            out[FPAD(j0 + 5 * Ns)] = csub(b1, b5);
            out[FPAD(j0 + 6 * Ns)] = csub(b2, b6);
            out[FPAD(j0 + 7 * Ns)] = csub(b3, b7);
        }
        cfloat *t = in;
        in = out;
        out = t;
    }
    if (tail16) {  // the last stage, radix 16: Ns = m / 16, so k = j; its twiddles exp(-2 pi i j / m) follow the radix-8 tables
        const int Ns = 1 << s, sixteenth = m >> 4;
        __syncthreads();
        if (const int j = tid; j < sixteenth) {
            const cfloat w1 = Wst[(Ns - 1) / 7 + j];
            const cfloat w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3);
            const cfloat w7 = cmul(w4, w3), w8 = cmul(w4, w4), w9 = cmul(w8, w1), w10 = cmul(w5, w5), w11 = cmul(w8, w3);
            const cfloat w12 = cmul(w6, w6), w13 = cmul(w8, w5), w14 = cmul(w7, w7), w15 = cmul(w8, w7);
            const cfloat w[16] = {cfloat{1.f, 0.f}, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15};
            cfloat u[16];
#pragma unroll
            for (int q = 0; q < 16; q++) u[q] = in[FPAD(j + q * sixteenth)];
#pragma unroll
            for (int q = 1; q < 16; q++) u[q] = cmul(u[q], w[q]);
            fft_dft16(u);
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (q * sixteenth < keep) out[FPAD(j + q * sixteenth)] = u[4 * (q & 3) + (q >> 2)];  // X[k1 + 4 k2] sits at [4 k1 + k2]
        }
        cfloat *t = in;
        in = out;
        out = t;
    }
    __syncthreads();
    return in;
}

// In-place forward FFT for lengths whose two ping-pong buffers would not fit LDS (m = 16384): radix-2
// decimation in time on BIT-REVERSED input, natural-order output.  Wf: m / 2 twiddles exp(-2 pi i t / m).
template <int NT>
__device__ void fft_inplace_lds(cfloat *buf, const int m, const int log2m, const cfloat *__restrict__ Wf, const int tid)
{
    for (int s = 0; s < log2m; s++) {
        const int half = 1 << s;
        __syncthreads();
        for (int j = tid; j < (m >> 1); j += FFT_THREADS) {
            const int pos = j & (half - 1);
            const int i0 = ((j - pos) << 1) + pos, i1 = i0 + half;
            const cfloat w = Wf[pos << (log2m - 1 - s)];
            const cfloat a = buf[FPAD(i0)], t = cmul(buf[FPAD(i1)], w);
            buf[FPAD(i0)] = cadd(a, t);
            buf[FPAD(i1)] = csub(a, t);
        }
    }
    __syncthreads();
}

// MODE: what is folded at compile time.  The code generated for this kernel is sensitive to what else is compiled into
// it (an odd-radix stage that was merely present once slowed every transform by 15-25 %), so the variants were
// measured side by side on one GPU: 1 = power-of-two row pass of the pictures (8-bit real rows in pairs -> transposed
// half spectrum), 0 = power-of-two column pass (complex lines -> |F|), 8 = everything else, flags at run time.
template <int NT, int MODE>
__global__ __launch_bounds__(NT) void fft_pass_kernel(FftPass p)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    {  // scan of the launch
        const int64_t z = blockIdx.y;
        if (p.src_u8) p.src_u8 += z * p.src_u8_scan_stride;
        if (p.src_c) p.src_c += z * p.c_scan_stride;
        if (p.dst) p.dst += z * p.c_scan_stride;
        if (p.mag_dst) p.mag_dst += z * p.mag_scan_stride;
        if (p.part) p.part += z * p.part_scan_stride;
    }
    cfloat *A = (cfloat *)lds_raw, *B = A + FFT_LDS_ELEMS(p.m);  // B is not allocated (nor used) on the in-place path
    const int tid = threadIdx.x, n = p.n, m = p.m;
    // MODE 8: every flag read at run time (the Bluestein and in-place paths measured 5 % FASTER that way -- the
    // compiler's choices, not ours); MODE 0 / 1: the power-of-two column / row pass with its flags folded
    const bool inplace = MODE == 8 ? m > OMR_FFT_MAX_PINGPONG : false;
    const int rshift = 32 - p.log2m;
    // Column passes (elem_stride > 1) touch 8 bytes per 64-byte sector: the eight columns that share a
    // sector must meet in one XCD's L2.  Workgroups go round-robin to the 8 XCDs, so XCD x takes the
    // lines [x * lines / 8, (x + 1) * lines / 8) in order instead of every eighth line.
    int64_t line = blockIdx.x;
    if (p.xcd_blocked) {
        const int per = gridDim.x / 8, body = per * 8;
        if ((int)blockIdx.x < body) line = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
    }
    const int64_t dls = p.dst_line_stride ? p.dst_line_stride : p.line_stride;
    const int64_t des = p.dst_elem_stride ? p.dst_elem_stride : p.elem_stride;
    const bool blue = MODE == 8 ? p.chirp != nullptr : false;
    const bool pairs = MODE == 8 ? p.real_pairs != 0 : (MODE & 1) != 0;
    const bool second = pairs && 2 * line + 1 < p.src_rows;
    if (pairs) line *= 2;
    auto source = [&](int k) -> cfloat {  // point k of the line, as the first transform takes it
        cfloat v{0.f, 0.f};
        if (k < n) {
            if (pairs) {
                v.x = (float)p.src_u8[line * p.src_step + k] * p.in_scale + 0.0f;
                if (second) v.y = (float)p.src_u8[(line + 1) * p.src_step + k] * p.in_scale + 0.0f;
            } else {
                v = p.src_c[line * p.line_stride + (int64_t)k * p.elem_stride];
            }
            if (blue) v = cmul(v, p.chirp[k]);
        }
        return v;
    };
    cfloat *P = A;
    if (inplace) {
        for (int k = tid; k < m; k += FFT_THREADS) A[FPAD((int)(__brev((unsigned)k) >> rshift))] = source(k);
        fft_inplace_lds<NT>(A, m, p.log2m, p.Wfull, tid);
    } else {
        P = fft_forward_lds<NT>(B, A, m, p.log2m, p.W, tid, source);  // (first stage: global -> A; B is its unused "in")
    }
    cfloat *Q = P == A ? B : A;
    cfloat *dst = p.dst + line * dls;
    // spectrum-picture mode (column pass): point k of this line is F(k, line); it lands at the quadrant-swapped
    // position of the |F| image
    const bool mag_mode = !pairs;
    const int sc = (int)line;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    auto emit = [&](int k, cfloat v) {
        if (!mag_mode) {
            dst[(int64_t)k * des] = v;
            return;
        }
        // |F(k, sc)| of this half of the spectrum (columns 0 .. C/2), at its own place: the picture kernel applies
        // fft_shift and takes the other half from the conjugate-symmetric points (F(-k, -c) = conj F(k, c))
        const float mg = sqrtf(v.x * v.x + v.y * v.y);
        p.mag_dst[(int64_t)sc * p.mag_pitch + k] = mg;
        lo = fminf(lo, mg);
        hi = fmaxf(hi, mg);
    };
    auto finish = [&]() {  // per-workgroup extrema of |F| (8 waves)
        if (!mag_mode) return;
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_down(lo, off));
            hi = fmaxf(hi, __shfl_down(hi, off));
        }
        __syncthreads();  // every wave is done with the LDS buffers
        float *red = (float *)lds_raw;
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = lo;
            red[2 * (tid >> 6) + 1] = hi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < FFT_THREADS / 64; w++) {
                lo = fminf(lo, red[2 * w]);
                hi = fmaxf(hi, red[2 * w + 1]);
            }
            p.part[2 * blockIdx.x] = lo;
            p.part[2 * blockIdx.x + 1] = hi;
        }
    };
    // two real rows per workgroup: split Z = FFT(a + i b) into FFT(a) and FFT(b), columns 0 .. n / 2.  zval(k): point k
    // of the pair's spectrum (read from LDS; on the Bluestein path the final chirp product is applied on the way)
    auto emit_pair = [&](auto zval) {
        cfloat *d0 = p.dst + line * dls, *d1 = d0 + dls;
        // transposed output: the two rows' points are neighbours (16 bytes, aligned when the line pitch is even): one store
        const bool wide = second && dls == 1 && (des & 1) == 0 && (((uintptr_t)p.dst) & 15) == 0;
        for (int k = tid; k <= n / 2; k += FFT_THREADS) {
            const cfloat zk = zval(k), zn = zval(k == 0 ? 0 : n - k);
            const cfloat a = cfloat{(0.5f * (zk.x + zn.x)) * p.out_scale, (0.5f * (zk.y - zn.y)) * p.out_scale};
            const cfloat b = cfloat{(0.5f * (zk.y + zn.y)) * p.out_scale, (-0.5f * (zk.x - zn.x)) * p.out_scale};
            if (wide) {
                *(float4 *)(d0 + (int64_t)k * des) = make_float4(a.x, a.y, b.x, b.y);
            } else {
                d0[(int64_t)k * des] = a;
                if (second) d1[(int64_t)k * des] = b;
            }
        }
    };
    if (!blue) {
        if (pairs) {
            emit_pair([&](int k) { return P[FPAD(k)]; });
            return;
        }
        for (int k = tid; k < n; k += FFT_THREADS) emit(k, cfloat{P[FPAD(k)].x * p.out_scale, P[FPAD(k)].y * p.out_scale});
        finish();
        return;
    }
    // convolution with the conjugate chirp: pointwise product, then an inverse FFT as conj(FFT(conj(.))) / m
    cfloat *R = P;
    if (inplace) {  // product, bit-reverse in place (pairs swap), then the second transform in the same buffer
        for (int k = tid; k < m; k += FFT_THREADS) {
            const cfloat c = cmul(P[FPAD(k)], p.Bf[k]);
            P[FPAD(k)] = cfloat{c.x, -c.y};
        }
        __syncthreads();
        for (int k = tid; k < m; k += FFT_THREADS) {
            const int r = (int)(__brev((unsigned)k) >> rshift);
            if (k < r) {
                const cfloat t = P[FPAD(k)];
                P[FPAD(k)] = P[FPAD(r)];
                P[FPAD(r)] = t;
            }
        }
        fft_inplace_lds<NT>(P, m, p.log2m, p.Wfull, tid);
    } else {
        R = fft_forward_lds<NT>(P, Q, m, p.log2m, p.W, tid, [&](int i) {  // (the product rides on the first stage's loads)
            const cfloat c = cmul(P[FPAD(i)], p.Bf[i]);
            return cfloat{c.x, -c.y};
        }, n);
    }
    const float inv_m = 1.0f / (float)m;
    if (pairs) {  // (the second transform ended with a barrier: the line's spectrum is complete in LDS)
        emit_pair([&](int k) { return cmul(cfloat{R[FPAD(k)].x * inv_m, -R[FPAD(k)].y * inv_m}, p.chirp[k]); });
        return;
    }
    for (int k = tid; k < n; k += FFT_THREADS) {
        const cfloat c = cmul(cfloat{R[FPAD(k)].x * inv_m, -R[FPAD(k)].y * inv_m}, p.chirp[k]);
        emit(k, cfloat{c.x * p.out_scale, c.y * p.out_scale});
    }
    finish();
}

hipError_t launch_fft_pass(const FftPass &p, hipStream_t s)
{
    if (p.lines <= 0) return hipSuccess;
    if (p.m > OMR_FFT_MAX_M || (1 << p.log2m) != p.m) return hipErrorInvalidValue;
    const size_t lds = (p.m > OMR_FFT_MAX_PINGPONG ? 1 : 2) * sizeof(cfloat) * (size_t)FFT_LDS_ELEMS(p.m);
    if (p.m > OMR_FFT_MAX_PINGPONG && !p.Wfull) return hipErrorInvalidValue;
    // Transforms of more than 4096 points hold the CU alone (> 80 KB of LDS): 1024 threads (4 waves per SIMD, one
    // radix-8 butterfly of a stage each) instead of 512 with two each.
    const dim3 grid(p.real_pairs ? (p.lines + 1) / 2 : p.lines, p.scans > 0 ? p.scans : 1);
    const int nt = p.m <= 4096 ? 512 : 1024;  // (640 threads for 5 * 1024 points measured no better than 1024)
#define FFT_LAUNCH(NT_, MODE_)                                                                                                \
    {                                                                                                                         \
        hipError_t e = hipFuncSetAttribute((const void *)fft_pass_kernel<NT_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                        \
        hipLaunchKernelGGL((fft_pass_kernel<NT_, MODE_>), grid, dim3(NT_), lds, s, p);                                        \
    }
    const bool rowpass = p.real_pairs != 0 && p.src_u8 != nullptr && p.dst != nullptr;
    const bool colpass = p.real_pairs == 0 && p.src_c != nullptr && p.mag_dst != nullptr;
    if (!rowpass && !colpass) return hipErrorInvalidValue;
    const bool plain = !p.chirp && p.m <= OMR_FFT_MAX_PINGPONG;  // power-of-two line, ping-pong buffers
    if (nt == 512) {
        if (plain && rowpass) FFT_LAUNCH(512, 1)
        else if (plain) FFT_LAUNCH(512, 0)
        else FFT_LAUNCH(512, 8)
    } else {
        FFT_LAUNCH(1024, 8)
    }
#undef FFT_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// float <-> unsigned keys that order like the floats (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t f2key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// per-block extrema go to part[2 * block .. +1]; minmax_final_kernel folds them (thousands of atomics
// on one address would serialise in L2 and cost more than the pass itself)
__device__ __forceinline__ void block_minmax(float lo, float hi, float *part, int block)
{
    __shared__ float s_lo[4], s_hi[4];
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            lo = fminf(lo, s_lo[w]);
            hi = fmaxf(hi, s_hi[w]);
        }
        part[2 * block] = lo;
        part[2 * block + 1] = hi;
    }
}

__global__ __launch_bounds__(1024) void minmax_final_kernel(const float *__restrict__ part, int n, uint32_t *mm,
                                                            int64_t part_scan_stride)
{
    part += (int64_t)blockIdx.x * part_scan_stride;
    mm += 4 * blockIdx.x;
    __shared__ float s_lo[16], s_hi[16];
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int i = threadIdx.x; i < n; i += 1024) {
        lo = fminf(lo, part[2 * i]);
        hi = fmaxf(hi, part[2 * i + 1]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) {
            lo = fminf(lo, s_lo[w]);
            hi = fmaxf(hi, s_hi[w]);
        }
        mm[0] = f2key(lo);
        mm[1] = f2key(hi);
    }
}

hipError_t launch_minmax_final(const float *d_part, int n, uint32_t *d_minmax, hipStream_t s, int scans,
                               int64_t part_scan_stride)
{
    hipLaunchKernelGGL(minmax_final_kernel, dim3(scans), dim3(1024), 0, s, d_part, n, d_minmax, part_scan_stride);
    return hipGetLastError();
}

// fft.rs:90-122, :134-138 in float32, the double alpha / beta of convert_to cast to float as OpenCV does.
// 4 pixels per lane along a row.
__device__ __forceinline__ float spec_m3(float mg, float beta, float alpha)
{
    const float c1 = mg * 1.0f + beta;
    const float c2 = c1 * alpha + 0.0f;
    return c2 * 255.0f + 0.0f;  // fft_magnitude
}
// (the hardware's v_log_f32 times ln 2 in place of logf passes the pictures' tolerance too but measured no faster)
__device__ __forceinline__ float spec_log(float m3) { return logf(m3 * 1.0f + (float)(1.0 / 255.0)); }

// magT: |F| of the half spectrum, transposed (cols / 2 + 1 lines of mag_pitch floats, line = spectrum column).  A workgroup turns a tile of 64 columns x 64 rows through LDS
// (reads run along a transposed line = down a picture column, writes along picture rows) into both pictures.
__global__ __launch_bounds__(256) void spec_pictures_kernel(const float *__restrict__ magT, int rows, int cols, int mag_pitch,
                                                            const uint32_t *__restrict__ mm, uint8_t *__restrict__ mag_u8,
                                                            uint8_t *__restrict__ log_u8, int64_t mag_scan_stride)
{
    __shared__ float tile[64][65];
    magT += (int64_t)blockIdx.z * mag_scan_stride;
    mm += 4 * blockIdx.z;
    if (mag_u8) mag_u8 += (int64_t)blockIdx.z * rows * cols;
    if (log_u8) log_u8 += (int64_t)blockIdx.z * rows * cols;
    const double mn = (double)key2f(mm[0]), mx = (double)key2f(mm[1]);
    const float beta = (float)(-mn), alpha = (float)(1.0 / (mx - mn));
    // extrema of the log picture = the log picture of the extrema (monotone float steps)
    const double lmn = (double)spec_log(spec_m3((float)mn, beta, alpha)), lmx = (double)spec_log(spec_m3((float)mx, beta, alpha));
    const float beta2 = (float)(-lmn), alpha2 = (float)(1.0 / (lmx - lmn));
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // picture pixel (r, c) <- fft_shift (fft.rs:67-86: the quadrants swap, an odd last row / column keeps its place) <-
    // spectrum point (k, sc); points of the right half come from their mirror images (R - k, C - sc)
    const int cxh = cols / 2, cyh = rows / 2;
    // Nearly every tile lies inside ONE quadrant and on one side of the mirror: the map (r, c) -> (k, sc) is then the
    // same affine one for all its pixels (workgroup-uniform, decided from the tile's corners) and the loads carry no
    // selects -- a v_cndmask costs this chip 17 cycles (profiles/r02_valu_issue.md) and the general form below has
    // nine per pixel.  The tiles on a quadrant border, in the first column tile (c = 0 is its own mirror), on the row
    // k = 0 of the mirrored half and on an odd last row / column take the general form.
    const int c_hi = min(c0 + 63, cols - 1), r_hi = min(r0 + 63, rows - 1);
    const bool mirrored = c0 >= 1 && c_hi < cxh, direct = c0 >= cxh && c_hi < 2 * cxh;
    const bool upper = r_hi < cyh, lower = r0 > cyh && r_hi < 2 * cyh;
    if ((mirrored || direct) && (upper || lower)) {
        const int r = min(r0 + tx, r_hi);  // (lanes past the picture repeat its last row / column: never stored)
        const int kq = upper ? r + cyh : r - cyh;
        const int k = mirrored ? rows - kq : kq;                 // (kq > 0 here)
        const int sc0 = mirrored ? cols - cxh : -cxh, scs = mirrored ? -1 : 1;  // sc = sc0 + scs * c
        const float *base = magT + k;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c = min(c0 + ty + 4 * i, c_hi);
            tile[ty + 4 * i][tx] = base[(int64_t)(sc0 + scs * c) * mag_pitch];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c = c0 + ty + 4 * i, r = r0 + tx;
            float v = 0.f;
            if (c < cols && r < rows) {
                const bool inq = c < 2 * cxh && r < 2 * cyh;
                int k = inq ? (r < cyh ? r + cyh : r - cyh) : r;
                int sc = inq ? (c < cxh ? c + cxh : c - cxh) : c;
                if (sc > cxh) {
                    sc = cols - sc;
                    k = k == 0 ? 0 : rows - k;
                }
                v = magT[(int64_t)sc * mag_pitch + k];
            }
            tile[ty + 4 * i][tx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int item = threadIdx.x + 256 * q;
        const int rl = item >> 4, cl = (item & 15) * 4;
        const int r = r0 + rl, cc = c0 + cl;
        if (r >= rows || cc >= cols) continue;
        uint32_t om = 0, ol = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (cc + j < cols) {
                const float m3 = spec_m3(tile[cl + j][rl], beta, alpha);
                const float u = rintf(m3 * 255.0f + 0.0f);  // convert_to(CV_8UC1, 255)
                om |= (uint32_t)fminf(fmaxf(u, 0.f), 255.f) << (8 * j);
                const float l = spec_log(m3);
                const float v = rintf(((l * 1.0f + beta2) * alpha2 + 0.0f) * 255.0f + 0.0f);
                ol |= (uint32_t)fminf(fmaxf(v, 0.f), 255.f) << (8 * j);
            }
        }
        const int64_t o = (int64_t)r * cols + cc;
        if (cc + 4 <= cols && (o & 3) == 0) {
            if (mag_u8) *(uint32_t *)(mag_u8 + o) = om;
            if (log_u8) *(uint32_t *)(log_u8 + o) = ol;
        } else {
            for (int j = 0; j < 4 && cc + j < cols; j++) {
                if (mag_u8) mag_u8[o + j] = (uint8_t)(om >> (8 * j));
                if (log_u8) log_u8[o + j] = (uint8_t)(ol >> (8 * j));
            }
        }
    }
}

hipError_t launch_spec_pictures(const float *d_mag, int rows, int cols, int mag_pitch, const uint32_t *d_minmax,
                                uint8_t *d_mag_u8, uint8_t *d_log_u8, hipStream_t s, int scans, int64_t mag_scan_stride)
{
    hipLaunchKernelGGL(spec_pictures_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, scans), dim3(256), 0, s, d_mag, rows, cols,
                       mag_pitch, d_minmax, d_mag_u8, d_log_u8, mag_scan_stride);
    return hipGetLastError();
}

}  // namespace omr
