// fft.hip -- 2-D complex float32 DFT and the spectrum pictures of the reference's FFT deskew path
// (packages/lib/src/fft.rs:42-141) for gfx950.
//
// The reference transforms the scan at its own size (no padding to a fast size), so lengths like
// 2480 = 2^4 * 5 * 31 or 3508 = 2^2 * 877 must work: a line is transformed entirely inside LDS, by a
// radix-2 Stockham FFT when its length is a power of two and by Bluestein's chirp-z (two power-of-two
// FFTs of length m >= 2n - 1 and three pointwise products) otherwise.  Twiddles and chirps are
// tabulated by the host in double precision.  The 2-D transform is rows -> transpose -> rows ->
// transpose; float32 throughout (the reference's dft is CV_32F), built without FMA contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft.hpp"

namespace omr {

__device__ __forceinline__ cfloat cmul(const cfloat a, const cfloat b)
{
    return cfloat{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

#define FFT_THREADS 512

// forward FFT of length m = 2^log2m, Stockham autosort, radix 2: ping-pong between `in` and `out`;
// returns the buffer that holds the result (natural order)
__device__ cfloat *fft_forward_lds(cfloat *in, cfloat *out, const int m, const int log2m,
                                   const cfloat *__restrict__ W, const int tid)
{
    const int half = m >> 1;
    for (int s = 0; s < log2m; s++) {
        const int Ns = 1 << s;
        __syncthreads();
        for (int j = tid; j < half; j += FFT_THREADS) {
            const int k = j & (Ns - 1);
            const cfloat u0 = in[j];
            const cfloat u1 = cmul(in[j + half], W[k << (log2m - 1 - s)]);  // exp(-2 pi i k / (2 Ns))
            const int j0 = ((j - k) << 1) + k;
            out[j0] = cfloat{u0.x + u1.x, u0.y + u1.y};
            out[j0 + Ns] = cfloat{u0.x - u1.x, u0.y - u1.y};
        }
        cfloat *t = in;
        in = out;
        out = t;
    }
    __syncthreads();
    return in;
}

__global__ __launch_bounds__(FFT_THREADS) void fft_pass_kernel(const FftPass p)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    cfloat *A = (cfloat *)lds_raw, *B = A + p.m;
    const int tid = threadIdx.x, n = p.n, m = p.m;
    const int64_t line = blockIdx.x;
    const bool blue = p.chirp != nullptr;
    for (int k = tid; k < m; k += FFT_THREADS) {
        cfloat v{0.f, 0.f};
        if (k < n) {
            if (p.src_u8) v.x = (float)p.src_u8[line * p.src_step + k] * p.in_scale + 0.0f;
            else v = p.src_c[line * n + k];
            if (blue) v = cmul(v, p.chirp[k]);
        }
        A[k] = v;
    }
    cfloat *P = fft_forward_lds(A, B, m, p.log2m, p.W, tid);
    cfloat *Q = P == A ? B : A;
    cfloat *dst = p.dst + line * n;
    if (!blue) {
        for (int k = tid; k < n; k += FFT_THREADS) dst[k] = cfloat{P[k].x * p.out_scale, P[k].y * p.out_scale};
        return;
    }
    // convolution with the conjugate chirp: pointwise product, then an inverse FFT as conj(FFT(conj(.))) / m
    for (int k = tid; k < m; k += FFT_THREADS) {
        const cfloat c = cmul(P[k], p.Bf[k]);
        P[k] = cfloat{c.x, -c.y};
    }
    cfloat *R = fft_forward_lds(P, Q, m, p.log2m, p.W, tid);
    const float inv_m = 1.0f / (float)m;
    for (int k = tid; k < n; k += FFT_THREADS) {
        const cfloat c = cmul(cfloat{R[k].x * inv_m, -R[k].y * inv_m}, p.chirp[k]);
        dst[k] = cfloat{c.x * p.out_scale, c.y * p.out_scale};
    }
}

hipError_t launch_fft_pass(const FftPass &p, hipStream_t s)
{
    if (p.lines <= 0) return hipSuccess;
    if (p.m > OMR_FFT_MAX_M || (1 << p.log2m) != p.m) return hipErrorInvalidValue;
    const size_t lds = 2 * sizeof(cfloat) * (size_t)p.m;
    hipError_t e = hipFuncSetAttribute((const void *)fft_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fft_pass_kernel, dim3(p.lines), dim3(FFT_THREADS), lds, s, p);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void transpose_c_kernel(const cfloat *__restrict__ src, int rows, int cols,
                                                          cfloat *__restrict__ dst)
{
    __shared__ cfloat t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int y = y0 + i, x = x0 + tx;
        if (y < rows && x < cols) t[i][tx] = src[(int64_t)y * cols + x];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int x = x0 + i, y = y0 + tx;  // dst row = source column
        if (x < cols && y < rows) dst[(int64_t)x * rows + y] = t[tx][i];
    }
}

hipError_t launch_transpose_c(const cfloat *d_src, int rows, int cols, cfloat *d_dst, hipStream_t s)
{
    hipLaunchKernelGGL(transpose_c_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, d_src, rows, cols,
                       d_dst);
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

__global__ void spec_reset_kernel(uint32_t *mm)
{
    mm[0] = mm[2] = 0xffffffffu;
    mm[1] = mm[3] = 0u;
}

hipError_t launch_spec_reset(uint32_t *d_minmax, hipStream_t s)
{
    hipLaunchKernelGGL(spec_reset_kernel, dim3(1), dim3(1), 0, s, d_minmax);
    return hipGetLastError();
}

__device__ __forceinline__ void block_minmax(float lo, float hi, uint32_t *mm)
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
        atomicMin(mm, f2key(lo));
        atomicMax(mm + 1, f2key(hi));
    }
}

// out(r, c) = |F(sr, sc)|: quadrants of cx x cy swapped diagonally, an odd last row / column untouched
__global__ __launch_bounds__(256) void spec_magnitude_kernel(const cfloat *__restrict__ F, int rows, int cols,
                                                             float *__restrict__ mag, uint32_t *__restrict__ mm)
{
    const int cx = cols / 2, cy = rows / 2;
    const int64_t total = (int64_t)rows * cols;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        int sr = r, sc = c;
        if (r < 2 * cy && c < 2 * cx) {
            sr = r < cy ? r + cy : r - cy;
            sc = c < cx ? c + cx : c - cx;
        }
        const cfloat v = F[(int64_t)sr * cols + sc];
        const float m = sqrtf(v.x * v.x + v.y * v.y);
        mag[i] = m;
        lo = fminf(lo, m);
        hi = fmaxf(hi, m);
    }
    block_minmax(lo, hi, mm);
}

hipError_t launch_spec_magnitude(const cfloat *d_F, int rows, int cols, float *d_mag, uint32_t *d_minmax, hipStream_t s)
{
    const int64_t total = (int64_t)rows * cols;
    const int blocks = (int)((total + 1023) / 1024 < 8192 ? (total + 1023) / 1024 : 8192);
    hipLaunchKernelGGL(spec_magnitude_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, d_F, rows, cols, d_mag, d_minmax);
    return hipGetLastError();
}

// fft.rs:90-122 in float32, the double alpha / beta of convert_to cast to float as OpenCV does
__global__ __launch_bounds__(256) void spec_normalise_kernel(const float *__restrict__ mag, int64_t total,
                                                             const uint32_t *__restrict__ mm_in,
                                                             uint8_t *__restrict__ mag_u8, float *__restrict__ lg,
                                                             uint32_t *__restrict__ mm_out)
{
    const double mn = (double)key2f(mm_in[0]), mx = (double)key2f(mm_in[1]);
    const float beta = (float)(-mn), alpha = (float)(1.0 / (mx - mn));
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float c1 = mag[i] * 1.0f + beta;
        const float c2 = c1 * alpha + 0.0f;
        const float m3 = c2 * 255.0f + 0.0f;                      // fft_magnitude
        const float u = rintf(m3 * 255.0f + 0.0f);                // convert_to(CV_8UC1, 255)
        mag_u8[i] = (uint8_t)fminf(fmaxf(u, 0.f), 255.f);
        const float l = logf(m3 * 1.0f + (float)(1.0 / 255.0));  // fft_magnitude_log before its correction
        lg[i] = l;
        lo = fminf(lo, l);
        hi = fmaxf(hi, l);
    }
    block_minmax(lo, hi, mm_out);
}

hipError_t launch_spec_normalise(const float *d_mag, int rows, int cols, const uint32_t *d_minmax_in, uint8_t *d_mag_u8,
                                 float *d_log, uint32_t *d_minmax_out, hipStream_t s)
{
    const int64_t total = (int64_t)rows * cols;
    const int blocks = (int)((total + 1023) / 1024 < 8192 ? (total + 1023) / 1024 : 8192);
    hipLaunchKernelGGL(spec_normalise_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, d_mag, total, d_minmax_in,
                       d_mag_u8, d_log, d_minmax_out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void spec_log_u8_kernel(const float *__restrict__ lg, int64_t total,
                                                          const uint32_t *__restrict__ mm, uint8_t *__restrict__ out)
{
    const double mn = (double)key2f(mm[0]), mx = (double)key2f(mm[1]);
    const float beta = (float)(-mn), alpha = (float)(1.0 / (mx - mn));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float c1 = lg[i] * 1.0f + beta;
        const float c2 = c1 * alpha + 0.0f;
        const float u = rintf(c2 * 255.0f + 0.0f);
        out[i] = (uint8_t)fminf(fmaxf(u, 0.f), 255.f);
    }
}

hipError_t launch_spec_log_u8(const float *d_log, int rows, int cols, const uint32_t *d_minmax, uint8_t *d_log_u8,
                              hipStream_t s)
{
    const int64_t total = (int64_t)rows * cols;
    const int blocks = (int)((total + 1023) / 1024 < 8192 ? (total + 1023) / 1024 : 8192);
    hipLaunchKernelGGL(spec_log_u8_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, d_log, total, d_minmax, d_log_u8);
    return hipGetLastError();
}

}  // namespace omr
