// fft_big.hip -- lines longer than one workgroup's LDS can transform (fft.hpp: OMR_FFT_MAX_M = 16384 points, i.e. a
// line of more than 8192 points that is not a power of two): Bluestein's chirp-z with transforms of M = 32768 or
// 65536 points that live in GLOBAL memory, each one a four-step FFT (M = M1 x 256):
//
//   x[n1 * 256 + n2]  --(M1-point FFTs over n1, one per n2)-->  Y[k1][n2]  --(* w_M^(n2 k1))-->
//                     --(256-point FFTs over n2, one per k1)-->  X[k1 + M1 k2] at [k1 * 256 + k2]
//
// so a forward transform leaves its result transposed ([k1][k2]); the chirp's spectrum is tabulated in that order and
// the inverse transform runs the same steps backwards (conjugate twiddles), which brings the natural order back.
// Each step stages tiles of 16 sequences through LDS (16 x 256 points = 32 KB): the column step reads 16 neighbouring
// n2 (128-byte pieces), the row step 16 whole rows.  The reference transforms a scan at its own size whatever it is
// (packages/lib/src/fft.rs:42-65, cv::dft); this is the slow, general path behind the LDS kernels of fft.hip /
// fft_mixed.hip, which keep every length up to 8192 (and 16384): six passes over a line's M points per transform
// pair instead of none.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft.hpp"

namespace omr {

namespace {

__device__ __forceinline__ cfloat bmul(const cfloat a, const cfloat b) { return cfloat{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cfloat bconj(const cfloat a) { return cfloat{a.x, -a.y}; }

#define BIG_C 16        // sequences per tile
#define BIG_NT 256      // threads per workgroup
#define BIG_M2 256

// radix-2 decimation in time over BIG_C sequences of L points held bit-reversed in a[c][.]; wM: M twiddles
// exp(-2 pi i t / M); INV: conjugate twiddles (no scaling)
template <int L, bool INV>
__device__ __forceinline__ void lds_fft(cfloat (*a)[L + 1], const cfloat *__restrict__ wM, int M, int tid)
{
    for (int len = 2; len <= L; len <<= 1) {
        const int half = len >> 1, tw = M / len;
        __syncthreads();
        for (int t = tid; t < BIG_C * (L / 2); t += BIG_NT) {
            const int c = t / (L / 2), b = t % (L / 2);
            const int j = b % half, i = (b / half) * len + j;
            cfloat w = wM[(int64_t)j * tw];
            if (INV) w.y = -w.y;
            const cfloat u = a[c][i], v = bmul(a[c][i + half], w);
            a[c][i] = cfloat{u.x + v.x, u.y + v.y};
            a[c][i + half] = cfloat{u.x - v.x, u.y - v.y};
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int brev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// column step: grid (256 / 16, lines).  Forward: FFT over n1 then * w_M^(n2 k1).  Inverse: * conj w_M^(n2 k1), then
// the inverse FFT over k1.
template <int M1, bool INV>
__global__ __launch_bounds__(BIG_NT) void big_cols_kernel(cfloat *__restrict__ buf, const cfloat *__restrict__ wM, int M)
{
    __shared__ cfloat a[BIG_C][M1 + 1];
    constexpr int LOG = M1 == 128 ? 7 : 8;
    cfloat *line = buf + (int64_t)blockIdx.y * M;
    const int n20 = blockIdx.x * BIG_C, tid = threadIdx.x;
    for (int t = tid; t < BIG_C * M1; t += BIG_NT) {
        const int c = t % BIG_C, r = t / BIG_C;
        cfloat v = line[(int64_t)r * BIG_M2 + n20 + c];
        if (INV) v = bmul(v, bconj(wM[(int64_t)(n20 + c) * r]));
        a[c][brev(r, LOG)] = v;
    }
    lds_fft<M1, INV>(a, wM, M, tid);
    for (int t = tid; t < BIG_C * M1; t += BIG_NT) {
        const int c = t % BIG_C, r = t / BIG_C;
        cfloat v = a[c][r];
        if (!INV) v = bmul(v, wM[(int64_t)(n20 + c) * r]);
        line[(int64_t)r * BIG_M2 + n20 + c] = v;
    }
}

// row step: grid (M1 / 16, lines): 256-point transforms of 16 rows; forward results are multiplied by BfT (the
// chirp's spectrum in this transposed order) on the way out -- Bluestein's pointwise product
template <bool INV>
__global__ __launch_bounds__(BIG_NT) void big_rows_kernel(cfloat *__restrict__ buf, const cfloat *__restrict__ wM, int M,
                                                          const cfloat *__restrict__ BfT)
{
    __shared__ cfloat a[BIG_C][BIG_M2 + 1];
    cfloat *line = buf + (int64_t)blockIdx.y * M;
    const int r0 = blockIdx.x * BIG_C, tid = threadIdx.x;
    for (int t = tid; t < BIG_C * BIG_M2; t += BIG_NT) {
        const int c = t / BIG_M2, k = t % BIG_M2;
        a[c][brev(k, 8)] = line[(int64_t)(r0 + c) * BIG_M2 + k];
    }
    lds_fft<BIG_M2, INV>(a, wM, M, tid);
    for (int t = tid; t < BIG_C * BIG_M2; t += BIG_NT) {
        const int c = t / BIG_M2, k = t % BIG_M2;
        cfloat v = a[c][k];
        if (!INV) v = bmul(v, BfT[(int64_t)(r0 + c) * BIG_M2 + k]);
        line[(int64_t)(r0 + c) * BIG_M2 + k] = v;
    }
}

// buf[l][t] = input point t of line l times the chirp, zero beyond n.  pairs: line l = rows 2 l (real part) and
// 2 l + 1 (imaginary part) of the 8-bit scan
__global__ __launch_bounds__(256) void big_load_kernel(BigLines p, int l0)
{
    const int l = blockIdx.y, line = l0 + l;
    cfloat *dst = p.buf + (int64_t)l * p.M;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < p.M; t += gridDim.x * 256) {
        cfloat v{0.f, 0.f};
        if (t < p.n) {
            if (p.src_u8) {
                const int64_t r = 2 * (int64_t)line;
                v.x = (float)p.src_u8[r * p.src_step + t] * p.in_scale + 0.0f;
                if (r + 1 < p.src_rows) v.y = (float)p.src_u8[(r + 1) * p.src_step + t] * p.in_scale + 0.0f;
            } else {
                v = p.src_c[(int64_t)line * p.line_stride + t];
            }
            v = bmul(v, p.chirp[t]);
        }
        dst[t] = v;
    }
}

// point k of line l's spectrum: the convolution's point times the chirp, / M (the inverse transform's scale)
__device__ __forceinline__ cfloat big_point(const BigLines &p, const cfloat *line, int k)
{
    const cfloat v = bmul(line[k], p.chirp[k]);
    return cfloat{v.x * p.inv_m, v.y * p.inv_m};
}

// row pass of the pictures: the pair's two spectra (Hermitian split, fft.hip emit_pair), columns 0 .. n / 2, stored
// TRANSPOSED: dst[k * pitch + row].  Tile = 32 pair lines x 32 points through LDS so that both sides run along
// consecutive addresses.  grid (ceil((n / 2 + 1) / 32), ceil(lines / 32))
__global__ __launch_bounds__(256) void big_store_pairs_kernel(BigLines p, int l0, int nl)
{
    __shared__ cfloat ta[32][65];
    const int k0 = blockIdx.x * 32, lb = blockIdx.y * 32, n = p.n;
    for (int t = threadIdx.x; t < 32 * 32; t += 256) {
        const int kk = t % 32, ll = t / 32, k = k0 + kk, l = lb + ll;
        if (k <= n / 2 && l < nl) {
            const cfloat *line = p.buf + (int64_t)l * p.M;
            const cfloat zk = big_point(p, line, k), zn = big_point(p, line, k == 0 ? 0 : n - k);
            ta[kk][2 * ll] = cfloat{(0.5f * (zk.x + zn.x)) * p.out_scale, (0.5f * (zk.y - zn.y)) * p.out_scale};
            ta[kk][2 * ll + 1] = cfloat{(0.5f * (zk.y + zn.y)) * p.out_scale, (-0.5f * (zk.x - zn.x)) * p.out_scale};
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * 64; t += 256) {
        const int rr = t % 64, kk = t / 64, k = k0 + kk;
        const int64_t row = 2 * (int64_t)(l0 + lb) + rr;
        if (k <= n / 2 && lb + rr / 2 < nl && row < p.src_rows) p.dst[(int64_t)k * p.dst_pitch + row] = ta[kk][rr];
    }
}

// column pass of the pictures: |F(k, line)| * out_scale -> mag[line * mag_pitch + k] and the line's extrema
// (fft.hip's spectrum-picture mode).  grid (lines of the chunk)
__global__ __launch_bounds__(256) void big_store_mag_kernel(BigLines p, int l0)
{
    __shared__ float red[2 * 4];
    const int l = blockIdx.x, line = l0 + l;
    const cfloat *src = p.buf + (int64_t)l * p.M;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int k = threadIdx.x; k < p.n; k += 256) {
        cfloat v = big_point(p, src, k);
        v.x *= p.out_scale;
        v.y *= p.out_scale;
        const float mg = sqrtf(v.x * v.x + v.y * v.y);
        p.mag_dst[(int64_t)line * p.mag_pitch + k] = mg;
        lo = fminf(lo, mg);
        hi = fmaxf(hi, mg);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        red[2 * (threadIdx.x >> 6)] = lo;
        red[2 * (threadIdx.x >> 6) + 1] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            lo = fminf(lo, red[2 * w]);
            hi = fmaxf(hi, red[2 * w + 1]);
        }
        p.part[2 * line] = lo;
        p.part[2 * line + 1] = hi;
    }
}

}  // namespace

// `lines` lines through the chunk buffer p.buf (p.chunk lines of M points): load, chirp-z, store
hipError_t launch_big_lines(const BigLines &p, hipStream_t s)
{
    if (p.lines <= 0) return hipSuccess;
    if ((p.M != 32768 && p.M != 65536) || p.n > p.M / 2 || p.chunk <= 0 || !p.buf || !p.chirp || !p.BfT || !p.wM) return hipErrorInvalidValue;
    const bool pairs = p.src_u8 != nullptr;
    if (pairs ? !p.dst : (!p.src_c || !p.mag_dst || !p.part)) return hipErrorInvalidValue;
    const int M1 = p.M / BIG_M2;
    for (int l0 = 0; l0 < p.lines; l0 += p.chunk) {
        const int nl = p.lines - l0 < p.chunk ? p.lines - l0 : p.chunk;
        hipLaunchKernelGGL(big_load_kernel, dim3(16, nl), dim3(256), 0, s, p, l0);
        if (M1 == 128) hipLaunchKernelGGL((big_cols_kernel<128, false>), dim3(BIG_M2 / BIG_C, nl), dim3(BIG_NT), 0, s, p.buf, p.wM, p.M);
        else hipLaunchKernelGGL((big_cols_kernel<256, false>), dim3(BIG_M2 / BIG_C, nl), dim3(BIG_NT), 0, s, p.buf, p.wM, p.M);
        hipLaunchKernelGGL((big_rows_kernel<false>), dim3(M1 / BIG_C, nl), dim3(BIG_NT), 0, s, p.buf, p.wM, p.M, p.BfT);
        hipLaunchKernelGGL((big_rows_kernel<true>), dim3(M1 / BIG_C, nl), dim3(BIG_NT), 0, s, p.buf, p.wM, p.M, p.BfT);
        if (M1 == 128) hipLaunchKernelGGL((big_cols_kernel<128, true>), dim3(BIG_M2 / BIG_C, nl), dim3(BIG_NT), 0, s, p.buf, p.wM, p.M);
        else hipLaunchKernelGGL((big_cols_kernel<256, true>), dim3(BIG_M2 / BIG_C, nl), dim3(BIG_NT), 0, s, p.buf, p.wM, p.M);
        if (pairs) hipLaunchKernelGGL(big_store_pairs_kernel, dim3((p.n / 2 + 1 + 31) / 32, (nl + 31) / 32), dim3(256), 0, s, p, l0, nl);
        else hipLaunchKernelGGL(big_store_mag_kernel, dim3(nl), dim3(256), 0, s, p, l0);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace omr
