// oics_fft.cpp -- host side of the FFT deskew path, exported through the C ABI
// with the reference's names and argument meaning:
//
//   oics::fft::get_fft_image                       packages/lib/src/fft.rs:124-141
//   oics::fft::get_angle_with_fft                  packages/lib/src/fft.rs:145-256
//   oics::omr::get_result_from_fourier_transform   packages/lib/src/omr.rs:304-337
//
// The DFT, the spectrum pictures, Canny and HoughLinesP run on the GPU (fft.hip, hough.hip); the host
// tabulates twiddles and Bluestein chirps in double precision (once per call -- a few thousand
// cos / sin and one length-m double FFT per axis), turns segments into angles with libm atan2 and
// applies the reference's vote.  No CPU fallback: without a HIP device every entry point returns -217.
#include <math.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <string.h>

#include <algorithm>
#include <complex>
#include <vector>

#include "../../include/omrdeskew.h"
#include "engine.hpp"
#include "fft.hpp"
#include "hough.hpp"
#include "hough_host.hpp"
#include "kernels.hpp"

using namespace omr;
using namespace omr::hh;

namespace {

typedef std::complex<double> cd;
const double kPi = 3.14159265358979323846;

void fft_host(std::vector<cd> &a)  // iterative radix-2, in place, forward
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                const double ang = -2.0 * kPi * (double)k / (double)len;
                const cd w(cos(ang), sin(ang));
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

void dft_host(std::vector<cd> &a)  // forward DFT of any length: radix 2 for powers of two, else the definition (once per axis length)
{
    const size_t n = a.size();
    if ((n & (n - 1)) == 0) return fft_host(a);
    std::vector<cd> w(n), out(n);
    for (size_t t = 0; t < n; t++) {
        const double ang = -2.0 * kPi * (double)t / (double)n;
        w[t] = cd(cos(ang), sin(ang));
    }
    for (size_t k = 0; k < n; k++) {
        cd acc(0, 0);
        size_t idx = 0;
        for (size_t j = 0; j < n; j++) {
            acc += a[j] * w[idx];
            idx += k;
            if (idx >= n) idx -= n;
        }
        out[k] = acc;
    }
    a.swap(out);
}

// device tables of one axis length
struct AxisTables {
    int n = 0, m = 0, log2m = 0;
    bool blue = false, mixed = false;  // mixed: one of fft_mixed.hip's kernels takes this length
    int sub = 0;                       // ... as Bluestein on this many sub-lines
    bool big = false;                  // fft_big.hip: chirp-z through global memory (m = 32768 or 65536; Bf = BfT, Wfull = wM)
    DevBuf W, Wfull, chirp, Bf;
    hipError_t launch(const FftPass &p, hipStream_t s) const { return mixed ? launch_fft_mixed(p, s) : launch_fft_pass(p, s); }
    // fft_mixed.hip's stage tables: per stage (radix R after Ns points) R == 16 -> w^k, else [q - 1][k] = w^(q k), w =
    // exp(-2 pi i / (R Ns)); nothing for the first stage
    static int stage_twiddles(const int *r, int ns, std::vector<cfloat> *w)
    {
        int Ns = 1;
        for (int i = 0; i < ns; i++) {
            if (Ns > 1) {
                const double step = -2.0 * kPi / ((double)r[i] * (double)Ns);
                for (int q = 1; q < (r[i] == 16 ? 2 : r[i]); q++)
                    for (int k = 0; k < Ns; k++) {
                        const double ang = step * (double)(((long long)q * k) % ((long long)r[i] * Ns));
                        w->push_back(cfloat{(float)cos(ang), (float)sin(ang)});
                    }
            }
            Ns *= r[i];
        }
        return Ns;
    }
    int build_mixed(hipStream_t s)
    {
        int r[4];
        const int ns = fft_mixed_radices(n, r);
        std::vector<cfloat> w;
        const int Ns = stage_twiddles(r, ns, &w);
        if ((int)w.size() != fft_mixed_table_size(n) || Ns != n) return fail(OMR_ERR_ASSERT, "mixed-radix plan of length %d", n);
        mixed = true;
        blue = false;
        m = n;
        log2m = 0;
        OMR_HIP(W.alloc(sizeof(cfloat) * w.size()));
        OMR_HIP(hipMemcpyAsync(W.p, w.data(), sizeof(cfloat) * w.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipStreamSynchronize(s));  // the host vector goes out of scope
        return OMR_OK;
    }
    // chirp of P points and the spectrum of its padded conjugate (Bluestein with transforms of mm points)
    static void chirp_tables(int P, int mm, std::vector<cd> *c, std::vector<cd> *b)
    {
        c->resize((size_t)P);
        for (int k = 0; k < P; k++) {
            const long long k2 = ((long long)k * k) % (2LL * P);  // exp(-i pi k^2 / P) has period 2 P in k^2
            const double ang = -kPi * (double)k2 / (double)P;
            (*c)[k] = cd(cos(ang), sin(ang));
        }
        b->assign((size_t)mm, cd(0, 0));
        (*b)[0] = std::conj((*c)[0]);
        for (int k = 1; k < P; k++) (*b)[k] = (*b)[mm - k] = std::conj((*c)[k]);
        dft_host(*b);
    }
    int build_sub(hipStream_t s)  // fft_mixed.hip: Bluestein on `sub` interleaved sub-lines of P = n / sub points, m = 1792 or 2048
    {
        const int P = n / sub;
        mixed = true;
        blue = true;
        log2m = 0;
        int r[3];
        const int ns = fft_bluesub_plan(n, &m, r);
        std::vector<cfloat> w;
        if (!ns || stage_twiddles(r, ns, &w) != m || (int)w.size() != fft_bluesub_stage_table_size(n) || m < 2 * P - 1)
            return fail(OMR_ERR_ASSERT, "sub-line plan of length %d", n);
        std::vector<cd> c, b;
        chirp_tables(P, m, &c, &b);
        for (int q = 0; q < sub; q++)
            for (int k = 0; k < P; k++) {
                const double ang = -2.0 * kPi * (double)(((long long)q * k) % n) / (double)n;
                const cd g = c[k] * cd(cos(ang), sin(ang)) / (double)m;
                w.push_back(cfloat{(float)g.real(), (float)g.imag()});
            }
        std::vector<cfloat> cf((size_t)P), bf((size_t)m);
        for (int k = 0; k < P; k++) cf[k] = cfloat{(float)c[k].real(), (float)c[k].imag()};
        for (int k = 0; k < m; k++) bf[k] = cfloat{(float)b[k].real(), (float)b[k].imag()};
        OMR_HIP(W.alloc(sizeof(cfloat) * w.size()));
        OMR_HIP(chirp.alloc(sizeof(cfloat) * cf.size()));
        OMR_HIP(Bf.alloc(sizeof(cfloat) * bf.size()));
        OMR_HIP(hipMemcpyAsync(W.p, w.data(), sizeof(cfloat) * w.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipMemcpyAsync(chirp.p, cf.data(), sizeof(cfloat) * cf.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipMemcpyAsync(Bf.p, bf.data(), sizeof(cfloat) * bf.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipStreamSynchronize(s));  // the host vectors go out of scope
        return OMR_OK;
    }
    int build_big(hipStream_t s)  // fft_big.hip: the chirp, its spectrum in four-step order, the m twiddles
    {
        big = true;
        std::vector<cd> c, b;
        chirp_tables(n, m, &c, &b);
        const int M1 = m / 256;
        std::vector<cfloat> cf((size_t)n), bt((size_t)m), wf((size_t)m);
        for (int k = 0; k < n; k++) cf[k] = cfloat{(float)c[k].real(), (float)c[k].imag()};
        for (int k1 = 0; k1 < M1; k1++)
            for (int k2 = 0; k2 < 256; k2++) {
                const cd v = b[(size_t)k1 + (size_t)M1 * k2];
                bt[(size_t)k1 * 256 + k2] = cfloat{(float)v.real(), (float)v.imag()};
            }
        for (int t = 0; t < m; t++) {
            const double ang = -2.0 * kPi * (double)t / (double)m;
            wf[t] = cfloat{(float)cos(ang), (float)sin(ang)};
        }
        OMR_HIP(chirp.alloc(sizeof(cfloat) * cf.size()));
        OMR_HIP(Bf.alloc(sizeof(cfloat) * bt.size()));
        OMR_HIP(Wfull.alloc(sizeof(cfloat) * wf.size()));
        OMR_HIP(hipMemcpyAsync(chirp.p, cf.data(), sizeof(cfloat) * cf.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipMemcpyAsync(Bf.p, bt.data(), sizeof(cfloat) * bt.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipMemcpyAsync(Wfull.p, wf.data(), sizeof(cfloat) * wf.size(), hipMemcpyHostToDevice, s));
        OMR_HIP(hipStreamSynchronize(s));  // the host vectors go out of scope
        return OMR_OK;
    }
    int build(int len, hipStream_t s)
    {
        n = len;
        {
            int r[4];
            if (fft_mixed_radices(n, r) > 0) return build_mixed(s);
            if ((sub = fft_bluesub_lines(n)) > 0) return build_sub(s);
        }
        blue = (n & (n - 1)) != 0;
        const int need = blue ? 2 * n - 1 : n;
        m = 1;
        log2m = 0;
        while (m < need) {
            m <<= 1;
            log2m++;
        }
        if (m > OMR_FFT_BIG_MAX_M)
            return fail(OMR_ERR_NOTIMPL, "DFT length %d needs a transform of %d points; at most %d are supported", n, m,
                        OMR_FFT_BIG_MAX_M);
        if (m > OMR_FFT_MAX_M) return build_big(s);
        // twiddles of the radix-8 stages, one contiguous table per stage (Ns = Ns0, 8 Ns0, ... < m)
        std::vector<cfloat> w;
        // (lengths 2^(3a+1) >= 128 end in one radix-16 stage instead of starting with a radix-2 stage: fft_forward_lds)
        const bool tail16 = log2m % 3 == 1 && log2m >= 7 && m <= OMR_FFT_MAX_PINGPONG;
        const int m8 = tail16 ? m / 16 : m;
        for (int Ns = tail16 ? 1 : 1 << (log2m % 3); Ns < m8; Ns *= 8)
            for (int k = 0; k < Ns; k++) {
                const double ang = -2.0 * kPi * (double)k / (8.0 * (double)Ns);
                w.push_back(cfloat{(float)cos(ang), (float)sin(ang)});
            }
        if (tail16)
            for (int k = 0; k < m / 16; k++) {
                const double ang = -2.0 * kPi * (double)k / (double)m;
                w.push_back(cfloat{(float)cos(ang), (float)sin(ang)});
            }
        if (w.empty()) w.push_back(cfloat{1.f, 0.f});
        if (m > OMR_FFT_MAX_PINGPONG) {  // the in-place radix-2 path reads one full table
            std::vector<cfloat> wf((size_t)m / 2);
            for (int t = 0; t < m / 2; t++) {
                const double ang = -2.0 * kPi * (double)t / (double)m;
                wf[t] = cfloat{(float)cos(ang), (float)sin(ang)};
            }
            OMR_HIP(Wfull.alloc(sizeof(cfloat) * wf.size()));
            OMR_HIP(hipMemcpy(Wfull.p, wf.data(), sizeof(cfloat) * wf.size(), hipMemcpyHostToDevice));
        }
        OMR_HIP(W.alloc(sizeof(cfloat) * w.size()));
        OMR_HIP(hipMemcpyAsync(W.p, w.data(), sizeof(cfloat) * w.size(), hipMemcpyHostToDevice, s));
        if (blue) {
            std::vector<cd> c, b;
            chirp_tables(n, m, &c, &b);
            std::vector<cfloat> cf((size_t)n), bf((size_t)m);
            for (int k = 0; k < n; k++) cf[k] = cfloat{(float)c[k].real(), (float)c[k].imag()};
            for (int k = 0; k < m; k++) bf[k] = cfloat{(float)b[k].real(), (float)b[k].imag()};
            OMR_HIP(chirp.alloc(sizeof(cfloat) * cf.size()));
            OMR_HIP(Bf.alloc(sizeof(cfloat) * bf.size()));
            OMR_HIP(hipMemcpyAsync(chirp.p, cf.data(), sizeof(cfloat) * cf.size(), hipMemcpyHostToDevice, s));
            OMR_HIP(hipMemcpyAsync(Bf.p, bf.data(), sizeof(cfloat) * bf.size(), hipMemcpyHostToDevice, s));
            OMR_HIP(hipStreamSynchronize(s));  // the host vectors go out of scope
        } else {
            OMR_HIP(hipStreamSynchronize(s));
        }
        return OMR_OK;
    }
};

// The tables of an axis length depend on nothing else: built once per (device, length) and kept for the life of the
// process (a few hundred KB each), so that a call pays for them only the first time.
const AxisTables *axis_tables(int len, hipStream_t s, int *rc)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, AxisTables *> *cache = new std::map<std::pair<int, int>, AxisTables *>();
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) {  // (never another device's tables)
        *rc = fail(OMR_ERR_GPU, "hipGetDevice: %s", hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache->find({dev, len});
    if (it != cache->end()) {
        *rc = OMR_OK;
        return it->second;
    }
    AxisTables *t = new AxisTables();
    *rc = t->build(len, s);  // synchronises the stream: the tables are complete for every later user
    if (*rc) {
        delete t;
        return nullptr;
    }
    (*cache)[{dev, len}] = t;
    return t;
}

// workspace + tables for scans of one shape
struct FftWork {
    int rows = 0, cols = 0, pitch = 0;  // pitch: elements per line of the TRANSPOSED half spectrum (cols / 2 + 1 lines of
                                        // rows + 8 elements: the row pass's stores of one column index then spread over
                                        // the memory channels instead of aliasing)
    const AxisTables *axc = nullptr, *axr = nullptr;  // transforms along a row (length cols) / along a column (length rows)
    int mag_pitch = 0;  // floats per line of the transposed |F| (cols / 2 + 1 lines of rows + 16)
    DevBuf c0, mag, mm, part, bigbuf;
    int big_chunk = 0;  // lines of the global-memory chirp-z's buffer (an axis of more than 8192 points, fft_big.hip)
    int group = 1;  // scans carried by one launch of each kernel (every per-scan array holds that many)
    int create(int r, int c, hipStream_t s, int scans_per_launch = 1)
    {
        rows = r;
        cols = c;
        group = scans_per_launch < 1 ? 1 : scans_per_launch;
        int rc = OMR_OK;
        if (!(axc = axis_tables(c, s, &rc))) return rc;
        if (!(axr = axis_tables(r, s, &rc))) return rc;
        pitch = (r + 9) & ~1;  // even: a row pair's two points of a line are one aligned 16-byte store
        mag_pitch = r + 16;
        OMR_HIP(c0.alloc(sizeof(cfloat) * (size_t)(c / 2 + 1) * pitch * group));
        OMR_HIP(mag.alloc(sizeof(float) * (size_t)(c / 2 + 1) * mag_pitch * group));
        OMR_HIP(mm.alloc(sizeof(uint32_t) * 4 * group));
        OMR_HIP(part.alloc(sizeof(float) * 2 * (size_t)c * group));
        if (axc->big || axr->big) {
            const int M = std::max(axc->big ? axc->m : 0, axr->big ? axr->m : 0);
            big_chunk = (int)std::min<size_t>(((size_t)1 << 30) / (sizeof(cfloat) * (size_t)M), (size_t)std::max(r, c));
            OMR_HIP(bigbuf.alloc(sizeof(cfloat) * (size_t)M * big_chunk));
        }
        return OMR_OK;
    }
    // an axis of more than 8192 points: its pass runs scan by scan through fft_big.hip
    hipError_t big_pass(const AxisTables *ax, const FftPass &f, int z, hipStream_t s) const
    {
        BigLines b{};
        b.n = ax->n;
        b.M = ax->m;
        b.inv_m = (float)(1.0 / (double)ax->m);
        b.chunk = big_chunk;
        b.buf = bigbuf.as<cfloat>();
        b.chirp = ax->chirp.as<cfloat>();
        b.BfT = ax->Bf.as<cfloat>();
        b.wM = ax->Wfull.as<cfloat>();
        b.out_scale = f.out_scale;
        if (f.src_u8) {  // row pass
            b.src_u8 = f.src_u8 + (int64_t)z * f.src_u8_scan_stride;
            b.src_step = f.src_step;
            b.in_scale = f.in_scale;
            b.src_rows = f.src_rows;
            b.lines = (f.src_rows + 1) / 2;
            b.dst = f.dst + (int64_t)z * f.c_scan_stride;
            b.dst_pitch = f.dst_elem_stride;
        } else {
            b.src_c = f.src_c + (int64_t)z * f.c_scan_stride;
            b.line_stride = f.line_stride;
            b.lines = f.lines;
            b.mag_dst = f.mag_dst + (int64_t)z * f.mag_scan_stride;
            b.mag_pitch = f.mag_pitch;
            b.part = f.part + (int64_t)z * f.part_scan_stride;
        }
        return launch_big_lines(b, s);
    }
    // one pass over `scans` scans: the LDS kernels take them in one launch, the global-memory path one by one
    hipError_t pass(const AxisTables *ax, const FftPass &f, hipStream_t s) const
    {
        if (!ax->big) return ax->launch(f, s);
        for (int z = 0; z < f.scans; z++)
            if (hipError_t e = big_pass(ax, f, z, s); e != hipSuccess) return e;
        return hipSuccess;
    }
    // fft.rs:124-141 for one device-resident 8-bit scan -> the two 8-bit pictures (device, packed)
    // `scans` (<= group) scans, scan_stride bytes apart; pictures packed one after the other
    int run(const uint8_t *d_gray, int64_t step, uint8_t *d_mag_u8, uint8_t *d_log_u8, hipStream_t s, int scans = 1,
            int64_t scan_stride = 0)
    {
        if (scans < 1 || scans > group) return fail(OMR_ERR_BADARG, "FFT launch group of %d scans, workspace holds %d", scans, group);
        FftPass p{};
        p.scans = scans;
        p.src_u8_scan_stride = scan_stride;
        p.c_scan_stride = (int64_t)(cols / 2 + 1) * pitch;
        // along rows: u8 * (1 / 255) -> complex spectrum lines, stored transposed (column index major)
        p.src_u8 = d_gray;
        p.src_step = step;
        p.in_scale = (float)(1.0 / 255.0);  // convert_to(CV_32F, 1.0 / 255.0): alpha cast to float
        p.dst = c0.as<cfloat>();
        p.n = cols;
        p.m = axc->m;
        p.log2m = axc->log2m;
        p.lines = rows;
        p.W = axc->W.as<cfloat>();
        p.Wfull = axc->Wfull.as<cfloat>();
        p.chirp = axc->blue ? axc->chirp.as<cfloat>() : nullptr;
        p.Bf = axc->blue ? axc->Bf.as<cfloat>() : nullptr;
        p.sub = axc->sub;
        p.out_scale = 1.0f;
        p.line_stride = pitch;  // (unused: the input is the 8-bit scan)
        p.elem_stride = 1;
        p.dst_line_stride = 1;  // row r, column index k -> c0[k * pitch + r]
        p.dst_elem_stride = pitch;
        p.xcd_blocked = 1;      // neighbouring rows' 16-byte pieces of a line meet in one XCD's L2
        p.real_pairs = 1;  // two real rows per workgroup, columns 0 .. cols / 2 written
        p.src_rows = rows;
        OMR_HIP(pass(axc, p, s));
        // along columns = along the lines of the transposed array, with DFT_SCALE
        FftPass q{};
        q.scans = scans;
        q.c_scan_stride = (int64_t)(cols / 2 + 1) * pitch;
        q.mag_scan_stride = (int64_t)(cols / 2 + 1) * mag_pitch;
        q.part_scan_stride = 2 * (int64_t)cols;
        q.src_c = c0.as<cfloat>();
        q.dst = nullptr;  // spectrum-picture mode: |F|, quadrant-swapped, straight from the column pass
        q.mag_dst = mag.as<float>();
        q.mag_pitch = mag_pitch;
        q.img_rows = rows;
        q.img_cols = cols;
        q.part = part.as<float>();
        q.line_stride = pitch;
        q.elem_stride = 1;
        q.n = rows;
        q.m = axr->m;
        q.log2m = axr->log2m;
        q.lines = cols / 2 + 1;  // the other columns are mirror images (real input)
        q.half_mirror = 1;
        q.W = axr->W.as<cfloat>();
        q.Wfull = axr->Wfull.as<cfloat>();
        q.chirp = axr->blue ? axr->chirp.as<cfloat>() : nullptr;
        q.Bf = axr->blue ? axr->Bf.as<cfloat>() : nullptr;
        q.sub = axr->sub;
        q.out_scale = (float)(1.0 / ((double)rows * (double)cols));
        OMR_HIP(pass(axr, q, s));
        OMR_HIP(launch_minmax_final(part.as<float>(), cols / 2 + 1, mm.as<uint32_t>(), s, scans, 2 * (int64_t)cols));
        OMR_HIP(launch_spec_pictures(mag.as<float>(), rows, cols, mag_pitch, mm.as<uint32_t>(), d_mag_u8, d_log_u8, s, scans,
                                     (int64_t)(cols / 2 + 1) * mag_pitch));
        return OMR_OK;
    }
};

int check_gray(const omr_image *im)
{
    int rc = check_img(im);
    if (rc) return rc;
    if (im->channels != 1) return fail(OMR_ERR_ASSERT, "the FFT path takes an 8-bit single-channel image");
    return OMR_OK;
}

int download_owned(const uint8_t *d, int rows, int cols, omr_image_owned *out, hipStream_t s)
{
    out->rows = rows;
    out->cols = cols;
    out->channels = 1;
    out->step_bytes = cols;
    out->data = (uint8_t *)malloc((size_t)rows * cols);
    if (!out->data) return fail(OMR_ERR_NOMEM, "out of host memory");
    int rc = staged_d2h(out->data, d, (size_t)rows * cols, s);
    if (rc) omr_image_free(out);
    return rc;
}

}  // namespace

extern "C" {

int omr_get_fft_image(const omr_image *gray, omr_image_owned *magnitude_image, omr_image_owned *magnitude_log_image)
{
    clear_error();
    int rc = check_gray(gray);
    if (rc) return rc;
    if (!magnitude_image && !magnitude_log_image) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    DevBuf in, m8, l8;
    if ((rc = upload(gray, &in, st.s))) return rc;
    FftWork w;
    if ((rc = w.create(gray->rows, gray->cols, st.s))) return rc;
    const size_t px = (size_t)gray->rows * gray->cols;
    OMR_HIP(m8.alloc(px));
    OMR_HIP(l8.alloc(px));
    if ((rc = w.run(in.as<uint8_t>(), gray->cols, m8.as<uint8_t>(), l8.as<uint8_t>(), st.s))) return rc;
    if (magnitude_image && (rc = download_owned(m8.as<uint8_t>(), gray->rows, gray->cols, magnitude_image, st.s))) return rc;
    if (magnitude_log_image &&
        (rc = download_owned(l8.as<uint8_t>(), gray->rows, gray->cols, magnitude_log_image, st.s))) {
        if (magnitude_image) omr_image_free(magnitude_image);
        return rc;
    }
    return OMR_OK;
}

// n device-resident scans of one shape -> their magnitude_log pictures (device, packed n x rows x cols)
int omr_fft_image_batch_device(const uint8_t *d_scans, int32_t n, int64_t scan_stride_bytes, int32_t rows, int32_t cols,
                               int64_t step_bytes, uint8_t *d_magnitude_log, void *stream)
{
    clear_error();
    if (!d_scans || !d_magnitude_log || n <= 0) return fail(OMR_ERR_BADARG, "null pointer or empty batch");
    if (rows <= 0 || cols <= 0 || rows >= 32767 || cols >= 32767) return fail(OMR_ERR_ASSERT, "bad image shape");
    if (step_bytes < cols) return fail(OMR_ERR_BADARG, "step_bytes too small");
    int rc = have_device();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // several scans per launch of each kernel: the fixed cost between dependent launches (tens of microseconds, as
    // for the sweep) is paid once per group; the workspace holds <= 1 GiB of spectra
    const size_t px = (size_t)rows * cols;
    int G = (int)std::min<size_t>((size_t)n, std::max<size_t>(1, ((size_t)1 << 30) / (sizeof(cfloat) * (size_t)rows * (cols + 8))));
    G = std::min(G, 8);
    FftWork w;
    if ((rc = w.create(rows, cols, s, G))) return rc;
    for (int i = 0; i < n; i += G) {
        const int g = std::min(G, n - i);
        if ((rc = w.run(d_scans + (int64_t)i * scan_stride_bytes, step_bytes, nullptr, d_magnitude_log + (size_t)i * px, s, g,
                        scan_stride_bytes)))
            return rc;
    }
    OMR_HIP(hipStreamSynchronize(s));  // the workspace is released on return
    return OMR_OK;
}

int omr_get_angle_with_fft(const omr_image *gray, double canny_threshold_1, double canny_threshold_2,
                           double min_line_length, double max_line_gap, double *angle_out)
{
    clear_error();
    int rc = check_gray(gray);
    if (rc) return rc;
    if (!angle_out) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    const int rows = gray->rows, cols = gray->cols;
    DevBuf in, m8, l8;
    if ((rc = upload(gray, &in, st.s))) return rc;
    FftWork w;
    if ((rc = w.create(rows, cols, st.s))) return rc;
    OMR_HIP(m8.alloc((size_t)rows * cols));
    OMR_HIP(l8.alloc((size_t)rows * cols));
    if ((rc = w.run(in.as<uint8_t>(), cols, m8.as<uint8_t>(), l8.as<uint8_t>(), st.s))) return rc;
    HoughParams hp;
    hp.low = canny_threshold_1;
    hp.high = canny_threshold_2;
    hp.threshold = 100;  // fft.rs:184
    hp.min_line_length = min_line_length;
    hp.max_line_gap = max_line_gap;
    std::vector<std::vector<int32_t>> lines;
    if ((rc = edges_lines_device(l8.as<uint8_t>(), 0, cols, rows, cols, 1, 1, hp, st.s, &lines))) return rc;
    // fft.rs:197-247: f64 angles folded into [-45, 45]; the inner loop re-reads line i (quirk B10), so
    // line i collects n - 1 votes iff its raw angle is within 0.1 of its folded angle, else none
    const std::vector<int32_t> &l = lines[0];
    const int n = (int)(l.size() / 4);
    double average_angle = 0.0;
    int max_votes = 0;
    for (int i = 0; i < n; i++) {
        const double x1 = l[4 * i], y1 = l[4 * i + 1], x2 = l[4 * i + 2], y2 = l[4 * i + 3];
        const double raw = (atan2(y2 - y1, x2 - x1) * 180.0) / kPi;
        const double angle = raw < -45.0 ? raw + 90.0 : (raw > 45.0 ? raw - 90.0 : raw);
        int votes = 0;
        for (int j = 0; j < n; j++) {
            if (i == j) continue;
            if (fabs(raw - angle) < 0.1) votes++;
        }
        if (votes > max_votes) {
            max_votes = votes;
            average_angle = angle;
        }
        if (max_votes == n - 1 && n > 1) break;  // nothing can beat n - 1 with a strict '>'
    }
    *angle_out = average_angle;
    return OMR_OK;
}

int omr_get_result_from_fourier_transform(const omr_image *src, double canny_threshold_weak,
                                          double canny_threshold_strong, double fourier_min_line_length,
                                          double fourier_max_line_gap, double *angle, int32_t *status,
                                          double *candidates, int32_t cand_cap, int32_t *cand_len)
{
    clear_error();
    int rc = check_img(src);
    if (rc) return rc;
    if (src->channels != 3 && src->channels != 4)
        return fail(OMR_ERR_ASSERT, "cvtColor(RGB2GRAY) needs 3 or 4 channels (omr.rs:314)");
    if (!angle) return fail(OMR_ERR_BADARG, "null output");
    if ((rc = have_device())) return rc;
    HStream st;
    if ((rc = st.create())) return rc;
    const int rows = src->rows, cols = src->cols;
    DevBuf in, gray, m8, l8, edges, flag, rowcnt;
    if ((rc = upload(src, &in, st.s))) return rc;
    OMR_HIP(gray.alloc((size_t)rows * cols));
    OMR_HIP(launch_rgb2gray(in.as<uint8_t>(), (int64_t)cols * src->channels, rows, cols, src->channels, gray.as<uint8_t>(),
                            cols, st.s));
    FftWork w;
    if ((rc = w.create(rows, cols, st.s))) return rc;
    OMR_HIP(m8.alloc((size_t)rows * cols));
    OMR_HIP(l8.alloc((size_t)rows * cols));
    if ((rc = w.run(gray.as<uint8_t>(), cols, m8.as<uint8_t>(), l8.as<uint8_t>(), st.s))) return rc;
    // omr.rs:323-330 Canny(weak, strong) on the log picture, then get_result_from_edges_detection on the
    // EDGE picture: Canny(50, 150) once more (omr.rs:236-240), HoughLinesP threshold 0, the omr.rs vote
    OMR_HIP(edges.alloc((size_t)rows * cols));
    OMR_HIP(flag.alloc(sizeof(int)));
    OMR_HIP(rowcnt.alloc(sizeof(int32_t) * (size_t)rows));
    if ((rc = canny_device(l8.as<uint8_t>(), 0, cols, rows, cols, 1, 1, canny_threshold_weak, canny_threshold_strong,
                           edges.as<uint8_t>(), flag.as<int>(), st.s, rowcnt.as<int32_t>())))
        return rc;
    HoughParams hp;
    hp.min_line_length = fourier_min_line_length;
    hp.max_line_gap = fourier_max_line_gap;
    std::vector<std::vector<int32_t>> lines;
    if ((rc = edges_lines_device(edges.as<uint8_t>(), 0, cols, rows, cols, 1, 1, hp, st.s, &lines))) return rc;
    std::vector<float> ang;
    std::vector<int32_t> cnt;
    line_angles(lines[0], &ang);
    if ((rc = vote_counts(ang, true, st.s, &cnt))) return rc;
    return select_omr_rs(ang, cnt, angle, status, candidates, cand_cap, cand_len);
}

}  // extern "C"
