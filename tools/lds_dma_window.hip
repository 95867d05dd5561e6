// lds_dma_window.hip -- probe for the sweep kernel's window fetch by LDS-DMA (gfx950).
//  1. data check: buffer_load_dword ... lds with per-lane offsets, lanes outside [0, num_records) must land as zeros
//  2. rate: every workgroup (512 threads, 2 per CU) fills a 552-row x 9-word window (row pitch 9 words in LDS, rows
//     80 words apart in a 1.1 MB bit image, L2-resident) again and again:
//       mode 0: dword DMA, 10 rounds of 504 lanes (the odd-pitch layout the sweep wants)
//       mode 1: dwordx4 DMA into a 12-word pitch (3 pieces per row) -- the upper bound, unusable pitch
//       mode 2: global_load_dwordx3 + 3 ds_write_b32 (register staging)
// Build + run (GPU box): hipcc -O2 --offload-arch=gfx950 tools/lds_dma_window.hip -o /tmp/lds_dma_window && /tmp/lds_dma_window
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// raw buffer descriptor (stride 0, no swizzle): base, num_records bytes, DATA_FORMAT = 32 (0x00020000)
__device__ __forceinline__ u32x4 make_rsrc(const void *base, uint32_t nbytes)
{
    const uint64_t b = (uint64_t)base;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(nbytes);
    r.w = 0x00020000u;
    return r;
}

__device__ __forceinline__ void dma_b32(u32x4 rsrc, uint32_t voff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ void dma_b128(u32x4 rsrc, uint32_t voff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

__global__ void check_kernel(const uint32_t *src, int nbytes, const int *offs, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const u32x4 r = make_rsrc(src, (uint32_t)nbytes);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    dma_b32(r, (uint32_t)offs[threadIdx.x], (uint32_t)(wave * 256 + 4));  // base 4: not 16-B aligned
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}

#define ROWS 552
#define WPR 80
template <int MODE>
__global__ __launch_bounds__(512, 1) void fill_kernel(const uint32_t *img, int img_rows, int reps, uint32_t *sink, long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32x4 rsrc = make_rsrc(img, (uint32_t)(img_rows * WPR * 4));
    uint32_t acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < reps; it++) {
        const int y0 = (blockIdx.x * 37 + it * 131) % (img_rows - ROWS), x0 = (blockIdx.x + it) % 60;
        if (MODE == 0) {
            if (tid < 504) {
                const int row = (tid * 7282) >> 16, col = tid - 9 * row;
                uint32_t voff = (uint32_t)(((y0 + row) * WPR + x0 + col) * 4);
#pragma unroll
                for (int n = 0; n < 10; n++) {
                    if (n * 56 < ROWS - 55 || row + n * 56 < ROWS) dma_b32(rsrc, voff, (uint32_t)((n * 504 + wave * 64) * 4));
                    voff += 56 * WPR * 4;
                }
            }
        } else if (MODE == 1) {
            if (tid < 510) {
                const int row = (tid * 21846) >> 16, col = tid - 3 * row;
                uint32_t voff = (uint32_t)(((y0 + row) * WPR + (x0 & ~3) + 4 * col) * 4);
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    if (row + n * 170 < ROWS) dma_b128(rsrc, voff, (uint32_t)((n * 510 + wave * 64) * 16));
                    voff += 170 * WPR * 4;
                }
            }
        } else {
            if (tid < 510) {
                const int row = (tid * 21846) >> 16, col = tid - 3 * row;
                const uint32_t *p = img + (y0 + row) * WPR + x0 + 3 * col;
                u32x3 v[4];
#pragma unroll
                for (int n = 0; n < 4; n++)
                    if (row + n * 170 < ROWS) v[n] = *(const u32x3 *)(p + n * 170 * WPR);
#pragma unroll
                for (int n = 0; n < 4; n++)
                    if (row + n * 170 < ROWS) {
                        uint32_t *d = lds + (row + n * 170) * 9 + 3 * col;
                        d[0] = v[n].x; d[1] = v[n].y; d[2] = v[n].z;
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += lds[(tid * 9 + it) % (ROWS * 9)];
        __syncthreads();
    }
    const long long t1 = clock64();
    sink[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void rate(const uint32_t *d_img, int img_rows, uint32_t *d_sink, long long *d_clk, const char *name, int lds_bytes)
{
    const int blocks = 512, reps = 400;
    CK(hipFuncSetAttribute((const void *)fill_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80000));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(blocks), dim3(512), 80000, 0, d_img, img_rows, 20, d_sink, d_clk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(blocks), dim3(512), 80000, 0, d_img, img_rows, reps, d_sink, d_clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * reps * ROWS * 9 * 4;
    printf("%-44s %7.3f ms  %6.2f TB/s of window bytes  (%.2f us per window per block, 2 blocks/CU)\n", name, ms, bytes / ms / 1e9,
           ms * 1e3 / reps / 2);
    (void)lds_bytes;
}

int main()
{
    const int N = 4096;
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; i++) h[i] = 0x1000000u + i;
    uint32_t *d, *o;
    int *doff;
    CK(hipMalloc(&d, N * 4)); CK(hipMalloc(&o, 4096)); CK(hipMalloc(&doff, 128 * 4));
    CK(hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<int> off(128);
    for (int t = 0; t < 128; t++) off[t] = 4 * (t * 5 + 1);
    off[3] = -8; off[7] = (int)0x80000000u; off[9] = 4000; off[70] = 3996; off[71] = 4004; off[100] = -4;
    CK(hipMemcpy(doff, off.data(), 128 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(128), 4096, 0, d, 4000, doff, o);
    std::vector<uint32_t> r(1024);
    CK(hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 128; t++) {
        const uint32_t got = r[(t >> 6) * 64 + 1 + (t & 63)];
        const uint32_t want = (off[t] >= 0 && off[t] + 4 <= 4000) ? h[off[t] / 4] : 0;
        if (got != want) { bad++; printf("  t=%d off=%d got=%08x want=%08x\n", t, off[t], got, want); }
    }
    printf("dword lds-dma, lds base 4 mod 16, out-of-range lanes expected as zeros: %d mismatches; lds[0]=%08x lds[65]=%08x\n", bad, r[0], r[65]);

    const int img_rows = 3508;
    std::vector<uint32_t> img((size_t)img_rows * WPR);
    for (size_t i = 0; i < img.size(); i++) img[i] = (uint32_t)(i * 2654435761u);
    uint32_t *d_img, *d_sink;
    long long *d_clk;
    CK(hipMalloc(&d_img, img.size() * 4)); CK(hipMalloc(&d_sink, 512 * 512 * 4)); CK(hipMalloc(&d_clk, 512 * 8));
    CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    rate<0>(d_img, img_rows, d_sink, d_clk, "dword DMA, pitch 9 (10 rounds x 504 lanes)", 0);
    rate<1>(d_img, img_rows, d_sink, d_clk, "dwordx4 DMA, pitch 12 (4 rounds x 510 lanes)", 0);
    rate<2>(d_img, img_rows, d_sink, d_clk, "dwordx3 loads + 3 ds_write_b32, pitch 9", 0);
    rate<0>(d_img, img_rows, d_sink, d_clk, "dword DMA again", 0);
    return 0;
}
