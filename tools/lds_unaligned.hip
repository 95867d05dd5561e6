// lds_unaligned.hip -- can the sweep kernel fetch its 64-bit source windows with ONE ds_read_b64 at a
// 4-byte-aligned (not 8-byte-aligned) LDS address instead of ds_read2_b32?  Checks the data and times
// both, 4 waves per SIMD, lane stride = 21 words (the kernel's odd window pitch).
// Build + run (GPU box): hipcc -O2 --offload-arch=gfx950 tools/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define REPS 2000
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>  // 0: ds_read2_b32, 1: ds_read_b64 (aligned 8), 2: ds_read_b64 at odd word, 3: ds_read_b64 mixed parity per lane
__global__ __launch_bounds__(256) void k(uint64_t *dt, uint32_t *sink, int pitch_words)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t word = (uint32_t)(lane * pitch_words) + (threadIdx.x >> 6) * 2;
    if (MODE == 1) word &= ~1u;
    if (MODE == 2) word |= 1u;
    if (MODE == 3) word = (word & ~1u) | (lane & 1);
    uint32_t addr = word * 4;
    uint32_t acc0 = 0, acc1 = 0;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < REPS; it++) {
        uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
        if (MODE == 0) {
            asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n\tds_read2_b32 %1, %4 offset0:21 offset1:22\n\t"
                         "ds_read2_b32 %2, %4 offset0:42 offset1:43\n\tds_read2_b32 %3, %4 offset0:63 offset1:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(*(uint64_t *)&a0), "=v"(*(uint64_t *)&b0), "=v"(*(uint64_t *)&c0), "=v"(*(uint64_t *)&d0) : "v"(addr) : "memory");
        }
        uint64_t A, B, C, D;
        if (MODE == 0) {
            asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n\tds_read2_b32 %1, %4 offset0:21 offset1:22\n\t"
                         "ds_read2_b32 %2, %4 offset0:42 offset1:43\n\tds_read2_b32 %3, %4 offset0:63 offset1:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(A), "=v"(B), "=v"(C), "=v"(D) : "v"(addr) : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %4 offset:0\n\tds_read_b64 %1, %4 offset:84\n\t"
                         "ds_read_b64 %2, %4 offset:168\n\tds_read_b64 %3, %4 offset:252\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(A), "=v"(B), "=v"(C), "=v"(D) : "v"(addr) : "memory");
        }
        acc0 ^= (uint32_t)A ^ (uint32_t)B ^ (uint32_t)C ^ (uint32_t)D;
        acc1 ^= (uint32_t)(A >> 32) ^ (uint32_t)(B >> 32) ^ (uint32_t)(C >> 32) ^ (uint32_t)(D >> 32);
        (void)a0; (void)a1; (void)b0; (void)b1; (void)c0; (void)c1; (void)d0; (void)d1;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // correctness: one more read, compared against plain indexing
    uint64_t X;
    asm volatile("ds_read_b64 %0, %1 offset:84\n\ts_waitcnt lgkmcnt(0)" : "=v"(X) : "v"(addr) : "memory");
    const uint32_t e0 = lds[word + 21], e1 = lds[word + 22];
    const int bad = ((uint32_t)X != e0) || ((uint32_t)(X >> 32) != e1);
    if ((threadIdx.x & 63) == 0) dt[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = (acc0 ^ acc1) * 0 + bad;
}

template <int MODE>
static void run(const char *name, int pitch)
{
    const int kk = 4, blocks = 256 * kk * 4;
    const size_t lds = 40 * 1024;
    uint64_t *d_dt; uint32_t *d_sink;
    CK(hipMalloc(&d_dt, sizeof(uint64_t) * blocks * 4));
    CK(hipMalloc(&d_sink, sizeof(uint32_t) * blocks * 256));
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_dt, d_sink, pitch);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> dt((size_t)blocks * 4);
    std::vector<uint32_t> sk((size_t)blocks * 256);
    CK(hipMemcpy(dt.data(), d_dt, sizeof(uint64_t) * dt.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(sk.data(), d_sink, sizeof(uint32_t) * sk.size(), hipMemcpyDeviceToHost));
    long bad = 0;
    for (uint32_t v : sk) bad += v;
    std::sort(dt.begin(), dt.end());
    // per CU: 4 blocks x 4 waves = 16 waves, each issuing REPS x 4 (x2 for mode 0) reads of 8 B per lane
    const double med = (double)dt[dt.size() / 2];
    const double reads = (double)REPS * 4 * (MODE == 0 ? 2 : 1);
    printf("%-44s pitch %2d words: %.2f cycles per 512-B wave-read per CU (16 waves/CU)  wrong lanes %ld\n", name, pitch,
           med / (reads * 16), bad);
    CK(hipFree(d_dt)); CK(hipFree(d_sink));
}

int main()
{
    for (int pitch : {21, 1, 2}) {
        run<0>("ds_read2_b32 (x2 per iteration)", pitch);
        run<1>("ds_read_b64, 8-byte aligned", pitch);
        run<2>("ds_read_b64, odd word (4-byte aligned only)", pitch);
        run<3>("ds_read_b64, parity mixed by lane", pitch);
    }
    return 0;
}
