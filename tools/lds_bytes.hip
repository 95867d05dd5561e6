// lds_bytes.hip -- do sub-dword LDS reads (ds_read_u8) conflict when lanes read DIFFERENT bytes of the
// SAME dword?  The sweep kernel's fraction -> tuple look-up is exactly that pattern (4 lanes per dword).
// Build + run (GPU box): hipcc -O2 --offload-arch=gfx950 tools/lds_bytes.hip -o /tmp/lds_bytes && /tmp/lds_bytes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define REPS 2000
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: u8, lane reads byte `lane * stride`; MODE 1: b32 at (lane*stride)&~3; MODE 2: b128 at 32*(lane/div)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *dt, uint32_t *sink, int stride, int div)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t addr;
    if (MODE == 0) addr = (uint32_t)(lane * stride) & 1023u;
    else if (MODE == 1) addr = ((uint32_t)(lane * stride) & 1023u) & ~3u;
    else addr = 32u * (uint32_t)(lane / div);
    addr += (threadIdx.x >> 6) * 4096;
    uint32_t acc = 0;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < REPS; it++) {
        uint32_t a, b, c, d;
        if (MODE == 0)
            asm volatile("ds_read_u8 %0, %4 offset:0\n\tds_read_u8 %1, %4 offset:1024\n\tds_read_u8 %2, %4 offset:2048\n\t"
                         "ds_read_u8 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
        else if (MODE == 1)
            asm volatile("ds_read_b32 %0, %4 offset:0\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:2048\n\t"
                         "ds_read_b32 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
        else {
            uint4 A, B;
            asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=v"(A), "=v"(B) : "v"(addr) : "memory");
            a = A.x ^ A.y; b = A.z ^ A.w; c = B.x ^ B.y; d = B.z ^ B.w;
        }
        acc ^= a ^ b ^ c ^ d;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) dt[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *name, int stride, int div)
{
    const int kk = 4, blocks = 256 * kk * 4;
    const size_t lds = 40 * 1024;
    uint64_t *d_dt; uint32_t *d_sink;
    CK(hipMalloc(&d_dt, sizeof(uint64_t) * blocks * 4));
    CK(hipMalloc(&d_sink, sizeof(uint32_t) * blocks * 256));
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_dt, d_sink, stride, div);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> dt((size_t)blocks * 4);
    CK(hipMemcpy(dt.data(), d_dt, sizeof(uint64_t) * dt.size(), hipMemcpyDeviceToHost));
    std::sort(dt.begin(), dt.end());
    const double med = (double)dt[dt.size() / 2];
    const double reads = (double)REPS * (MODE == 2 ? 2 : 4);
    printf("%-40s %.2f cycles per wave-instruction per CU (16 waves/CU)\n", name, med / (reads * 16));
    CK(hipFree(d_dt)); CK(hipFree(d_sink));
}

int main()
{
    run<0>("ds_read_u8, all lanes same byte", 0, 1);
    run<0>("ds_read_u8, lane stride 1 byte", 1, 1);
    run<0>("ds_read_u8, lane stride 4 bytes", 4, 1);
    run<0>("ds_read_u8, lane stride 16 bytes (4 dwords)", 16, 1);
    run<0>("ds_read_u8, lane stride 89 bytes (random-ish)", 89, 1);
    run<1>("ds_read_b32, lane stride 1 byte (&~3)", 1, 1);
    run<1>("ds_read_b32, lane stride 4 bytes", 4, 1);
    run<1>("ds_read_b32, lane stride 89 bytes (&~3)", 89, 1);
    run<2>("ds_read_b128, all lanes same tuple", 0, 64);
    run<2>("ds_read_b128, tuple = lane/8 (8 distinct)", 0, 8);
    run<2>("ds_read_b128, tuple = lane/2 (32 distinct)", 0, 2);
    run<2>("ds_read_b128, tuple = lane (64 distinct)", 0, 1);
    return 0;
}
