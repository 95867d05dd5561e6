// LDS micro-benchmark (development aid): does ds_read_u16 / ds_read_u16_d16(_hi) take an ODD byte address, also one
// whose two bytes straddle a dword, what does it return and what does it cost next to two ds_read_u8?  And v_dot4_u32_u8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint8_t *src, uint32_t *out, int iters, int stride, uint32_t amask)
{
    __shared__ __attribute__((aligned(16))) uint8_t box[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) box[i] = src[i];
    __syncthreads();
    uint32_t acc = 0;
    uint32_t a = (threadIdx.x * (uint32_t)stride) & 8191u;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)box;
    for (int it = 0; it < iters; it++) {
        uint32_t v = 0;
        const uint32_t ad = base + (a & amask);  // amask = ~0: any byte address; ~1: even addresses only
        if (MODE == 0) {  // two byte reads per row, packed by hand
            uint32_t b0, b1, b2, b3;
            asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:1\n\tds_read_u8 %2, %4 offset:160\n\tds_read_u8 %3, %4 offset:161\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(ad));
            v = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        } else if (MODE == 1) {  // u16 reads, d16 forms: (b0, b1, b2, b3) in one dword, no vector instruction
            asm volatile("ds_read_u16_d16 %0, %1\n\tds_read_u16_d16_hi %0, %1 offset:160\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ad));
        } else {  // plain u16 reads
            uint32_t lo, hi;
            asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:160\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(ad));
            v = lo | (hi << 16);
        }
        acc += v * (uint32_t)(it + 1);
        a = (a + 7u * (uint32_t)stride + 1u) & 8191u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void dot4(const uint32_t *a, const uint32_t *b, uint32_t *o) { o[threadIdx.x] = __builtin_amdgcn_udot4(a[threadIdx.x], b[threadIdx.x], 7u, false); }

int main()
{
    uint8_t *h = (uint8_t *)malloc(16384), *d;
    for (int i = 0; i < 16384; i++) h[i] = (uint8_t)((i * 131 + (i >> 7) * 17) & 255);
    CHECK(hipMalloc(&d, 16384));
    CHECK(hipMemcpy(d, h, 16384, hipMemcpyHostToDevice));
    uint32_t *o;
    const int blocks = 2048, iters = 2000;
    CHECK(hipMalloc(&o, blocks * 256 * 4));
    uint32_t *ho = (uint32_t *)malloc(blocks * 256 * 4);
    for (int pass = 0; pass < 5; pass++) {
        const int stride = pass < 4 ? pass + 1 : 2;
        const uint32_t amask = pass < 4 ? ~0u : ~1u;
        uint32_t ref[256];
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters, stride, amask);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters, stride, amask);
                else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters, stride, amask);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            CHECK(hipMemcpy(ho, o, blocks * 256 * 4, hipMemcpyDeviceToHost));
            int wrong = 0;
            if (mode == 0) memcpy(ref, ho, sizeof ref);
            else for (int i = 0; i < 256; i++) wrong += ho[i] != ref[i];
            const double per = ms * 1e-3 * 2.4e9 / ((double)iters * blocks * 4 / 1024.0);  // cycles per wave-iteration per SIMD
            printf("%slane stride %d bytes, %s: %.3f ms, %.1f cycles per wave and tap pair on a SIMD, wrong lanes %d\n", amask == ~1u ? "EVEN addresses only, " : "", stride,
                   mode == 0 ? "4 x ds_read_u8 + 3 pack ops" : mode == 1 ? "ds_read_u16_d16 + _d16_hi (SRAM ECC: the other half is zeroed)" : "2 x ds_read_u16 + 1 pack op    ", ms, per, wrong);
        }
    }
    uint32_t ha[64], hb[64], *da, *db, *dd;
    for (int i = 0; i < 64; i++) ha[i] = 0x01020304u * (i + 1), hb[i] = 0x20100804u + i;
    CHECK(hipMalloc(&da, 256)); CHECK(hipMalloc(&db, 256)); CHECK(hipMalloc(&dd, 256));
    CHECK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dot4, dim3(1), dim3(64), 0, 0, da, db, dd);
    uint32_t hd[64];
    CHECK(hipMemcpy(hd, dd, 256, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        uint32_t e = 7;
        for (int k = 0; k < 4; k++) e += ((ha[i] >> (8 * k)) & 255u) * ((hb[i] >> (8 * k)) & 255u);
        bad += e != hd[i];
    }
    printf("v_dot4_u32_u8: %d wrong of 64\n", bad);
    return 0;
}
