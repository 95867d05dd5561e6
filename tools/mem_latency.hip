// mem_latency.hip -- what one dependent round trip costs a single wave on gfx950 (the Hough stage's unit of time).
// One wave; per iteration every lane loads K words from pseudo-random, distinct cache lines of a region of
// `bytes`, optionally stores them back (+1) or decrements them with no-return atomics, and the next iteration's
// addresses depend on the loaded values.  Prints shader cycles per iteration (s_memtime).
// Build: hipcc -O3 --offload-arch=gfx950 -o mem_latency tools/mem_latency.hip ; run: ./mem_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int K, int MODE, int SCOPE>  // MODE 0: loads; 1: load + store; 2: load + no-return atomic; 3: atomic only (no loads)
__global__ __launch_bounds__(64) void chase(int32_t *buf, uint32_t mask_words, int iters, unsigned long long *out)
{
    const int lane = threadIdx.x;
    uint32_t x = lane * 2654435761u + 12345u;
    unsigned long long t0, t1;
    int acc = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
        int32_t *p[K];
        int v[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            x = x * 1664525u + 1013904223u;
            p[k] = buf + (((x >> 4) + acc) & mask_words & ~31u) + (lane & 31);  // a line of its own (mostly)
        }
        if (MODE != 3) {
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = __hip_atomic_load(p[k], __ATOMIC_RELAXED, SCOPE);
#pragma unroll
            for (int k = 0; k < K; k++) acc += v[k] & 1;  // zeros in memory: acc stays 0, but the chain is real
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < K; k++) __hip_atomic_store(p[k], v[k], __ATOMIC_RELAXED, SCOPE);
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int k = 0; k < K; k++) __hip_atomic_fetch_sub(p[k], 0, __ATOMIC_RELAXED, SCOPE);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) {
        out[0] = t1 - t0;
        out[1] = (unsigned long long)acc;
    }
}

// no-return atomics, nothing waits: how the 64 lanes' addresses are spread.  SPREAD 0: a line of its own per lane;
// 1: consecutive words (two lines per instruction); 2: one word for all lanes; 3: consecutive words, every 4th lane active
template <int SPREAD>
__global__ __launch_bounds__(64) void atomics(int32_t *buf, uint32_t mask_words, int iters, unsigned long long *out)
{
    const int lane = threadIdx.x;
    uint32_t x = 777u;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t base = (x >> 4) & mask_words & ~2047u;
        int32_t *p = buf + base + (SPREAD == 0 ? lane * 32 : SPREAD == 2 ? 0 : lane);
        if (SPREAD != 3 || (lane & 3) == 0) __hip_atomic_fetch_sub(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[0] = t1 - t0;
}
template <int SPREAD>
static void run_atomics(const char *name, int32_t *d, size_t bytes, unsigned long long *dout)
{
    const int iters = 20000;
    atomics<SPREAD><<<1, 64>>>(d, (uint32_t)(bytes / 4 - 1), iters, dout);
    unsigned long long h[2];
    hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    printf("%-60s %8.1f cycles / instruction\n", name, (double)h[0] / iters);
}

template <int K, int MODE, int SCOPE>
static void run(const char *name, int32_t *d, size_t bytes, unsigned long long *dout)
{
    const int iters = 20000;
    const uint32_t mw = (uint32_t)(bytes / 4 - 1);
    chase<K, MODE, SCOPE><<<1, 64>>>(d, mw, 2000, dout);
    chase<K, MODE, SCOPE><<<1, 64>>>(d, mw, iters, dout);
    unsigned long long h[2];
    hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f MB  K=%d  %8.1f cycles / iteration\n", name, bytes / 1048576.0, K, (double)h[0] / iters);
}

int main()
{
    int32_t *d;
    unsigned long long *dout;
    const size_t maxb = 64u << 20;
    hipMalloc(&d, maxb);
    hipMemset(d, 0, maxb);
    hipMalloc(&dout, 16);
    run_atomics<0>("no-return atomic, 64 lanes, a line each", d, 2u << 20, dout);
    run_atomics<1>("no-return atomic, 64 lanes, consecutive words", d, 2u << 20, dout);
    run_atomics<2>("no-return atomic, 64 lanes, one word", d, 2u << 20, dout);
    run_atomics<3>("no-return atomic, 16 lanes, consecutive words", d, 2u << 20, dout);
    const size_t sizes[] = {16u << 10, 2u << 20, 64u << 20};
    for (size_t b : sizes) {
        run<1, 0, __HIP_MEMORY_SCOPE_WORKGROUP>("load, workgroup scope", d, b, dout);
        run<1, 0, __HIP_MEMORY_SCOPE_AGENT>("load, agent scope (sc1)", d, b, dout);
        run<3, 0, __HIP_MEMORY_SCOPE_WORKGROUP>("3 loads in flight, workgroup scope", d, b, dout);
        run<3, 1, __HIP_MEMORY_SCOPE_WORKGROUP>("3 loads + 3 stores, workgroup scope", d, b, dout);
        run<3, 2, __HIP_MEMORY_SCOPE_WORKGROUP>("3 loads + 3 no-return atomics", d, b, dout);
        run<3, 3, __HIP_MEMORY_SCOPE_WORKGROUP>("3 no-return atomics, nothing waits", d, b, dout);
    }
    return 0;
}
