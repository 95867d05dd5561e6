// valu_issue.hip -- how many cycles does one wave64 VALU instruction of the sweep kernel's mix hold a
// gfx950 SIMD for, at 1 / 2 / 4 / 8 resident waves per SIMD?  (Round-1 verdict, item 1: DESIGN.md
// priced it at 4 cycles, MI355X_MICROARCH.md's constants table says 2 with >= 2 waves per SIMD.)
//
// Method: 256-thread workgroups (one wave per SIMD each), k workgroups per CU forced by the dynamic
// LDS size (160 KB / k), grid = 256 CUs x k x 4 rounds.  Every wave runs REPS iterations of a 64-
// instruction straight-line block made of 8 independent dependency chains and stamps s_memtime
// around the loop; cycles per instruction per SIMD = median(dt) / (REPS * 64 * k).  Wall-clock
// throughput from HIP events is printed beside it.
//
// Build + run (GPU box):  hipcc -O2 --offload-arch=gfx950 tools/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define REPS 2000

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// 8 chains x 8 instructions; MIX selects the opcodes
#define CHAIN8(OP)                                                                                         \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

// mix 0: v_add_u32 (the table's reference instruction class)
#define ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
// mix 1: the merge loop's opcodes -- v_alignbit_b32, v_bfi_b32, v_and_or_b32, v_bcnt_u32_b32, v_xor, v_and, v_lshlrev, v_or
#define ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n\t"
#define BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n\t"
#define ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n\t"
#define XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define AND(i) "v_and_b32 %" #i ", %" #i ", %9\n\t"
#define SHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define OR(i) "v_or_b32 %" #i ", %" #i ", %8\n\t"
// mix 2: v_add_co_u32 with an SGPR-pair carry-out (the column flush) -- VOP3 with scalar destination
#define ADDCO(i) "v_add_co_u32_e64 %" #i ", s[20:21], %" #i ", %" #i "\n\t"

template <int MIX>
__global__ __launch_bounds__(256) void issue_kernel(uint64_t *__restrict__ dt, uint32_t *__restrict__ sink, uint32_t seed)
{
    extern __shared__ char lds[];
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17,
             r7 = r0 * 19;
    const uint32_t ka = seed | 0x01010101u, kb = ~seed;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < REPS; it++) {
        if (MIX == 0) {
            asm volatile(CHAIN8(ADD) CHAIN8(ADD) CHAIN8(ADD) CHAIN8(ADD) CHAIN8(ADD) CHAIN8(ADD) CHAIN8(ADD) CHAIN8(ADD)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(ka), "v"(kb));
        } else if (MIX == 1) {
            asm volatile(CHAIN8(ALIGNBIT) CHAIN8(BFI) CHAIN8(ANDOR) CHAIN8(BCNT) CHAIN8(XOR) CHAIN8(AND) CHAIN8(SHL) CHAIN8(OR)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(ka), "v"(kb));
        } else {
            asm volatile(CHAIN8(ADDCO) CHAIN8(ADDCO) CHAIN8(ADDCO) CHAIN8(ADDCO) CHAIN8(ADDCO) CHAIN8(ADDCO) CHAIN8(ADDCO)
                             CHAIN8(ADDCO)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(ka), "v"(kb)
                         : "s20", "s21");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) dt[wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if (seed == 0xdeadbeefu) lds[threadIdx.x] = (char)r0;  // keep the LDS allocation alive
}

template <int MIX>
static void run(const char *name, int k)
{
    const int blocks = 256 * k * 4;
    const size_t lds = (160 * 1024 / k) & ~255u;
    uint64_t *d_dt;
    uint32_t *d_sink;
    CK(hipMalloc(&d_dt, sizeof(uint64_t) * blocks * 4));
    CK(hipMalloc(&d_sink, sizeof(uint32_t) * blocks * 256));
    CK(hipFuncSetAttribute((const void *)issue_kernel<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(issue_kernel<MIX>, dim3(blocks), dim3(256), lds, 0, d_dt, d_sink, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(issue_kernel<MIX>, dim3(blocks), dim3(256), lds, 0, d_dt, d_sink, 12345u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> dt((size_t)blocks * 4);
    CK(hipMemcpy(dt.data(), d_dt, sizeof(uint64_t) * dt.size(), hipMemcpyDeviceToHost));
    std::sort(dt.begin(), dt.end());
    const double med = (double)dt[dt.size() / 2], n_inst = (double)REPS * 64;
    // two views: s_memtime (tick = shader cycle, MI355X_MICROARCH.md constants table) inside the waves, and
    // the wall clock over the whole grid converted at the 2.4 GHz peak clock (reads high if the chip clocks lower)
    const double wave_instr = (double)blocks * 4 * n_inst;
    const double per_simd_ns = ms * 1e6 / (wave_instr / 1024.0);  // ns of SIMD time per wave-instruction
    printf("%-26s k=%d waves/SIMD  s_memtime: %.2f cycles per wave-instr per SIMD (%.2f per instr per wave)  |  wall %.3f ms: "
           "%.3f ns per wave-instr per SIMD = %.2f cycles @2.4GHz\n",
           name, k, med / n_inst / k, med / n_inst, ms, per_simd_ns, per_simd_ns * 2.4);
    CK(hipFree(d_dt));
    CK(hipFree(d_sink));
}

int main()
{
    const int ks[] = {1, 2, 4, 8};
    for (int k : ks) run<0>("v_add_u32", k);
    for (int k : ks) run<1>("sweep-kernel mix", k);
    for (int k : ks) run<2>("v_add_co_u32 (sgpr cout)", k);
    return 0;
}
