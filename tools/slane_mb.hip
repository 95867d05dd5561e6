// slane_mb.hip -- micro-benchmark behind the scan-lane sweep: can a wave keep a ring of
// source words in VGPRs and address it with a wave-uniform DYNAMIC index (gfx9 VGPR index mode, M0), with the
// shift and the mask of every segment coming from SGPRs filled by s_load_dwordx16?
//   variant 0: v_lshrrev_b64 with SRC1_REL (64-bit pair at ring[idx], ring[idx + 1]; odd idx included), v_and_or inside
//              the mode region (its VGPRs sit in src0 / src2, its src1 is the SGPR mask)
//   variant 1: v_alignbit_b32 with SRC0_REL | SRC1_REL, all alignbits of a word first, mode off, then the and_ors
//   variant 2: as 1 with an s_nop 0 after every index change (hazard probe: results must not differ)
//   variant 3: the LDS form: v_add_u32 (address) + ds_read2_b32 + v_alignbit + v_and_or per segment
//   variant 4: commit probe: v_mov_b32 with DST_REL into ring[idx] before every word
// Every variant is checked against the host on every lane.
// Build + run (GPU box): hipcc -O2 --offload-arch=gfx950 tools/slane_mb.hip -o /tmp/slane_mb && /tmp/slane_mb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define RING 64   // ring registers v[64 .. 127]
#define NSEG 8

// one word = 16 dwords: (mask, pk) x 8, pk = idx | sh << 8
#define L(i, o) "global_load_dword v" #i ", %[voff], s[20:21] offset:" #o "\n\t"
#define LOADRING                                                                                                             \
    "s_mov_b64 s[20:21], %[src]\n\t"                                                                                         \
    L(64, 0) L(65, 256) L(66, 512) L(67, 768) L(68, 1024) L(69, 1280) L(70, 1536) L(71, 1792) L(72, 2048) L(73, 2304)         \
    L(74, 2560) L(75, 2816) L(76, 3072) L(77, 3328) L(78, 3584) L(79, 3840)                                                   \
    "s_add_u32 s20, s20, 4096\n\ts_addc_u32 s21, s21, 0\n\t"                                                                  \
    L(80, 0) L(81, 256) L(82, 512) L(83, 768) L(84, 1024) L(85, 1280) L(86, 1536) L(87, 1792) L(88, 2048) L(89, 2304)         \
    L(90, 2560) L(91, 2816) L(92, 3072) L(93, 3328) L(94, 3584) L(95, 3840)                                                   \
    "s_add_u32 s20, s20, 4096\n\ts_addc_u32 s21, s21, 0\n\t"                                                                  \
    L(96, 0) L(97, 256) L(98, 512) L(99, 768) L(100, 1024) L(101, 1280) L(102, 1536) L(103, 1792) L(104, 2048) L(105, 2304)   \
    L(106, 2560) L(107, 2816) L(108, 3072) L(109, 3328) L(110, 3584) L(111, 3840)                                             \
    "s_add_u32 s20, s20, 4096\n\ts_addc_u32 s21, s21, 0\n\t"                                                                  \
    L(112, 0) L(113, 256) L(114, 512) L(115, 768) L(116, 1024) L(117, 1280) L(118, 1536) L(119, 1792) L(120, 2048)            \
    L(121, 2304) L(122, 2560) L(123, 2816) L(124, 3072) L(125, 3328) L(126, 3584) L(127, 3840)                                \
    "s_waitcnt vmcnt(0)\n\t"

// segment i of variant 0: mask s[36 + 2 i], pk s[37 + 2 i]
#define SEG0(m, p, first)                                       \
    "s_set_gpr_idx_idx " p "\n\t"                               \
    "s_lshr_b32 s52, " p ", 8\n\t"                              \
    "v_lshrrev_b64 v[4:5], s52, v[64:65]\n\t" first(m)
#define FIRST0(m) "v_and_or_b32 v2, v4, " m ", 0\n\t"
#define NEXT0(m) "v_and_or_b32 v2, v4, " m ", v2\n\t"

#define AL1(p, w, nop)                                          \
    "s_set_gpr_idx_idx " p "\n\t" nop                           \
    "s_lshr_b32 s52, " p ", 8\n\t"                              \
    "v_alignbit_b32 " w ", v65, v64, s52\n\t"

template <int V>
__global__ __launch_bounds__(256) void mb_kernel(const uint32_t *__restrict__ prog, const uint32_t *__restrict__ src,
                                                 uint32_t *__restrict__ out, uint64_t *__restrict__ dt, int nwords)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane4 = (threadIdx.x & 63) * 4;
    uint32_t chk;
    unsigned long long cyc;
    if (V == 3) {  // the ring of this wave in LDS: [64 entries][64 lanes]
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int i = 0; i < RING; i++) lds[(wave * RING + i) * 64 + lane] = src[i * 64 + lane];
        __syncthreads();
    }
    const uint32_t ldsbase = (threadIdx.x >> 6) * RING * 256 + lane4;
    if (V == 0) {
        asm volatile(LOADRING
                     "s_mov_b64 s[22:23], %[prog]\n\t"
                     "s_mov_b32 s24, %[n]\n\t"
                     "v_mov_b32 v3, 0\n\t"
                     "s_memtime s[26:27]\n\t"
                     "s_load_dwordx16 s[36:51], s[22:23], 0\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "1:\n\t"
                     "s_add_u32 s22, s22, 64\n\ts_addc_u32 s23, s23, 0\n\t"
                     "s_set_gpr_idx_on s37, gpr_idx(SRC1)\n\t"
                     SEG0("s36", "s37", FIRST0) SEG0("s38", "s39", NEXT0) SEG0("s40", "s41", NEXT0) SEG0("s42", "s43", NEXT0)
                     SEG0("s44", "s45", NEXT0) SEG0("s46", "s47", NEXT0) SEG0("s48", "s49", NEXT0) SEG0("s50", "s51", NEXT0)
                     "s_set_gpr_idx_off\n\t"
                     "s_load_dwordx16 s[36:51], s[22:23], 0\n\t"
                     "v_xor_b32 v3, v3, v2\n\t"
                     "v_alignbit_b32 v3, v3, v3, 1\n\t"
                     "s_sub_u32 s24, s24, 1\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_cmp_lg_u32 s24, 0\n\t"
                     "s_cbranch_scc1 1b\n\t"
                     "s_memtime s[28:29]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_sub_u32 %[cyc0], s28, s26\n\ts_subb_u32 %[cyc1], s29, s27\n\t"
                     "v_mov_b32 %[chk], v3\n\t"
                     : [chk] "=v"(chk), [cyc0] "=s"(((uint32_t *)&cyc)[0]), [cyc1] "=s"(((uint32_t *)&cyc)[1])
                     : [voff] "v"(lane4), [src] "s"(src), [prog] "s"(prog), [n] "s"(nwords)
                     : "memory", "scc", "m0", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "s20",
                       "s21", "s22", "s23", "s24", "s26", "s27", "s28", "s29", "s36", "s37", "s38", "s39", "s40", "s41", "s42",
                       "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "v64", "v65", "v66", "v67", "v68",
                       "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83",
                       "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",
                       "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
                       "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",
                       "v125", "v126", "v127");
    } else if (V == 1 || V == 2 || V == 4) {
#define NOPX "s_nop 0\n\t"
#define BODY1(nop, commit)                                                                                                  \
    asm volatile(LOADRING                                                                                                    \
                 "s_mov_b64 s[22:23], %[prog]\n\t"                                                                            \
                 "s_mov_b32 s24, %[n]\n\t"                                                                                    \
                 "v_mov_b32 v3, 0\n\t"                                                                                        \
                 "s_memtime s[26:27]\n\t"                                                                                     \
                 "s_load_dwordx16 s[36:51], s[22:23], 0\n\t"                                                                  \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
                 "1:\n\t"                                                                                                     \
                 "s_add_u32 s22, s22, 64\n\ts_addc_u32 s23, s23, 0\n\t" commit                                                \
                 "s_set_gpr_idx_on s37, gpr_idx(SRC0,SRC1)\n\t" nop                                                           \
                 AL1("s37", "v4", nop) AL1("s39", "v5", nop) AL1("s41", "v6", nop) AL1("s43", "v7", nop)                        \
                 AL1("s45", "v8", nop) AL1("s47", "v9", nop) AL1("s49", "v10", nop) AL1("s51", "v11", nop)                      \
                 "s_set_gpr_idx_off\n\t"                                                                                      \
                 "v_and_b32 v2, s36, v4\n\t"                                                                                  \
                 "v_and_or_b32 v2, v5, s38, v2\n\t"                                                                           \
                 "v_and_or_b32 v2, v6, s40, v2\n\t"                                                                           \
                 "v_and_or_b32 v2, v7, s42, v2\n\t"                                                                           \
                 "v_and_or_b32 v2, v8, s44, v2\n\t"                                                                           \
                 "v_and_or_b32 v2, v9, s46, v2\n\t"                                                                           \
                 "v_and_or_b32 v2, v10, s48, v2\n\t"                                                                          \
                 "v_and_or_b32 v2, v11, s50, v2\n\t"                                                                          \
                 "s_load_dwordx16 s[36:51], s[22:23], 0\n\t"                                                                  \
                 "v_xor_b32 v3, v3, v2\n\t"                                                                                   \
                 "v_alignbit_b32 v3, v3, v3, 1\n\t"                                                                           \
                 "s_sub_u32 s24, s24, 1\n\t"                                                                                  \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
                 "s_cmp_lg_u32 s24, 0\n\t"                                                                                    \
                 "s_cbranch_scc1 1b\n\t"                                                                                      \
                 "s_memtime s[28:29]\n\t"                                                                                     \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
                 "s_sub_u32 %[cyc0], s28, s26\n\ts_subb_u32 %[cyc1], s29, s27\n\t"                                            \
                 "v_mov_b32 %[chk], v3\n\t"                                                                                   \
                 : [chk] "=v"(chk), [cyc0] "=s"(((uint32_t *)&cyc)[0]), [cyc1] "=s"(((uint32_t *)&cyc)[1])                    \
                 : [voff] "v"(lane4), [src] "s"(src), [prog] "s"(prog), [n] "s"(nwords)                                       \
                 : "memory", "scc", "m0", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "s20",  \
                   "s21", "s22", "s23", "s24", "s26", "s27", "s28", "s29", "s36", "s37", "s38", "s39", "s40", "s41", "s42",   \
                   "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "v64", "v65", "v66", "v67", "v68",   \
                   "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83",   \
                   "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",   \
                   "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",     \
                   "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",    \
                   "v125", "v126", "v127")
        // commit probe: ring[idx of segment 0] ^= mask of segment 7 (a VALU write through DST_REL before the word)
#define COMMIT                                                  \
    "v_xor_b32 v12, s50, v3\n\t"                               \
    "s_set_gpr_idx_on s37, gpr_idx(DST)\n\t"                    \
    "v_mov_b32 v64, v12\n\t"                                    \
    "s_set_gpr_idx_off\n\t"
        if (V == 1) BODY1("", "");
        if (V == 2) BODY1(NOPX, "");
        if (V == 4) BODY1("", COMMIT);
    } else {
        // LDS form, compiler-scheduled: the reference point for "what C++ gives"
        uint32_t acc = 0;
        const uint32_t *__restrict__ p = prog;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int w = 0; w < nwords; w++) {
            uint32_t D = 0;
#pragma unroll
            for (int i = 0; i < NSEG; i++) {
                const uint32_t mask = __builtin_amdgcn_readfirstlane(p[w * 16 + 2 * i]);
                const uint32_t pk = __builtin_amdgcn_readfirstlane(p[w * 16 + 2 * i + 1]);
                const uint32_t a = ldsbase + (pk & 255u) * 256u;
                const uint32_t lo = *(const uint32_t *)((const char *)lds + a), hi = *(const uint32_t *)((const char *)lds + a + 256);
                D |= __builtin_amdgcn_alignbit(hi, lo, pk >> 8) & mask;
            }
            acc ^= D;
            acc = __builtin_amdgcn_alignbit(acc, acc, 1);
        }
        cyc = __builtin_readcyclecounter() - t0;
        chk = acc;
    }
    out[blockIdx.x * 256 + threadIdx.x] = chk;
    if ((threadIdx.x & 63) == 0) dt[(blockIdx.x * 256 + threadIdx.x) >> 6] = cyc;
}

static uint32_t rnd(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

template <int V>
static void run(const char *name, int nwords, int wg_per_cu)
{
    std::vector<uint32_t> src((RING + 1) * 64), prog((size_t)(nwords + 1) * 16);
    uint32_t s = 12345u + V;
    for (auto &x : src) x = rnd(s) * 2654435761u ^ rnd(s);
    for (int w = 0; w <= nwords; w++)
        for (int i = 0; i < NSEG; i++) {
            const uint32_t lo = (rnd(s) % 29), len = 1 + rnd(s) % 4;
            prog[(size_t)w * 16 + 2 * i] = ((len >= 32 ? 0xffffffffu : ((1u << len) - 1u)) << lo);
            const uint32_t idx = rnd(s) % (RING - 1), sh = rnd(s) % 32;
            prog[(size_t)w * 16 + 2 * i + 1] = idx | (sh << 8);
        }
    const int blocks = 256 * wg_per_cu;
    uint32_t *d_prog, *d_src, *d_out;
    uint64_t *d_dt;
    CK(hipMalloc(&d_prog, prog.size() * 4));
    CK(hipMalloc(&d_src, src.size() * 4));
    CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CK(hipMalloc(&d_dt, (size_t)blocks * 4 * 8));
    CK(hipMemcpy(d_prog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = V == 3 ? 4 * RING * 256 + 1024 : (160 * 1024 / wg_per_cu) & ~255u;
    CK(hipFuncSetAttribute((const void *)mb_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mb_kernel<V>, dim3(blocks), dim3(256), lds, 0, d_prog, d_src, d_out, d_dt, nwords);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> out((size_t)blocks * 256);
    std::vector<uint64_t> dt((size_t)blocks * 4);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dt.data(), d_dt, dt.size() * 8, hipMemcpyDeviceToHost));
    // host model
    long bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        std::vector<uint32_t> ring(RING + 1);
        for (int i = 0; i <= RING; i++) ring[i] = src[i * 64 + lane];
        uint32_t acc = 0;
        for (int w = 0; w < nwords; w++) {
            if (V == 4) ring[prog[(size_t)w * 16 + 1] & 255u] = prog[(size_t)w * 16 + 14] ^ acc;
            uint32_t D = 0;
            for (int i = 0; i < NSEG; i++) {
                const uint32_t mask = prog[(size_t)w * 16 + 2 * i], pk = prog[(size_t)w * 16 + 2 * i + 1];
                const uint32_t idx = pk & 255u, sh = (pk >> 8) & 31u;
                const uint64_t pair = ((uint64_t)ring[idx + 1] << 32) | ring[idx];
                D |= (uint32_t)(pair >> sh) & mask;
            }
            acc ^= D;
            acc = (acc >> 1) | (acc << 31);
        }
        for (int b = 0; b < blocks * 4; b++)
            if (out[(size_t)b * 64 + lane] != acc) bad++;
    }
    std::sort(dt.begin(), dt.end());
    const double cyc_per_seg_wave = (double)dt[dt.size() / 2] / ((double)nwords * NSEG);
    const double ns_per_seg_simd = (double)ms * 1e6 / ((double)nwords * NSEG * wg_per_cu);  // wg_per_cu waves per SIMD
    printf("%-44s %d waves/SIMD: %6.2f s_memtime ticks per segment per wave; wall %.3f ms = %.2f ns per segment per SIMD = %.2f cycles @2.4GHz; wrong lanes %ld\n",
           name, wg_per_cu, cyc_per_seg_wave, ms, ns_per_seg_simd, ns_per_seg_simd * 2.4, bad);
    CK(hipFree(d_prog)); CK(hipFree(d_src)); CK(hipFree(d_out)); CK(hipFree(d_dt));
}

int main()
{
    const int NW = 4000;
    for (int k : {1, 2, 4}) {
        run<0>("v0 lshrrev_b64 SRC1_REL + and_or in mode", NW, k);
        run<1>("v1 alignbit SRC0|SRC1_REL batch, and_or after", NW, k);
        run<2>("v2 = v1 + s_nop after every index change", NW, k);
        run<4>("v4 = v1 + DST_REL commit per word", NW, k);
        run<3>("v3 LDS ring, compiler-scheduled C++", NW, k);
    }
    return 0;
}
