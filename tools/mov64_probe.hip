// mov64_probe.hip -- does v_mov_b64 take a DST_REL index (gfx9 VGPR index mode), and which indices?  For the scan-lane kernel's
// ring commits: two landing registers into two adjacent ring registers with ONE indexed move instead of two.
// Build: hipcc --offload-arch=gfx950 -O2 -Wno-inline-asm -o tools/bin/mov64_probe tools/mov64_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(64) void probe(uint32_t *out, uint32_t idx)
{
    const uint32_t lane4 = threadIdx.x * 4;
    const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(idx | 0x8000u));
    const uint64_t base = (uint64_t)out;
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base), bhi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    asm volatile(
        "s_mov_b32 s20, %[blo]\n\t"
        "s_mov_b32 s21, %[bhi]\n\t"
        "v_mov_b32 v44, 0x11110000\n\t"
        "v_mov_b32 v45, 0x22220000\n\t"
        "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\tv_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
        "s_set_gpr_idx_on s22, gpr_idx(SRC0)\n\t"
        "s_mov_b32 m0, %[m0v]\n\t"
        "v_mov_b64 v[60:61], v[44:45]\n\t"
        "s_mov_b32 m0, 0\n\t"
        "s_set_gpr_idx_off\n\t"
        "global_store_dword %[lane4], v60, s[20:21] offset:0\n\t"
        "global_store_dword %[lane4], v61, s[20:21] offset:256\n\t"
        "global_store_dword %[lane4], v62, s[20:21] offset:512\n\t"
        "global_store_dword %[lane4], v63, s[20:21] offset:768\n\t"
        "global_store_dword %[lane4], v64, s[20:21] offset:1024\n\t"
        "global_store_dword %[lane4], v65, s[20:21] offset:1280\n\t"
        "global_store_dword %[lane4], v66, s[20:21] offset:1536\n\t"
        "global_store_dword %[lane4], v67, s[20:21] offset:1792\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        :
        : [blo] "s"(blo), [bhi] "s"(bhi), [lane4] "v"(lane4), [m0v] "s"(m0v)
        : "memory", "scc", "m0", "s20", "s21", "s22", "v44", "v45", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
}

int main()
{
    uint32_t *d, h[8 * 64];
    CK(hipMalloc(&d, sizeof h));
    for (uint32_t idx = 0; idx < 6; idx++) {
        CK(hipMemset(d, 0xff, sizeof h));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, d, idx);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        printf("DST_REL index %u: v60..v67 (lane 0) =", idx);
        for (int r = 0; r < 8; r++) printf(" %08x", h[r * 64]);
        int ok = h[idx * 64] == 0x11110000u && h[(idx + 1) * 64] == 0x22220000u;
        for (int r = 0; r < 8; r++)
            if (r != (int)idx && r != (int)idx + 1 && h[r * 64] != 0) ok = 0;
        printf("  -> %s\n", ok ? "the pair landed at v[60 + idx : 61 + idx]" : "NOT as an indexed 64-bit move");
    }
    return 0;
}
