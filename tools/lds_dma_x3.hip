// lds_dma_x3.hip -- probe for the sweep kernel's window fetch: does buffer_load_dwordx3 ... lds (gfx950) take per-lane
// 4-byte-aligned offsets, and what lands in LDS for lanes whose offset is outside [0, num_records)?
// Build + run (GPU box): hipcc -O2 --offload-arch=gfx950 tools/lds_dma_x3.hip -o /tmp/lds_dma_x3 && /tmp/lds_dma_x3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
// probe: buffer_load_dwordx3 ... lds with per-lane offsets, out-of-range lanes
__global__ void k(const uint32_t* src, int nbytes, const int* offs, uint32_t* out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* l = (uint32_t*)lds;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) l[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + wave * 768), 12, offs[threadIdx.x], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = l[i];
}
int main()
{
    const int N = 4096;
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; i++) h[i] = 0x1000000u + i;
    uint32_t *d, *o; int* doff;
    hipMalloc(&d, N * 4); hipMalloc(&o, 4096); hipMalloc(&doff, 128 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<int> off(128);
    for (int t = 0; t < 128; t++) off[t] = 4 * (t * 5 + 1);      // unaligned-to-12 word offsets
    off[3] = -8; off[7] = (int)0x80000000u; off[9] = 1000 * 4 - 4;  // negative, far, straddling the end
    off[70] = 1000 * 4;                                          // at the end
    hipMemcpy(doff, off.data(), 128 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096, 0, d, 1000 * 4, doff, o);
    std::vector<uint32_t> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 128; t++) {
        const int w = t >> 6, lane = t & 63;
        for (int j = 0; j < 3; j++) {
            const uint32_t got = r[(w * 768 + lane * 12) / 4 + j];
            const long long byte = (long long)off[t] + 4 * j;
            uint32_t want = (byte >= 0 && byte + 4 <= 4000 && off[t] >= 0) ? h[byte / 4] : 0;
            if (got != want) { bad++; printf("t=%d j=%d off=%d got=%08x want=%08x\n", t, j, off[t], got, want); }
        }
    }
    printf("x3 lds dma probe: %d mismatches (partial-OOB rows show the per-dword rule)\n", bad);
    return 0;
}
