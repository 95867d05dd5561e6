// deskew.hip -- the batch's last stage: rotate every scan by the angle its sweep found, CONTAIN geometry
// (transfer.rs:487-519), NEAREST as correct_default does (omr.rs:408-445) or LINEAR as the benchmark drivers
// do (core/src/main.rs:72-81, app/src-tauri/src/test.rs:322-331).  The winning candidate's index is read on the
// device, so sweep -> arg-max -> warp need no host round trip; the fixed-point tables of OpenCV's warpAffine
// (adelta / bdelta per destination column, X0 / Y0 per destination row, AB_BITS = 10) exist per CANDIDATE, built
// once per batch context from host-computed matrices (fp64, same libm as the oracle), so a tap costs a few
// integer operations and the result is bit-identical to omr_rotate_device on the same scan.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.hpp"

namespace omr {

#define DW_TW 128  // destination tile: 128 x 32 pixels per 256-thread workgroup, 4 x 4 pixels per thread
// tile height: 64 rows for NEAREST (the fixed chain of memory round trips per tile is spread over more pixels),
// 32 for LINEAR (four taps per pixel: at 64 rows the pixels in flight cost 200 VGPRs)
#ifndef DW_TH_NN
#define DW_TH_NN 64
#define DW_TH_LIN 64
#define DW_LDS 16384
#endif
#define DW_TH (LINEAR ? DW_TH_LIN : DW_TH_NN)

__device__ __forceinline__ int dw_mad24(int a, int b, int c) { return __mul24(a, b) + c; }  // v_mad_i32_i24
// The same, spelled out: the compiler turns __mul24 of operands it has folded into other arithmetic back into a plain multiply
// and then picks v_mul_lo_u32 -- a quarter-rate instruction (16 cycles per wave: four of the usual) -- for it; the vertical
// blend of the bilinear tap was one (round 5, second half: the disassembly showed 12 such multiplies per 8 pixels, 29 % of the
// loop's vector time).  |a|, |b| < 2^23 is the caller's business.
__device__ __forceinline__ int dw_mad24_asm(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint8_t dw_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

template <bool LINEAR>
__device__ __forceinline__ int dw_tap_global(const uint8_t *__restrict__ src, int64_t sstep, int srows, int scols, int Xf,
                                             int Yf, int border)
{
    if (!LINEAR) {
        const int X = max(-32768, min(32767, Xf >> 10)), Y = max(-32768, min(32767, Yf >> 10));
        return ((unsigned)X < (unsigned)scols && (unsigned)Y < (unsigned)srows) ? src[(int64_t)Y * sstep + X] : border;
    }
    const int X = Xf >> 5, Y = Yf >> 5;
    const int sx = max(-32768, min(32767, X >> 5)), sy = max(-32768, min(32767, Y >> 5));
    const int fx = X & 31, fy = Y & 31;
    if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) return border;
    const bool in_x0 = sx >= 0 && sx < scols, in_x1 = sx + 1 >= 0 && sx + 1 < scols;
    const bool in_y0 = sy >= 0 && sy < srows, in_y1 = sy + 1 >= 0 && sy + 1 < srows;
    const int v0 = in_x0 && in_y0 ? src[(int64_t)sy * sstep + sx] : border;
    const int v1 = in_x1 && in_y0 ? src[(int64_t)sy * sstep + sx + 1] : border;
    const int v2 = in_x0 && in_y1 ? src[(int64_t)(sy + 1) * sstep + sx] : border;
    const int v3 = in_x1 && in_y1 ? src[(int64_t)(sy + 1) * sstep + sx + 1] : border;
    const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    return dw_sat_u8((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
}

// ---- per-tile records (round 5).  The warp kernel used to be a chain of FOUR dependent memory round trips per tile
// (winner -> canvas size -> table entries of the corners -> box -> taps) with five workgroups per CU to hide them: 30 / 50
// lane-cycles per pixel for 9 / 25 operations.  Everything that depends only on (scan, tile) -- the winner, the canvas, the
// four corner samples and the source box they bound -- is now worked out by one thread per tile in a kernel of its own
// (deskew_tiles_kernel: 10 000 tiles per 8 A4 scans), and the warp kernel starts from a 32-byte record: record -> {table
// entries, box} -> taps.
struct DeskewTile {
    int32_t a;        // winning candidate; -1 = the tile lies outside this scan's canvas
    int32_t bb0, by0; // the box: first byte column (a multiple of 4), first row
    int32_t bwb, bh;  // its row length in bytes (a multiple of 4), its rows; bwb = 0: every tap of the tile is border
    int32_t a1, b1;   // adelta / bdelta of the tile's last column (columns past it repeat it)
    int32_t last;     // tx1 | ty1 << 16: the tile's last column and row inside the canvas
};

template <bool LINEAR>
__global__ __launch_bounds__(64) void deskew_tiles_kernel(const DeskewPass p, DeskewTile *__restrict__ tiles, int ntx, int nty)
{
    const int t = blockIdx.x * 64 + threadIdx.x, z = blockIdx.y;
    if (t >= ntx * nty) return;
    const int tyi = t / ntx, txi = t - tyi * ntx;
    const int a = p.best[z];
    const int drows = p.wsize[2 * a], dcols = p.wsize[2 * a + 1];
    if (t == 0 && p.out_size) {
        p.out_size[2 * z] = drows;
        p.out_size[2 * z + 1] = dcols;
    }
    DeskewTile r;
    r.a = -1, r.bb0 = r.by0 = r.bwb = r.bh = r.a1 = r.b1 = r.last = 0;
    const int tx0 = txi * DW_TW, ty0 = tyi * DW_TH;
    if (tx0 < dcols && ty0 < drows) {
        const int32_t *__restrict__ AD = p.adelta + (int64_t)a * p.DC, *__restrict__ BD = p.bdelta + (int64_t)a * p.DC;
        const int2_t *__restrict__ XY = p.xy0 + (int64_t)a * p.DR;
        const int rd = LINEAR ? 16 : 512;
        const int tx1 = min(dcols, tx0 + DW_TW) - 1, ty1 = min(drows, ty0 + DW_TH) - 1;
        const int2_t r0 = XY[ty0], r1 = XY[ty1];
        const int a0 = AD[tx0], a1 = AD[tx1], b0 = BD[tx0], b1 = BD[tx1];
        // fixed-point source coordinates of the tile's corner samples.  X0(y) and adelta(x) are both monotone, so the four
        // corners bound every sample of the tile (one pixel of slack for the rounding of the tables, one more for the
        // bilinear taps)
        const int cx[4] = {(r0.x + rd + a0) >> 10, (r0.x + rd + a1) >> 10, (r1.x + rd + a0) >> 10, (r1.x + rd + a1) >> 10};
        const int cy[4] = {(r0.y + rd + b0) >> 10, (r0.y + rd + b1) >> 10, (r1.y + rd + b0) >> 10, (r1.y + rd + b1) >> 10};
        const int bx0 = min(min(cx[0], cx[1]), min(cx[2], cx[3])) - 1;
        const int bx1 = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 1 + (LINEAR ? 1 : 0);
        const int by0 = min(min(cy[0], cy[1]), min(cy[2], cy[3])) - 1;
        const int by1 = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 1 + (LINEAR ? 1 : 0);
        const int bb0 = bx0 & ~3, bb1 = (bx1 + 4) & ~3;  // the box in bytes of a source row, widened to whole dwords: [bb0, bb1)
        const int bwb = bb1 - bb0, bh = by1 - by0 + 1;
        // (a scan whose rows are not whole aligned dwords -- width, pitch or address not a multiple of 4 -- takes the
        // unstaged path: the staging loop then has no partial dwords and no branches)
        // ... nor a scan of 2 GB or more: the staging loads address it by 32-bit byte offsets through a buffer descriptor
        const bool dwords = ((p.sstep | p.scan_stride | (int64_t)(uintptr_t)p.src | (int64_t)p.scols) & 3) == 0 &&
                            (int64_t)p.srows * p.sstep < (int64_t)0x7fffffff;
        // ... nor a canvas of 2 GB or more, or with rows of 16 MB: the staged path's stores address the scan's canvas by 32-bit
        // offsets made with a 24-bit multiply (a 64-bit address per store row cost two full-rate-quarter multiplies, see below)
        const bool canvas32 = p.dstep < (1 << 24) && (int64_t)p.DR * p.dstep < (int64_t)0x7fffffff;
        const bool staged = dwords && canvas32 && bwb > 0 && bh > 0 && (int64_t)bwb * bh <= DW_LDS && bx0 > -30000 && bx1 < 30000 &&
                            by0 > -30000 && by1 < 30000;
        // A tile whose box lies wholly outside the scan (the corners of a CONTAIN canvas: up to a fifth of it at 10 degrees)
        // is the border value: every tap of every sample is outside, NEAREST takes it as it is and the bilinear weights of
        // four equal taps sum to 2^15
        const bool all_border = bx1 < 0 || by1 < 0 || bx0 >= p.scols || by0 >= p.srows;
        // a box that lies wholly inside the image -- all but the tiles on the scan's edges -- is fetched in 16-byte pieces
        // (rows padded to a multiple of 16 bytes in LDS; the padding stays inside the image row too): a quarter of the
        // DMA instructions.  Bit 30 of bh says so.
        const int bwb16 = (bwb + 15) & ~15;
        const bool x4 = staged && bb0 >= 0 && by0 >= 0 && by0 + bh <= p.srows && bb0 + bwb16 <= p.scols && (int64_t)bwb16 * bh <= DW_LDS;
        r.a = a, r.bb0 = bb0, r.by0 = by0;
        r.bwb = all_border ? 0 : x4 ? bwb16 : staged ? bwb : -1;  // the box's pitch in LDS; -1: taps from global memory
        r.bh = bh | (x4 ? 1 << 30 : 0), r.a1 = a1, r.b1 = b1, r.last = tx1 | (ty1 << 16);
    }
    tiles[(int64_t)z * ntx * nty + t] = r;
}

// grid = (tiles across the largest canvas, tiles down it, scans of the launch); a tile outside its scan's canvas
// leaves at once.  (Scans fastest instead -- XCD x warps only scan x, so overlapping boxes meet in one L2 -- halves the
// HBM fetch, 71.8 -> 34.3 MB per 8 scans = the scans' own bytes, but is no faster: 65 / 114 us against 66 / 105.  The
// kernel is not waiting for HBM.  Nor for its chain of round trips alone: a software-pipelined form -- a vertical strip of
// four 128 x 32 tiles per workgroup, tile i sampled from one LDS box while the box of tile i + 1 is in flight and the
// table entries of tile i + 2 are requested -- was built, bit-exact, and slower: 103 / 138 us at 112 / 168 VGPRs and
// 32 KB of LDS per workgroup.)  The source bounding box of a tile (four corner samples; one pixel of slack for the rounding of
// the fixed-point tables, one more for the bilinear taps) is staged in LDS with row-contiguous dword loads, border
// value outside the image, so a tap is one LDS byte read with no bounds test.  The kernel is a chain of dependent
// memory round trips (tile record -> {table entries, box} -> taps): everything that depends on (scan, tile) alone was
// worked out by deskew_tiles_kernel, and the thread's own column and row table entries are requested with the box.
template <bool LINEAR>
// (forcing more waves per SIMD was measured and lost: __launch_bounds__(256, 6) spills, 127 / 126 us per 8 A4 scans
// against 62 / 104 at the compiler's own 100 / 92 VGPRs)
#ifndef DW_MIN_BLOCKS
#define DW_MIN_BLOCKS 8
#endif
__global__ __launch_bounds__(256, DW_MIN_BLOCKS) void deskew_warp_kernel(const DeskewPass p, const DeskewTile *__restrict__ tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t box[DW_LDS];
    // grid = (scans, tiles across, tiles down), scans fastest: workgroup ids go round-robin to the 8 XCDs, so XCD x warps the
    // scans x, x + 8, .. only and the overlapping boxes of a scan's neighbouring tiles meet in ONE L2 (with tiles fastest every
    // L2 saw every eighth tile of a row and each box was fetched from memory in full: 2 x the scans' bytes)
    // (p.order = 0: tiles fastest, grid = (tiles across, tiles down, scans); 1: scans fastest; 2: a 1-D grid in which XCD k --
    // workgroup ids go round-robin to the 8 XCDs -- warps the tile ROWS k, k + 8, .. of every scan, left to right: the
    // overlapping boxes of a row's neighbouring tiles meet in one L2, and every XCD has the same share of every scan)
    int z, txi, tyi, ntx, nty;
    if (p.order == 2) {
        ntx = p.ntx, nty = p.nty;
        const int nty8 = (nty + 7) >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        txi = j % ntx;
        const int q = j / ntx, r = q % nty8;
        z = q / nty8, tyi = r * 8 + xcd;
        if (tyi >= nty) return;
    } else {
        z = p.order ? blockIdx.x : blockIdx.z, txi = p.order ? blockIdx.y : blockIdx.x, tyi = p.order ? blockIdx.z : blockIdx.y;
        ntx = p.order ? gridDim.y : gridDim.x, nty = p.order ? gridDim.z : gridDim.y;
    }
    // the tile's record: wave-uniform, fetched by the scalar unit
    const DeskewTile *__restrict__ tr = tiles + ((int64_t)z * nty + tyi) * ntx + txi;
    const int a = tr->a;
    if (a < 0) return;  // outside this scan's canvas
    const int bb0 = tr->bb0, by0 = tr->by0, bwb_rec = tr->bwb, bh_rec = tr->bh, a1 = tr->a1, b1 = tr->b1, last = tr->last;
    const int bh = bh_rec & 0xffff;
    const bool x4 = (bh_rec >> 30) & 1;  // the box is wholly inside the image: 16-byte pieces
    const int tx0 = txi * DW_TW, ty0 = tyi * DW_TH;
    const int tx1 = last & 0xffff, ty1 = last >> 16;
    const int dcols = tx1 + 1, drows = ty1 + 1;  // as far as this tile can tell: enough for every test below
    const uint8_t *__restrict__ src = p.src + (int64_t)z * p.scan_stride;
    uint8_t *__restrict__ dst = p.dst + (int64_t)z * p.out_stride;
    const int32_t *__restrict__ AD = p.adelta + (int64_t)a * p.DC, *__restrict__ BD = p.bdelta + (int64_t)a * p.DC;
    const int2_t *__restrict__ XY = p.xy0 + (int64_t)a * p.DR;
    const int rd = LINEAR ? 16 : 512;
    const int lx = (threadIdx.x & 31) * 4, x0 = tx0 + lx, yq = ty0 + (threadIdx.x >> 5);
    const uint32_t border4 = (uint32_t)p.border * 0x01010101u;
    if (bwb_rec == 0) {  // every tap of the tile is border (workgroup-uniform)
        if (x0 >= dcols) return;
        const bool whole4 = x0 + 4 <= dcols && ((p.dstep | p.out_stride | (int64_t)(uintptr_t)p.dst) & 3) == 0;
#pragma unroll
        for (int k = 0; k < DW_TH / 8; k++) {
            uint8_t *D = dst + (int64_t)min(yq + 8 * k, ty1) * p.dstep + x0;
            if (whole4) {
                *(uint32_t *)D = border4;
            } else {
                for (int j = 0; j < 4 && x0 + j < dcols; j++) D[j] = (uint8_t)p.border;
            }
        }
        return;
    }
    const bool staged = bwb_rec > 0;
    const int bwb = staged ? bwb_rec : 4;
    // ---- the thread's own column and row table entries: requested together with the box, they depend on the record only
    const int xl = min(x0, p.DC - 4);  // DC is a multiple of 4 and the tables are 16-byte aligned
    const int4 ad = *(const int4 *)(AD + xl), bd = *(const int4 *)(BD + xl);
    // (rows past the tile's last one repeat it and columns past its last one repeat that column below: every sample a
    // thread computes then lies inside the tile's box, stored or not, and a tap needs no bounds test)
    // the tile's row table entries travel through LDS (thread t brings row ty0 + t): one load per thread instead of DW_TH / 8,
    // and no registers hold them while the box is in flight
    __shared__ int2_t rowtab[DW_TH];
    static_assert(DW_TH_NN <= 256 && DW_TH_LIN <= 256, "one thread per row of the tile brings its table entry");
    if (threadIdx.x < DW_TH) rowtab[threadIdx.x] = XY[min(ty0 + (int)threadIdx.x, ty1)];
    if (staged) {
        // the box as dwords, thread t takes dwords t, t + 256, ..: LDS-DMA (`buffer_load_dword ... lds`: the 64 lanes of a
        // wave-instruction write 64 consecutive dwords of the box, each from its own offset into the scan) -- no registers
        // hold the box on its way (round 4 kept 16 dwords and 16 flags per thread across the round trip: 91 VGPRs, five
        // workgroups per CU), no ds_write instructions, and every load of the tile is in flight at once.  The scan is a raw
        // buffer: an offset outside it (a box that pokes out above or below the image; dwords past the box's end) lands as
        // zeros.  A box that is not wholly inside the image is patched afterwards (workgroup-uniform branch, tiles on the
        // scan's edges only): dwords outside the image become the border value -- those read through the row pitch into the
        // neighbouring row included.  (row, dword in row) of a thread's next piece follows from the previous one without a
        // division; ONE branch on the (workgroup-uniform) size of the box selects among three straight-line versions.
        constexpr int NP = DW_LDS / 4 / 256;  // 16 pieces per thread at most
        const int bq = bwb >> 2, total = bq * bh;
        const int dq = 256 / bq, dr = 256 - dq * bq;  // 256 = dq * bq + dr
        const uint64_t sb = (uint64_t)src;
        const uint32_t nbytes = (uint32_t)min((int64_t)0xfffffff0ll, (int64_t)p.srows * p.sstep);
        uint32_t rs[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb),
                          (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32) & 0xffffu),
                          (uint32_t)__builtin_amdgcn_readfirstlane(nbytes), 0x00020000u};
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u rsrc = {rs[0], rs[1], rs[2], rs[3]};
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)box +
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 256u;  // this wave's 64 dwords of a piece
        const bool inside = bb0 >= 0 && bb0 + bwb <= p.scols && by0 >= 0 && by0 + bh <= p.srows;
        auto stage = [&](auto np_tag) {
            constexpr int NPV = decltype(np_tag)::value;
            int ly = (int)threadIdx.x / bq, lq = (int)threadIdx.x - ly * bq;
#pragma unroll
            for (int n = 0; n < NPV; n++) {
                const int gy = by0 + ly, gb = bb0 + lq * 4;
                // (a negative row or a piece past the box: an offset no buffer of less than 4 GB contains)
                const uint32_t voff = ((int)threadIdx.x + n * 256 < total && gy >= 0) ? (uint32_t)(gy * (int)p.sstep + gb) : 0xfffffff0u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                             :
                             : "s"(lds0 + (uint32_t)n * 1024u), "v"(voff), "s"(rsrc)
                             : "memory", "m0");
                ly += dq, lq += dr;
                if (lq >= bq) lq -= bq, ly++;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!inside) {  // workgroup-uniform
                __syncthreads();
                ly = (int)threadIdx.x / bq, lq = (int)threadIdx.x - ly * bq;
#pragma unroll
                for (int n = 0; n < NPV; n++) {
                    const int gy = by0 + ly, gb = bb0 + lq * 4;
                    if ((int)threadIdx.x + n * 256 < total && !((unsigned)gy < (unsigned)p.srows && (unsigned)gb < (unsigned)p.scols))
                        *(uint32_t *)&box[((int)threadIdx.x + n * 256) * 4] = border4;
                    ly += dq, lq += dr;
                    if (lq >= bq) lq -= bq, ly++;
                }
            }
        };
        // 16-byte pieces (buffer_load_dwordx4 ... lds: lane L writes M0 + 16 L) for a box wholly inside the image
        auto stage16 = [&](auto np_tag) {
            constexpr int NPV = decltype(np_tag)::value;
            const int bq4 = bwb >> 4, total4 = bq4 * bh;
            const int dq4 = 256 / bq4, dr4 = 256 - dq4 * bq4;
            int ly = (int)threadIdx.x / bq4, lq = (int)threadIdx.x - ly * bq4;
            const uint32_t lds16 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)box +
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 1024u;
#pragma unroll
            for (int n = 0; n < NPV; n++) {
                const uint32_t voff = (int)threadIdx.x + n * 256 < total4 ? (uint32_t)((by0 + ly) * (int)p.sstep + bb0 + lq * 16) : 0xfffffff0u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             :
                             : "s"(lds16 + (uint32_t)n * 4096u), "v"(voff), "s"(rsrc)
                             : "memory", "m0");
                ly += dq4, lq += dr4;
                if (lq >= bq4) lq -= bq4, ly++;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (x4) {
            constexpr int NP4 = DW_LDS / 16 / 256;  // 16-byte pieces per thread at most
            const int total4 = (bwb >> 4) * bh;
            if (total4 <= (NP4 / 2) * 256) stage16(std::integral_constant<int, NP4 / 2>{});
            else if (total4 <= (3 * NP4 / 4) * 256) stage16(std::integral_constant<int, 3 * NP4 / 4>{});
            else stage16(std::integral_constant<int, NP4>{});
        } else if (total <= (NP / 2) * 256) stage(std::integral_constant<int, NP / 2>{});
        else if (total <= (3 * NP / 4) * 256) stage(std::integral_constant<int, 3 * NP / 4>{});
        else stage(std::integral_constant<int, NP>{});
    }
    __syncthreads();
    if (x0 >= dcols) return;
    int adv[4] = {ad.x, ad.y, ad.z, ad.w}, bdv[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {  // (xl == x0 unless the thread starts past the tables' last four columns, and then past tx1 too)
        const bool past = x0 + j > tx1;
        adv[j] = past ? a1 : adv[j];
        bdv[j] = past ? b1 : bdv[j];
    }
    if (!staged) {  // workgroup-uniform: a box too large for LDS (strong shear) or rows that are not whole dwords; taps from global memory
        for (int k = 0; k < DW_TH / 8; k++) {
            const int y = yq + 8 * k;
            if (y >= drows) break;
            uint8_t *D = dst + (int64_t)y * p.dstep + x0;
            for (int j = 0; j < 4 && x0 + j < dcols; j++)
                D[j] = (uint8_t)dw_tap_global<LINEAR>(src, p.sstep, p.srows, p.scols, rowtab[y - ty0].x + rd + adv[j],
                                                      rowtab[y - ty0].y + rd + bdv[j], p.border);
        }
        return;
    }
    // box[(sy - by0) * bwb + (sx - bb0)]: the box's origin rides on the row's X0 / Y0 (multiples of 1024: the fraction
    // bits stay; |coordinates| < 30000 pixels, so nothing overflows).  Straight-line: the 16 / 32 pixels' taps are all requested before the first is blended,
    // and nothing below is conditional except the last tile column's byte stores -- a row past the tile's last one
    // recomputes that row's pixels (its table entry was clamped) and stores them there again, the same bytes.
    // The box's origin rides on the thread's column entries (multiples of 1024: the fraction bits stay; |coordinates| <
    // 30000 pixels, so nothing overflows): a sample is two additions, two shifts and one multiply-add away from its LDS byte.
    const int orgx = rd - (bb0 << 10), orgy = rd - (by0 << 10);
#pragma unroll
    for (int j = 0; j < 4; j++) adv[j] += orgx, bdv[j] += orgy;
    // stores: 32-bit offsets from the scan's canvas (a canvas is far below 4 GB); a tile whose every thread stores whole
    // aligned dwords -- all but the canvas's last tile column, or an unaligned canvas -- has no per-lane store branches
    const uint32_t dstep32 = (uint32_t)p.dstep;
    // (a staged tile's canvas is below 2 GB with rows below 16 MB: deskew_tiles_kernel)
    const bool aligned4 = ((p.dstep | p.out_stride | (int64_t)(uintptr_t)p.dst) & 3) == 0;
    const bool tile_whole = aligned4 && tx0 + DW_TW <= dcols;  // workgroup-uniform
    const bool whole = aligned4 && x0 + 4 <= dcols;
    const int rt0 = (int)(threadIdx.x >> 5);  // this thread's first row of the tile; rowtab is clamped to the tile's last row
    // DW_KU rows of four pixels at a time: their taps are all requested before the first is blended; the loop over the
    // groups is NOT unrolled (with all 8 rows in flight the kernel needed 104 VGPRs: four workgroups per CU)
    constexpr int DW_KU = LINEAR ? 2 : 4;
#pragma unroll 1
    for (int k0 = 0; k0 < DW_TH / 8; k0 += DW_KU) {
        uint32_t outs[DW_KU];
#pragma unroll
        for (int kk = 0; kk < DW_KU; kk++) {
            const int2_t rw = rowtab[rt0 + 8 * (k0 + kk)];
            uint32_t bts[4];
            if constexpr (LINEAR) {
                // ((v0 w0 + v1 w1 + v2 w2 + v3 w3 + 2^14) >> 15 with w = 32 (32 - fy)(32 - fx), ..) = two horizontal blends and
                // a vertical one: the same integer, and <= 255 without a clamp.  Round 5, second half -- 18.5 -> 16.7 vector
                // instructions per pixel: the two taps of a box row are packed into 16-bit halves (one v_lshl_or) and blended by
                // ONE v_dot2_u32_u16 with the weights (64 - 2 fx, 2 fx) -- twice the reference's, so that with fy * 32 for the
                // vertical weight the accumulator is the reference's << 6 and the output byte is its byte 2: no shift, v_perm
                // packs the four pixels.  (ds_read_u8_d16 / _d16_hi would pack for free, but on this chip -- SRAM ECC -- a d16
                // load zeroes the half it does not write: tried, wrong bytes; 16-bit LDS reads at byte addresses would halve the
                // LDS instructions and cost 6 - 12 x their time: tools/lds_u16_probe.hip.)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
                    const int Xf = rw.x + adv[j], Yf = rw.y + bdv[j];
                    const uint8_t *Bx = &box[dw_mad24(Yf >> 10, bwb, Xf >> 10)];
                    const uint32_t t = (uint32_t)Bx[0] | ((uint32_t)Bx[1] << 16), u = (uint32_t)Bx[bwb] | ((uint32_t)Bx[bwb + 1] << 16);
                    const v2u16 w = __builtin_bit_cast(v2u16, __umul24((uint32_t)(Xf >> 5) & 31u, 131070u) + 64u);  // (64 - 2 fx) | 2 fx << 16
                    const int top2 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, t), w, 0u, false);
                    const int bot2 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, u), w, 0u, false);
                    bts[j] = (uint32_t)dw_mad24_asm(Yf & 0x3e0, bot2 - top2, (top2 << 10) + 32768);  // the pixel is byte 2
                }
                outs[kk] = __builtin_amdgcn_perm(bts[1], bts[0], 0x0c0c0602u) | __builtin_amdgcn_perm(bts[3], bts[2], 0x06020c0cu);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int Xf = rw.x + adv[j], Yf = rw.y + bdv[j];
                const int idx = dw_mad24(Yf >> 10, bwb, Xf >> 10);
                bts[j] = box[idx];
            }
            // four bytes, each < 256 in a register of its own -> one dword: three v_lshl_or
            outs[kk] = ((bts[1] << 8) | bts[0]) | (((bts[3] << 8) | bts[2]) << 16);
        }
#pragma unroll
        for (int kk = 0; kk < DW_KU; kk++) {
            // a row past the tile's last one recomputed that row's pixels (rowtab is clamped) and stores them there again
            const uint32_t off = (uint32_t)__mul24(min(yq + 8 * (k0 + kk), ty1), (int)dstep32) + (uint32_t)x0;
            if (tile_whole) {
                *(uint32_t *)(dst + off) = outs[kk];
            } else if (whole) {
                *(uint32_t *)(dst + off) = outs[kk];
            } else {
                // (the same 32-bit offset: a 64-bit row address here is computed for every row, taken or not -- the compiler
                // hoists it above the branch -- with two quarter-rate multiplies)
                uint8_t *D = dst + off;
                for (int j = 0; j < 4 && x0 + j < dcols; j++) D[j] = (uint8_t)(outs[kk] >> (8 * j));
            }
        }
    }
}

size_t deskew_tile_bytes(const DeskewPass &p, int scans)
{
    // (tiles of the LINEAR kernel: the smaller tile, so the buffer serves both)
    constexpr int th = DW_TH_LIN < DW_TH_NN ? DW_TH_LIN : DW_TH_NN;
    return sizeof(DeskewTile) * (size_t)((p.DC + DW_TW - 1) / DW_TW) * (size_t)((p.DR + th - 1) / th) * (size_t)scans;
}

hipError_t launch_deskew_warp(const DeskewPass &p0, int scans, int interp, void *d_tiles, hipStream_t s)
{
    if (scans <= 0) return hipSuccess;
    DeskewPass p = p0;
#ifdef DW_ORDER
    p.order = DW_ORDER;
#else
    p.order = 2;
#endif
    if ((p.DC & 3) != 0 || !d_tiles) return hipErrorInvalidValue;
    DeskewTile *tiles = (DeskewTile *)d_tiles;
    if (interp == 0) {
        constexpr bool LINEAR = false;
        const int ntx = (p.DC + DW_TW - 1) / DW_TW, nty = (p.DR + DW_TH - 1) / DW_TH;
        hipLaunchKernelGGL(deskew_tiles_kernel<false>, dim3((ntx * nty + 63) / 64, scans), dim3(64), 0, s, p, tiles, ntx, nty);
        p.ntx = ntx, p.nty = nty;
        hipLaunchKernelGGL(deskew_warp_kernel<false>, p.order == 2 ? dim3(8 * ntx * ((nty + 7) / 8) * scans) : p.order ? dim3(scans, ntx, nty) : dim3(ntx, nty, scans), dim3(256), 0, s, p, tiles);
    } else {
        constexpr bool LINEAR = true;
        const int ntx = (p.DC + DW_TW - 1) / DW_TW, nty = (p.DR + DW_TH - 1) / DW_TH;
        hipLaunchKernelGGL(deskew_tiles_kernel<true>, dim3((ntx * nty + 63) / 64, scans), dim3(64), 0, s, p, tiles, ntx, nty);
        p.ntx = ntx, p.nty = nty;
        hipLaunchKernelGGL(deskew_warp_kernel<true>, p.order == 2 ? dim3(8 * ntx * ((nty + 7) / 8) * scans) : p.order ? dim3(scans, ntx, nty) : dim3(ntx, nty, scans), dim3(256), 0, s, p, tiles);
    }
    return hipGetLastError();
}

}  // namespace omr
