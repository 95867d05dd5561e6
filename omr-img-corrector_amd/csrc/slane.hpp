// slane.hpp -- the scan-lane sweep: lane = SCAN.
//
// All scans of a batch share one shape and one candidate set, so the geometry of the sweep -- which source
// row, which source word column, which shift and which destination bits make up destination word (r, w) of
// candidate a (projection.rs:47-65 -> transfer.rs:459-486 -> OpenCV warpAffine NEAREST, AB_BITS = 10) -- is the
// same for every scan.  64 scans ride in the 64 lanes of a wave; the geometry becomes a wave-uniform PROGRAM,
// enumerated once per plan from the integer tables of warpAffine (bit-exact by construction) and streamed
// through the scalar unit; the only per-lane data is the scan's own bits.
//
//   bit image  : interleaved, entry E(s, c) = 64 dwords = word column c of source row s of the 64 scans of a
//                scan group; entry 0 is all zero (dummy fetches), SL_GX zero guard columns left and right
//   strip      : SL_K = 2 adjacent destination word columns; a wave owns (scan group, candidate, strip) and
//                walks ALL destination rows top to bottom
//   ring       : the wave keeps the source entries it needs in 64 VGPRs: register (s & 15) * 4 + (c - cb(s));
//                every source row s has its own column base cb(s), chosen by the generator
//   records    : one per destination row, plus SL_PRE virtual rows ahead of row 0 that only fill the ring and a
//                virtual row at the end when that makes the count even (the kernel sweeps two rows per turn).
//                Two streams per strip:
//                  fetch stream, 4 dwords per row: E << 8 | ring register (64 = dummy, E = 0): the entries to
//                                   load while this row is swept; they are committed to the ring before the
//                                   NEXT row (byte offset of the entry = the dword with its low byte cleared)
//                  segment stream, SL_K words x S x (mask, pk) per row: destination bits `mask` = bits (sh + i) of
//                                   the register pair (ring[idx + 1] : ring[idx]); pk = idx | sh << 8, the
//                                   first pk of a word also carries its segment count n << 16
//                S = 2 / 4 / 8 slots per word (class 0 / 1 / 2: 8 / 16 / 32 dwords per row), chosen per strip
#pragma once
#include <stdint.h>

#include <vector>

namespace omr {

constexpr int SL_K = 2;
constexpr int SL_RING_ROWS = 16;
constexpr int SL_RING_COLS = 4;
constexpr int SL_FETCH = 4;
constexpr int SL_DUMMY = SL_RING_ROWS * SL_RING_COLS;  // ring register that swallows dummy fetches
constexpr int SL_PRE = 16;                              // virtual rows ahead of row 0
constexpr int SL_GX = 4;                                // zero guard word columns on either side
constexpr int SL_LANES = 64;

struct SlaneGeom {
    int rows = 0, cols = 0;  // image
    int NW = 0;              // words per row
    int colsG = 0;           // NW + 2 SL_GX
    int NS = 0;              // strips
    int64_t entries = 0;     // 1 + rows * colsG
    void set(int r, int c)
    {
        rows = r, cols = c, NW = (c + 31) / 32, colsG = NW + 2 * SL_GX, NS = (NW + SL_K - 1) / SL_K;
        entries = 1 + (int64_t)r * colsG;
    }
    int64_t entry(int s, int c) const { return 1 + (int64_t)s * colsG + (c + SL_GX); }
};

struct SlaneStrip {     // per (candidate, strip)
    int64_t seg_offset; // of the segment stream, in dwords from the program base (a multiple of 64)
    int64_t fet_offset; // of the fetch stream
    int32_t cls;        // 0 / 1 / 2 = 2 / 4 / 8 segment slots per word; -1 = the strip does not fit this scheme
    int32_t nseg;       // most segments any of its words needs
};

inline int slane_slots(int cls) { return 2 << cls; }
inline int slane_seg_dwords(int cls) { return SL_K * 2 * slane_slots(cls); }  // per row
inline int slane_class(int most) { return most <= 2 ? 0 : most <= 4 ? 1 : most <= 8 ? 2 : -1; }
inline int slane_records(int rows) { return (SL_PRE + rows + 1) & ~1; }

// warpAffine's integer tables of one candidate on the host (the expressions of tables_kernel, kernels.hip;
// built -ffp-contract=off): adelta / bdelta per column, (X0, Y0) per row with round_delta = 512
void slane_host_tables(const double Minv[6], int rows, int cols, std::vector<int32_t> &ad, std::vector<int32_t> &bd,
                       std::vector<int32_t> &x0, std::vector<int32_t> &y0);

// Pass 1: the most segments a word of strip `strip` needs, or -1 when the strip does not fit (a source row
// that needs more than SL_RING_COLS word columns, more than 8 segments).
int slane_strip_segments(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip);
// Pass 2: the strip's program: slane_records(rows) rows of slane_seg_dwords(cls) dwords into seg, of SL_FETCH dwords
// into fet.  Returns false when the ring schedule fails (the candidate then stays with the run-merging kernel).
bool slane_strip_program(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip, int cls, uint32_t *seg, uint32_t *fet);

// What a wave of slane_kernel starts from (64 bytes, read by scalar loads; tools/gen_slane_asm.py fixes the layout)
struct SlaneTask {
    uint64_t seg, fet;   // the strip's segment / fetch stream
    uint64_t hrow;       // row counts of (candidate, scan group): u32 [record][scan], record 0 = first virtual row
    uint64_t planes;     // counter dump of (task, scan group): [word][18][64]
    uint32_t rsrc[4];    // buffer descriptor of the scan group's interleaved bit image
    uint32_t nrec;       // records (even)
    uint32_t hpitch;     // bytes between rows of hrow
    int32_t cls;         // 0 / 1 / 2 = 2 / 4 / 8 slots per word
    int32_t pad;
};
static_assert(sizeof(SlaneTask) == 64, "SlaneTask layout (slane_asm.inc loads it by offset)");

}  // namespace omr

#include <hip/hip_runtime_api.h>
namespace omr {
hipError_t launch_slane_pack(const uint8_t *d_img, int64_t scan_stride, int64_t step, const SlaneGeom &g, int nscans,
                             int black_max, uint32_t *d_bits, hipStream_t s);
// nsg_used = scan groups that carry scans in this launch, nsg = scan groups the scratch (and its descriptors) is laid out for
hipError_t launch_slane(const SlaneTask *d_descs, int nsg_used, int nsg, int ntasks, hipStream_t s);
hipError_t launch_slane_vproj(const uint32_t *d_planes, const int32_t *d_tasks, int ntasks, int nsg_used, int nsg, int NS,
                              int cols, int nrec, uint32_t *d_vproj, hipStream_t s);
hipError_t launch_slane_stddev(const uint32_t *d_vproj, const uint32_t *d_hproj, int A, int cols, int rows, int hrows_per_cand,
                               int hrow0, int nsg_used, int nsg, int nscans, double *d_v_sd, double *d_h_sd, hipStream_t s);
}  // namespace omr
