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
//                scan group; entry 0 is all zero (dummy fetches); a zero guard of gx word columns and gy rows all round.
//                The images of the scan groups lie group_stride() bytes apart -- a power of two -- from a base aligned to it,
//                so no image straddles a 4 GB boundary: the kernel forms a load's address by ONE 32-bit add to the image's
//                base (below)
//   strip      : SL_K = 2 adjacent destination word columns; a wave owns (scan group, candidate, strip) and
//                walks ALL destination rows top to bottom
//   ring       : the wave keeps the source entries it needs in 64 VGPRs: source row s owns registers (s & 15) * 4 .. + 3,
//                its word columns from the row's own column base cb(s) on (slane_ring_register: a row of three columns keeps
//                its middle column twice)
//   words      : destination word w covers destination columns 32 w - off .. 32 w - off + 31 with off = (32 - cols % 32)
//                % 32 (slane_grid_offset): the row's LAST column is the last bit of the last word, the columns that do not
//                exist are the first bits of word 0 -- a white run, one segment.  When that would leave word 0 with 29 bits or
//                more (a word that can need all 8 slots for itself at 10 degrees) the grid is moved 16 columns further: word 0
//                and the last word are both short then, for one more word per row
//   records    : one per destination row, plus SL_PRE virtual rows ahead of row 0 that only fill the ring and virtual
//                rows at the end up to a whole turn (slane_exec_records; they are numbered so that the last is a multiple of 64).  Two streams per strip (format v3, round 5):
//                  fetch stream, 4 dwords per row (format v4): 2 x (E << 8) = the byte offsets of the first entries of the two
//                                   PAIRS whose loads are ISSUED while this row is swept (0 = the all-zero entry: a dummy) -- a
//                                   pair = entries E and E + 1, two adjacent word columns of one source row, into an aligned
//                                   pair of landing registers.  The kernel adds the offset to the base of a copy of the
//                                   image's buffer descriptor (one s_add_u32 per pair) and issues the two loads with the
//                                   CONSTANT offsets 0 and 256: a vector-memory instruction whose offset comes out of an SGPR
//                                   costs an issue turn more than one with a constant offset, 2.7 ms per launch
//                                   (profiles/r05_lanes_ablation.md) --, then 1
//                                   dword = 2 x u16 (ring register | 0x8000) to COMMIT before this row: where the two pairs
//                                   issued SL_AHEAD rows earlier belong -- an EVEN ring register: one v_mov_b64 with DST_REL
//                                   moves a pair (an odd index is rounded down by the hardware, tools/mov64_probe.hip);
//                                   register SL_DUMMY = 66 swallows a dummy pair; 0x8000 = M0's DST_REL bit --, then 1 dword
//                                   = the TURN HEADER: the most segments any word of the 16 rows from this record on needs
//                                   (1 .. 8; the kernel reads it at the first record of a turn and executes exactly that
//                                   many slots per word).  One s_load_dwordx4 per row
//                  segment stream, SL_K words x S dwords per row.  A word is assembled from its segments in
//                                   increasing bit order by funnel shifts, no masks (format v4):
//                                     first segment : D = (ring[i] : ring[i - 1]) >> sh   read straight into the word so that
//                                                     its bits are TOP-aligned (whatever lies below is pushed out by the
//                                                     shifts that follow: the segments of a word cover all 32 bits); i = the
//                                                     register index + 1: register -1 (a landing register) may be named
//                                                     when all the segment's bits come from ring register 0
//                                     others        : X = (ring[idx + 1] : ring[idx]) >> sh   (the segment's bits, bit 0 first)
//                                                     D = (X : D) >> q                        (q = its length: X's low bits
//                                                                                              enter at the top)
//                                   White runs (columns that do not exist, samples beyond the guard) are segments that read
//                                   ring register 64, which holds 0.
//                                   pk = sh | idx << 5 | 0x60000 | q' << 21 (pk >> 5 is M0 for the indexed v_alignbit:
//                                   index + SRC0_REL | SRC1_REL; pk itself is its shift operand); q' = the funnel shift of
//                                   the NEXT slot (pk >> 21 of slots 2 p and 2 p + 1 come out of one s_lshr_b64 and serve
//                                   slots 2 p + 1 and 2 p + 2); the first pk of a word also carries its segment count
//                                   n << 26 (n >= 1)
//                S = 2 / 4 / 8 slots per word laid out (4 / 8 / 16 dwords per row), 2 .. 8 executed, chosen per strip
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace omr {

constexpr int SL_K = 2;
constexpr int SL_RING_ROWS = 16;
constexpr int SL_RING_COLS = 4;
constexpr int SL_FETCH = 4;                             // loads (= landing registers) per row: SL_PAIRS pairs of adjacent entries
constexpr int SL_FREC = 4;                              // dwords per row of the fetch stream
constexpr int SL_AHEAD = 4;                             // rows between a load and its commit
constexpr int SL_ZERO = SL_RING_ROWS * SL_RING_COLS;   // ring register that holds 0 (white runs read it)
constexpr int SL_DUMMY = SL_ZERO + 2;                   // (even) register pair that swallows dummy fetch pairs
constexpr int SL_PAIRS = 2;                             // fetch pairs per row
constexpr int SL_TURN = 16;                             // rows per turn of the kernel's loop: one header per turn
constexpr uint32_t SL_PK_MODE = 0x60000u;               // pk >> 5 -> M0[13:12]: SRC0_REL | SRC1_REL
constexpr uint32_t SL_COMMIT_MODE = 0x8000u;            // M0[15]: DST_REL
constexpr int SL_PRE = 24;                              // virtual rows ahead of row 0
constexpr int SL_QSHIFT = 21, SL_NSHIFT = 26;           // pk fields: second shift, segment count
constexpr uint32_t SL_SHORT = 0x80000000u;              // first pk of a word: at most four segments (no longer read)
constexpr int SL_GX = 4;                                // least zero guard, word columns on either side
constexpr int SL_LANES = 64;
#ifndef SL_CHUNK_CANDIDATES
#define SL_CHUNK_CANDIDATES 32                          // (development builds try 16 / 64: profiles/r05_lanes_ablation.md)
#endif
constexpr int SL_CHUNK = SL_CHUNK_CANDIDATES;           // candidates of a unit of slane_kernel's launch order (one workgroup each, on one XCD)
constexpr int SL_SLOT = SL_CHUNK / 4;                   // workgroups per slot of an XCD's list of units (a quarter unit: slane_deal_units)
#ifndef SL_BLOCK_ROWS
#define SL_BLOCK_ROWS 64                                // (development builds try 16 / 32 with SLANE_BLOCK of tools/gen_slane_asm.py)
#endif
constexpr int SL_BLOCK = SL_BLOCK_ROWS;                 // rows between two meetings of a workgroup (row counts leave its LDS)
constexpr int SL_DUMP = 13;                             // registers a wave dumps per word: planes p0..p12 (the parked carries of the
                                                        // carry-save tree are spent at the end: the records are a multiple of 64)
constexpr int SL_MAX_RECORDS = 1 << SL_DUMP;            // a column count must fit the planes

// columns that do not exist in front of destination column 0 (see "words" above)
inline int slane_grid_offset(int cols)
{
    const int off = (32 - cols % 32) % 32;
    return off >= 1 && off <= 3 ? off + 16 : off;
}

struct SlaneGeom {
    int rows = 0, cols = 0;  // image
    int off = 0;             // columns that do not exist at the start of destination word 0
    int NW = 0;              // source word columns
    int NWd = 0;             // destination words per row: (cols + off + 31) / 32
    int gx = SL_GX, gy = 0;  // zero guard around the bit image: word columns left / right, rows above / below.  A sample
                             // that falls outside the image (warpAffine: BORDER_CONSTANT white) reads the guard's zeros
                             // through the same run as its neighbours, so border words need no extra segments
    int colsG = 0, rowsG = 0;
    int NS = 0;              // strips
    int64_t entries = 0;     // 1 + rowsG * colsG (+ 1 spare in memory: the partner of a pair that starts at the last entry)
    void set(int r, int c, int guard_cols = SL_GX, int guard_rows = 0)
    {
        rows = r, cols = c, NW = (c + 31) / 32, off = slane_grid_offset(c), NWd = (c + off + 31) / 32, NS = (NWd + SL_K - 1) / SL_K;
        gx = guard_cols < SL_GX ? SL_GX : guard_cols, gy = guard_rows < 0 ? 0 : guard_rows;
        colsG = NW + 2 * gx, rowsG = rows + 2 * gy;
        entries = 1 + (int64_t)rowsG * colsG;
    }
    int64_t entry(int s, int c) const { return 1 + (int64_t)(s + gy) * colsG + (c + gx); }
    size_t image_bytes() const { return (size_t)(entries + 1) * SL_LANES * 4; }  // per scan group
    size_t group_stride() const  // bytes between the images of two scan groups: the power of two that holds one
    {
        size_t st = 1 << 20;
        while (st < image_bytes()) st <<= 1;
        return st;
    }
};

// how far the samples of a candidate reach outside the image: word columns / rows of guard it needs (from the four
// corners: the fixed-point map is monotone in x and in y)
void slane_guard_need(const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0, int rows, int cols, int *gx,
                      int *gy);

struct SlaneStrip {     // per (candidate, strip)
    int64_t seg_offset; // of the segment stream, in dwords from the program base (a multiple of 64)
    int64_t fet_offset; // of the fetch stream
    int32_t cls;        // slot class (slane_slots / slane_exec_slots); -1 = the strip does not fit this scheme
    int32_t nseg;       // most segments any of its words needs
};

// Classes: the segment stream of a strip is LAID OUT with 2, 4 or 8 slots per word (s_load sizes), the wave EXECUTES
// exactly as many slots as the strip's busiest word needs (a word of at most four segments still skips the rest):
//   class            0  1  2  3  4  5  6
//   slots laid out   2  4  8  4  8  8  8
//   slots executed   2  4  8  3  5  6  7
#if defined(__HIPCC__)
#define SL_HD __host__ __device__
#else
#define SL_HD
#endif
SL_HD inline int slane_slots(int cls) { return cls == 0 ? 2 : (cls == 1 || cls == 3) ? 4 : 8; }
SL_HD inline int slane_exec_slots(int cls) { return cls <= 2 ? 2 << cls : cls == 3 ? 3 : cls + 1; }
SL_HD inline int slane_seg_dwords(int cls) { return SL_K * slane_slots(cls); }  // per row
inline int slane_class(int most)
{
    return most <= 2 ? 0 : most == 3 ? 3 : most == 4 ? 1 : most <= 7 ? most - 1 : most == 8 ? 2 : -1;
}
// Where word column d (0 .. 3, counted from the row's first column) of a source row of ncols columns sits among the row's four
// ring registers, for a segment that starts in it (two_words: some of its bits come from column d + 1, so the register above
// must hold that column).  Rows of 1, 2 or 4 columns: register d.  Rows of THREE columns are loaded as the pairs (0, 1) and
// (1, 2) -- registers 0, 1, 2, 3 = columns 0, 1, 1, 2.
SL_HD inline int slane_ring_register(int d, int ncols, bool two_words)
{
    if (ncols != 3) return d;
    return d == 2 ? 3 : (d == 1 && two_words) ? 2 : d;
}
// The first segment of a word (format v4): the window that puts its len bits at the TOP of the word.  In: the ring register
// idx of the word column its first bit lies in and that bit's place sh; out: the slot's index field (register + 1: the
// kernel's instruction names the pair (index - 1, index)) and shift.
SL_HD inline void slane_first_segment(uint32_t &idx, uint32_t &sh, int len)
{
    const int shp = (int)sh - (32 - len);  // < 0: every bit comes from register idx itself (sh + len < 32): window (idx - 1, idx)
    if (shp < 0) sh = (uint32_t)(shp + 32);
    else sh = (uint32_t)shp, idx += 1;
}
// Records are NUMBERED up to a multiple of 64 -- slane_records: the kernel takes its row phases (LDS slot, carry-save level,
// flush) from the number of rows left, modulo 64, and the row counts are laid out by record number -- but only the last
// slane_exec_records of them EXIST (whole turns): a wave starts at record number slane_records - slane_exec_records, as if
// the records before it -- they would be virtual rows that fetch nothing -- had been swept already.
inline int slane_records(int rows) { return (SL_PRE + rows + 63) & ~63; }
inline int slane_exec_records(int rows) { return (SL_PRE + rows + SL_TURN - 1) & ~(SL_TURN - 1); }

// warpAffine's integer tables of one candidate on the host (the expressions of tables_kernel, kernels.hip;
// built -ffp-contract=off): adelta / bdelta per column, (X0, Y0) per row with round_delta = 512
void slane_host_tables(const double Minv[6], int rows, int cols, std::vector<int32_t> &ad, std::vector<int32_t> &bd,
                       std::vector<int32_t> &x0, std::vector<int32_t> &y0);

void slane_null_program(int nrec, int cls, uint32_t *seg, uint32_t *fet);  // empty words, nothing to fetch

// Pass 1: the most segments a word of strip `strip` needs, or -1 when the strip does not fit (a source row
// that needs more than SL_RING_COLS word columns, more than 8 segments).
int slane_strip_segments(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip);
// Pass 2: the strip's program: slane_records(rows) rows of slane_seg_dwords(cls) dwords into seg, of SL_FREC dwords
// into fet.  Returns false when the ring schedule fails (the candidate then stays with the run-merging kernel).
bool slane_strip_program(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip, int cls, uint32_t *seg, uint32_t *fet);

// What a wave of slane_kernel starts from (read by scalar loads; tools/gen_slane_asm.py fixes the layout)
struct SlaneTask {
    uint64_t seg, fet;   // the strip's segment / fetch stream
    uint32_t hrsrc[4];   // buffer descriptor of the row counts of (candidate, scan group): [record / 2][scan], each u32 = the
                         // counts of records 2 i (low half) and 2 i + 1 (high half)
    uint32_t rsrc[4];    // buffer descriptor of the scan group's interleaved bit image
    uint32_t nrec;       // records as they are numbered (slane_records: a multiple of 64)
    uint32_t hpitch;     // bytes between pair rows of the row counts
    int32_t cls;         // slot class: slots laid out / executed per word (slane_slots, slane_exec_slots)
    int32_t wave;        // the pair rows of its scan group's LDS accumulators this wave flushes: first (bits 4:0, SL_WAVE_FIRST_BITS),
                         // their number 0 .. 8 (bits 11:8, SL_WAVE_COUNT_SHIFT) -- tools/gen_slane_asm.py reads the same fields
    uint64_t planes;     // counter dump of (task, scan group): [word][SL_DUMP planes][64]
    uint32_t lds_base;   // LDS address of the scan group's row-count accumulators in the workgroup
    uint32_t nexec;      // records the streams hold (slane_exec_records: whole turns, the last of the numbered ones)
    uint64_t pad2[6];
};
constexpr int SL_WAVE_FIRST_BITS = 5, SL_WAVE_COUNT_SHIFT = 8, SL_WAVE_COUNT_BITS = 4;
static_assert(sizeof(SlaneTask) == 128, "SlaneTask layout (slane_asm.inc loads it by offset)");
// the offsets the wave program loads from (tools/gen_slane_asm.py: s_load ... %[desc], <offset>)
static_assert(offsetof(SlaneTask, seg) == 0 && offsetof(SlaneTask, fet) == 8 && offsetof(SlaneTask, hrsrc) == 16 &&
                  offsetof(SlaneTask, rsrc) == 32 && offsetof(SlaneTask, nrec) == 48 && offsetof(SlaneTask, hpitch) == 52 &&
                  offsetof(SlaneTask, cls) == 56 && offsetof(SlaneTask, wave) == 60 && offsetof(SlaneTask, planes) == 64 &&
                  offsetof(SlaneTask, lds_base) == 72 && offsetof(SlaneTask, nexec) == 76,
              "SlaneTask field offsets are part of the wave program");

// slane_build.hip: the same programs generated on the device (the default; the host generator above is the reference
// implementation).  Scratch per task (= candidate * NS + strip): cmin / cmax / first / last [task][rowsG], most [task],
// used [task][nrec] (zeroed), freg [task][nrec][SL_FETCH] (filled with SL_DUMMY).
struct int2_t;
struct SlaneBuild {
    SlaneGeom g;
    int32_t nrec;
    const int32_t *adelta, *bdelta;  // [A][cols]
    const int2_t *xy0;               // [A][rows], round delta 512 included
    int32_t *cmin, *cmax, *first, *last, *most, *bad;
    uint8_t *used, *freg;
    uint32_t *prog;
    const int64_t *seg_off, *fet_off;  // [task], dwords from prog
    const int32_t *cls;                // [task]
    int64_t null_seg, null_fet;
};

}  // namespace omr

#include <hip/hip_runtime_api.h>
namespace omr {
hipError_t launch_slane_build_scan(const SlaneBuild &b, int ntasks, hipStream_t s);   // -> most, cmin, cmax, first, last
hipError_t launch_slane_build_emit(const SlaneBuild &b, int ntasks, hipStream_t s);   // -> prog (needs cls, seg_off, fet_off)
hipError_t launch_slane_pack(const uint8_t *d_img, int64_t scan_stride, int64_t step, const SlaneGeom &g, int nscans,
                             int black_max, uint32_t *d_bits, hipStream_t s);
// scans that arrive packed to 1 bit per pixel: [scan][rows][NW] dwords, bit i of word c = pixel 32 c + i is black
hipError_t launch_slane_pack_bits(const uint32_t *d_packed, int64_t scan_stride_dwords, const SlaneGeom &g, int nscans,
                                  uint32_t *d_bits, hipStream_t s);
// nsg_used = scan groups that carry scans in this launch, nsg = scan groups the scratch (and its descriptors) is laid out for
hipError_t launch_slane(const SlaneTask *d_descs, int nsgq, int nsgp, int A, int NQ, int sgw_log, int32_t *d_guard,
                        const int32_t *d_unit_tab, int per_xcd, hipStream_t s);
std::vector<int32_t> slane_deal_units(const std::vector<double> &chunk_weight, const std::vector<int> &chunk_size, int ncq, int *per_xcd);
hipError_t launch_slane_vproj(const uint32_t *d_planes, const int32_t *d_tasks, int ntasks, int nsg_used, int nsg, int NS,
                              int cols, int off, int nrec, uint16_t *d_vproj, uint32_t *d_total, hipStream_t s);
hipError_t launch_slane_stddev(const uint16_t *d_vproj, const uint32_t *d_hproj, const uint32_t *d_total, int A, int cols, int rows, int hpairs_per_cand,
                               int hrow0, int nsg_used, int nsg, int nscans, double *d_v_sd, double *d_h_sd, hipStream_t s);
}  // namespace omr
