// slane_plan.cpp -- host-side generator of the scan-lane sweep's programs (slane.hpp).
//
// A program is the geometry of one (candidate, strip) spelled out row by row: for every destination word the
// runs of consecutive source bits it is made of, straight from the integer tables of OpenCV's warpAffine
// (transfer.rs:459-486 -> imgwarp.cpp: X0 / Y0 per row, adelta / bdelta per column, >> AB_BITS) -- every
// destination pixel is evaluated with the reference's own integer expression and grouped, there is no
// threshold arithmetic, so the program is bit-exact by construction.  The generator also SIMULATES the wave's
// register ring: which source entry has to be fetched while which row is swept, and where it lives.
#include "slane.hpp"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace omr {

void slane_host_tables(const double M[6], int rows, int cols, std::vector<int32_t> &ad, std::vector<int32_t> &bd,
                       std::vector<int32_t> &x0, std::vector<int32_t> &y0)
{
    ad.resize((size_t)cols), bd.resize((size_t)cols), x0.resize((size_t)rows), y0.resize((size_t)rows);
    for (int i = 0; i < cols; i++) {
        const double x = (double)i;
        ad[(size_t)i] = (int32_t)rint(M[0] * x * 1024.0);
        bd[(size_t)i] = (int32_t)rint(M[3] * x * 1024.0);
    }
    for (int i = 0; i < rows; i++) {
        const double y = (double)i;
        x0[(size_t)i] = (int32_t)rint((M[1] * y + M[2]) * 1024.0) + 512;
        y0[(size_t)i] = (int32_t)rint((M[4] * y + M[5]) * 1024.0) + 512;
    }
}

namespace {

struct Seg {
    int32_t s;      // source row
    int32_t base;   // source column of destination bit 0 of the word (sx - i): 32 c + sh
    uint32_t mask;  // destination bits
};
constexpr int MAXSEG = 8;

inline int floor_div32(int v) { return v >= 0 ? v >> 5 : -((-v + 31) >> 5); }

// the segments of destination word w of row r; -1 when there are more than MAXSEG
inline int word_segments(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, int32_t X0, int32_t Y0, int w, Seg *out)
{
    int n = 0;
    if (w >= g.NW) return 0;
    const int xe = g.cols - 32 * w < 32 ? g.cols - 32 * w : 32;
    for (int i = 0; i < xe; i++) {
        const int x = 32 * w + i;
        const int sx = (X0 + ad[x]) >> 10, sy = (Y0 + bd[x]) >> 10;
        if ((unsigned)sx >= (unsigned)g.cols || (unsigned)sy >= (unsigned)g.rows) continue;  // border: white
        const int base = sx - i;
        int j = n - 1;
        while (j >= 0 && !(out[j].s == sy && out[j].base == base)) j--;
        if (j < 0) {
            if (n == MAXSEG) return -1;
            j = n++;
            out[j].s = sy, out[j].base = base, out[j].mask = 0;
        }
        out[j].mask |= 1u << i;
    }
    return n;
}

// word columns a segment touches: lo always (its register is addressed), hi when a selected bit comes from it
inline void seg_columns(const Seg &q, int &c, int &chi)
{
    c = floor_div32(q.base);
    const int sh = q.base - 32 * c;
    const int top = 31 - __builtin_clz(q.mask);
    chi = (sh + top >= 32) ? c + 1 : c;
}

}  // namespace

int slane_strip_segments(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip)
{
    std::vector<int16_t> cmin((size_t)g.rows, 32767), cmax((size_t)g.rows, -32768);
    Seg sg[MAXSEG];
    int most = 0;
    for (int r = 0; r < g.rows; r++)
        for (int k = 0; k < SL_K; k++) {
            const int n = word_segments(g, ad, bd, x0[r], y0[r], strip * SL_K + k, sg);
            if (n < 0) return -1;
            if (n > most) most = n;
            for (int j = 0; j < n; j++) {
                int c, chi;
                seg_columns(sg[j], c, chi);
                if (c < cmin[(size_t)sg[j].s]) cmin[(size_t)sg[j].s] = (int16_t)c;
                if (chi > cmax[(size_t)sg[j].s]) cmax[(size_t)sg[j].s] = (int16_t)chi;
            }
        }
    for (int s = 0; s < g.rows; s++)
        if (cmax[(size_t)s] >= cmin[(size_t)s] && cmax[(size_t)s] - cmin[(size_t)s] + 1 > SL_RING_COLS) return -1;
    return most;
}

bool slane_strip_program(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip, int cls, uint32_t *seg, uint32_t *fet)
{
    const int R = g.rows, RD = slane_seg_dwords(cls), S = slane_slots(cls), NREC = slane_records(R);
    std::vector<Seg> segs((size_t)R * SL_K * MAXSEG);
    std::vector<uint8_t> nseg((size_t)R * SL_K);
    std::vector<int16_t> cmin((size_t)R, 32767), cmax((size_t)R, -32768);
    std::vector<int32_t> first((size_t)R, INT32_MAX), last((size_t)R, -1);
    for (int r = 0; r < R; r++)
        for (int k = 0; k < SL_K; k++) {
            Seg *sg = &segs[((size_t)r * SL_K + k) * MAXSEG];
            const int n = word_segments(g, ad, bd, x0[r], y0[r], strip * SL_K + k, sg);
            if (n < 0 || n > S) return false;
            nseg[(size_t)r * SL_K + k] = (uint8_t)n;
            for (int j = 0; j < n; j++) {
                int c, chi;
                seg_columns(sg[j], c, chi);
                const size_t s = (size_t)sg[j].s;
                if (c < cmin[s]) cmin[s] = (int16_t)c;
                if (chi > cmax[s]) cmax[s] = (int16_t)chi;
                if (r < first[s]) first[s] = r;
                if (r > last[s]) last[s] = r;
            }
        }
    // ---- fetch schedule.  Record q (row q - SL_PRE) issues its loads while its row is swept; they are committed to
    // the ring before row q + SL_AHEAD (in between they sit in the wave's landing registers).  Source row s lives in
    // registers (s & 15) * 4 + j: it may be committed only after the last row that reads source row s - 16 is done.
    std::vector<uint8_t> used((size_t)NREC, 0);
    for (size_t i = 0; i < (size_t)NREC * RD; i += 2) seg[i] = 0u, seg[i + 1] = SL_PK_MODE;  // empty slots
    for (size_t i = 0; i < (size_t)NREC * SL_FREC; i++) fet[i] = (i % SL_FREC) < SL_FETCH ? 0u : (SL_DUMMY | SL_COMMIT_MODE);
    std::vector<uint8_t> freg((size_t)NREC * SL_FETCH, (uint8_t)SL_DUMMY);  // ring register of every fetch
    for (int s = 0; s < R; s++) {
        if (last[(size_t)s] < 0) continue;
        const int ncols = cmax[(size_t)s] - cmin[(size_t)s] + 1;
        if (ncols > SL_RING_COLS) return false;
        if (cmin[(size_t)s] < -SL_GX || cmax[(size_t)s] >= g.NW + SL_GX) return false;
        int lb = -SL_PRE;  // earliest record (as a row number) that may carry this source row
        if (s >= SL_RING_ROWS && last[(size_t)(s - SL_RING_ROWS)] >= 0)
            lb = std::max(lb, last[(size_t)(s - SL_RING_ROWS)] - SL_AHEAD + 1);
        int rec = first[(size_t)s] - SL_AHEAD;
        for (int j = ncols - 1; j >= 0; j--) {
            while (rec >= lb && used[(size_t)(rec + SL_PRE)] == SL_FETCH) rec--;
            if (rec < lb) return false;
            const size_t q = (size_t)(rec + SL_PRE);
            const uint32_t reg = (uint32_t)((s & (SL_RING_ROWS - 1)) * SL_RING_COLS + j);
            freg[q * SL_FETCH + used[q]] = (uint8_t)reg;
            fet[q * SL_FREC + used[q]++] = (uint32_t)(g.entry(s, cmin[(size_t)s] + j) << 8);
        }
    }
    for (int q = SL_AHEAD; q < NREC; q++)  // what row q commits = what row q - SL_AHEAD fetched
        for (int f = 0; f < SL_FETCH; f++) fet[(size_t)q * SL_FREC + SL_FETCH + f] = freg[(size_t)(q - SL_AHEAD) * SL_FETCH + f] | SL_COMMIT_MODE;
    // ---- the rows' words
    for (int r = 0; r < R; r++)
        for (int k = 0; k < SL_K; k++) {
            const Seg *sg = &segs[((size_t)r * SL_K + k) * MAXSEG];
            uint32_t *w = seg + (size_t)(r + SL_PRE) * RD + (size_t)k * 2 * S;
            w[1] |= (uint32_t)nseg[(size_t)r * SL_K + k] << 24;
            for (int j = 0; j < nseg[(size_t)r * SL_K + k]; j++) {
                const int c = floor_div32(sg[j].base), sh = sg[j].base - 32 * c;
                const size_t s = (size_t)sg[j].s;
                const uint32_t idx = (uint32_t)((sg[j].s & (SL_RING_ROWS - 1)) * SL_RING_COLS + (c - cmin[s]));
                w[2 * j] = sg[j].mask;
                w[2 * j + 1] |= (uint32_t)sh | (idx << 5);
            }
        }
    return true;
}

}  // namespace omr
