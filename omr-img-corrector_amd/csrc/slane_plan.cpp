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

void slane_guard_need(const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0, int rows, int cols, int *gx,
                      int *gy)
{
    int mnx = 0, mxx = cols - 1, mny = 0, mxy = rows - 1;
    const int xs[2] = {0, cols - 1}, ys[2] = {0, rows - 1};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            const int sx = (x0[ys[j]] + ad[xs[i]]) >> 10, sy = (y0[ys[j]] + bd[xs[i]]) >> 10;
            mnx = std::min(mnx, sx), mxx = std::max(mxx, sx), mny = std::min(mny, sy), mxy = std::max(mxy, sy);
        }
    const int64_t ox = std::max<int64_t>(-(int64_t)mnx, (int64_t)mxx - (cols - 1)), oy = std::max<int64_t>(-(int64_t)mny, (int64_t)mxy - (rows - 1));
    // (a map that samples far outside the image -- a magnifying or translating matrix -- gets the cap: what lies beyond
    // the guard is swept as white runs)
    *gx = (int)std::min<int64_t>((ox + 31) / 32 + 2, 64);
    *gy = (int)std::min<int64_t>(oy + 2, 2048);
}

// The empty program: every word = one white run of 32 bits (first segment: the zero register read straight), the other slots no-ops (0 bits
// shifted in from the zero register), nothing to fetch, dummy commits.
void slane_null_program(int nrec, int cls, uint32_t *seg, uint32_t *fet)
{
    const int RD = slane_seg_dwords(cls), S = slane_slots(cls);
    const uint32_t white = ((uint32_t)(SL_ZERO + 1) << 5) | SL_PK_MODE | (1u << SL_NSHIFT), pad = ((uint32_t)SL_ZERO << 5) | SL_PK_MODE;
    for (size_t q = 0; q < (size_t)nrec; q++)
        for (int k = 0; k < SL_K; k++)
            for (int j = 0; j < S; j++) seg[q * RD + (size_t)k * S + j] = j == 0 ? white : pad;
    const uint32_t nocommit = (uint32_t)SL_DUMMY | SL_COMMIT_MODE;
    for (size_t q = 0; q < (size_t)nrec; q++) {
        uint32_t *f = fet + q * SL_FREC;
        f[0] = f[1] = 0u;                  // two dummy pairs
        f[2] = nocommit | (nocommit << 16);
        f[3] = 1u;                         // turn header: one segment per word
    }
}

namespace {

struct Seg {
    int32_t s;     // source row; -1 = a white run (reads the zero register)
    int32_t src;   // source column of the run's first bit
    int32_t len;   // bits
};
constexpr int MAXSEG = 8;

inline int floor_div32(int v) { return v >= 0 ? v >> 5 : -((-v + 31) >> 5); }

// The runs of destination word w of row r in increasing bit order; they cover all 32 bits (a white run -- columns that do not
// exist, samples beyond the guard -- reads the zero register; format v4: the first segment is read straight into the word,
// so nothing below it is zero by itself).  Returns the number of runs, -1 when there are more than MAXSEG.
inline int word_segments(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, int32_t X0, int32_t Y0, int w, Seg *out)
{
    int n = 0;
    int cur_s = -2, cur_base = 0;  // run under construction: source row (-1 white), sx - i
    for (int i = 0; i < 32; i++) {
        const int x = 32 * w - g.off + i;
        int sy = -1, base = 0;
        if (w < g.NWd && x >= 0 && x < g.cols) {
            const int sx = (X0 + ad[x]) >> 10, yy = (Y0 + bd[x]) >> 10;
            // inside the image, or inside its zero guard (BORDER_CONSTANT white = the guard's zeros, read through the
            // same run); beyond the guard: a white run.  Source rows are kept as row + gy (never negative).
            if (sx >= -32 * (g.gx - 1) && sx < g.cols + 32 * (g.gx - 1) && yy >= -g.gy && yy < g.rows + g.gy) sy = yy + g.gy, base = sx - i;
        }
        if (n > 0 && cur_s == sy && (sy < 0 || cur_base == base)) {
            out[n - 1].len++;
        } else {
            if (n == MAXSEG) return -1;
            out[n].s = sy, out[n].src = sy < 0 ? 0 : base + i, out[n].len = 1;
            cur_s = sy, cur_base = base;
            n++;
        }
    }
    return n;
}

// word columns a run touches: lo always (its register is addressed), hi when one of its bits comes from it
inline void seg_columns(const Seg &q, int &c, int &chi)
{
    c = floor_div32(q.src);
    const int sh = q.src - 32 * c;
    chi = (sh + q.len > 32) ? c + 1 : c;
}

}  // namespace

int slane_strip_segments(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip)
{
    std::vector<int16_t> cmin((size_t)g.rowsG, 32767), cmax((size_t)g.rowsG, -32768);
    Seg sg[MAXSEG];
    int most = 0;
    for (int r = 0; r < g.rows; r++)
        for (int k = 0; k < SL_K; k++) {
            const int n = word_segments(g, ad, bd, x0[r], y0[r], strip * SL_K + k, sg);
            if (n < 0) return -1;
            if (n > most) most = n;
            for (int j = 0; j < n; j++) {
                if (sg[j].s < 0) continue;
                int c, chi;
                seg_columns(sg[j], c, chi);
                if (c < cmin[(size_t)sg[j].s]) cmin[(size_t)sg[j].s] = (int16_t)c;
                if (chi > cmax[(size_t)sg[j].s]) cmax[(size_t)sg[j].s] = (int16_t)chi;
            }
        }
    for (int s = 0; s < g.rowsG; s++)
        if (cmax[(size_t)s] >= cmin[(size_t)s] && cmax[(size_t)s] - cmin[(size_t)s] + 1 > SL_RING_COLS) return -1;
    return most;
}

bool slane_strip_program(const SlaneGeom &g, const int32_t *ad, const int32_t *bd, const int32_t *x0, const int32_t *y0,
                         int strip, int cls, uint32_t *seg, uint32_t *fet)
{
    const int R = g.rows, RD = slane_seg_dwords(cls), S = slane_slots(cls), NREC = slane_exec_records(R);
    std::vector<Seg> segs((size_t)R * SL_K * MAXSEG);
    std::vector<uint8_t> nseg((size_t)R * SL_K);
    const int RG = g.rowsG;  // source rows incl. the guard, indexed row + gy
    std::vector<int16_t> cmin((size_t)RG, 32767), cmax((size_t)RG, -32768);
    std::vector<int32_t> first((size_t)RG, INT32_MAX), last((size_t)RG, -1);
    for (int r = 0; r < R; r++)
        for (int k = 0; k < SL_K; k++) {
            Seg *sg = &segs[((size_t)r * SL_K + k) * MAXSEG];
            const int n = word_segments(g, ad, bd, x0[r], y0[r], strip * SL_K + k, sg);
            if (n < 0 || n > S) return false;
            nseg[(size_t)r * SL_K + k] = (uint8_t)n;
            for (int j = 0; j < n; j++) {
                if (sg[j].s < 0) continue;
                int c, chi;
                seg_columns(sg[j], c, chi);
                const size_t s = (size_t)sg[j].s;
                if (c < cmin[s]) cmin[s] = (int16_t)c;
                if (chi > cmax[s]) cmax[s] = (int16_t)chi;
                if (r < first[s]) first[s] = r;
                if (r > last[s]) last[s] = r;
            }
        }
    slane_null_program(NREC, cls, seg, fet);  // every word one white run, nothing fetched: the real content follows
    // ---- fetch schedule.  Record q (row q - SL_PRE) issues its loads while its row is swept; they are committed to
    // the ring before row q + SL_AHEAD (in between they sit in the wave's landing registers).  Source row s lives in
    // registers (s & 15) * 4 + j: it may be committed only after the last row that reads source row s - 16 is done.
    std::vector<uint8_t> used((size_t)NREC, 0);                                       // fetch PAIRS taken per record
    std::vector<uint8_t> freg((size_t)NREC * SL_PAIRS, (uint8_t)SL_DUMMY);            // (even) ring register of every pair
    for (int s = 0; s < RG; s++) {
        if (last[(size_t)s] < 0) continue;
        const int ncols = cmax[(size_t)s] - cmin[(size_t)s] + 1;
        if (ncols > SL_RING_COLS) return false;
        if (cmin[(size_t)s] < -g.gx || cmax[(size_t)s] >= g.NW + g.gx) return false;
        int lb = -SL_PRE;  // earliest record (as a row number) that may carry this source row
        if (s >= SL_RING_ROWS && last[(size_t)(s - SL_RING_ROWS)] >= 0)
            lb = std::max(lb, last[(size_t)(s - SL_RING_ROWS)] - SL_AHEAD + 1);
        int rec = first[(size_t)s] - SL_AHEAD;
        for (int pr = (ncols + 1) / 2 - 1; pr >= 0; pr--) {  // word columns 2 pr, 2 pr + 1 of the row: one pair
            while (rec >= lb && used[(size_t)(rec + SL_PRE)] == SL_PAIRS) rec--;
            if (rec < lb) return false;
            const size_t q = (size_t)(rec + SL_PRE);
            const int u = used[q]++;
            freg[q * SL_PAIRS + (size_t)u] = (uint8_t)((s & (SL_RING_ROWS - 1)) * SL_RING_COLS + 2 * pr);
            // (a pair's second entry is fetched needed or not: of one or two columns it lands in a ring register no segment
            // reads.  A row of THREE columns -- the rule -- loads its second pair one column to the left, (column 1, column 2):
            // its registers hold columns 0, 1, 1, 2 (slane_ring_register) and the load that would have fetched a column nobody
            // reads asks for the lines its neighbour has just requested -- a quarter less traffic from L2.)
            fet[q * SL_FREC + (size_t)u] = (uint32_t)(g.entry(s - g.gy, cmin[(size_t)s] + 2 * pr - (ncols == 3 && pr == 1 ? 1 : 0)) << 8);
        }
    }
    for (int q = SL_AHEAD; q < NREC; q++) {  // what row q commits = what row q - SL_AHEAD fetched
        const uint8_t *fr = &freg[(size_t)(q - SL_AHEAD) * SL_PAIRS];
        fet[(size_t)q * SL_FREC + 2] = ((uint32_t)fr[0] | SL_COMMIT_MODE) | (((uint32_t)fr[1] | SL_COMMIT_MODE) << 16);
    }
    // ---- the rows' words
    for (int r = 0; r < R; r++)
        for (int k = 0; k < SL_K; k++) {
            const Seg *sg = &segs[((size_t)r * SL_K + k) * MAXSEG];
            uint32_t *w = seg + (size_t)(r + SL_PRE) * RD + (size_t)k * S;
            const int n = nseg[(size_t)r * SL_K + k];
            for (int j = 0; j < n; j++) {
                uint32_t idx = SL_ZERO, sh = 0;
                if (sg[j].s >= 0) {
                    const int c = floor_div32(sg[j].src);
                    sh = (uint32_t)(sg[j].src - 32 * c);
                    const size_t s = (size_t)sg[j].s;
                    idx = (uint32_t)((sg[j].s & (SL_RING_ROWS - 1)) * SL_RING_COLS +
                                     slane_ring_register(c - cmin[s], cmax[s] - cmin[s] + 1, sh + (uint32_t)sg[j].len > 32u));
                }
                if (j == 0) slane_first_segment(idx, sh, sg[j].len);  // read straight into the word: top-aligned by the read itself
                w[j] = sh | (idx << 5) | SL_PK_MODE;
                if (j > 0) w[j - 1] |= (uint32_t)sg[j].len << SL_QSHIFT;  // a funnel shift's amount rides in the slot before it
            }
            w[0] |= (uint32_t)n << SL_NSHIFT;
        }
    // ---- turn headers: the most segments any word of a turn of SL_TURN records needs (virtual rows: one white run)
    for (int q0 = 0; q0 < NREC; q0 += SL_TURN) {
        uint32_t most = 1;
        for (int q = q0; q < q0 + SL_TURN && q < NREC; q++)
            for (int k = 0; k < SL_K; k++) most = std::max(most, (seg[(size_t)q * RD + (size_t)k * S] >> SL_NSHIFT) & 15u);
        for (int q = q0; q < q0 + SL_TURN && q < NREC; q++) fet[(size_t)q * SL_FREC + 3] = most;
    }
    return true;
}

// The units of a launch (unit u = chunk (u / ncq) of SL_CHUNK candidates in launch order x (strip group, scan-group group) u % ncq)
// dealt to the 8 XCDs in slots of SL_SLOT workgroups (a quarter of a unit): heaviest first, each item to the XCD with the least
// work so far.  A full chunk's unit is ONE item -- its four quarters stay on one XCD, that is the point of a unit --, the
// quarters of the last, partial chunk are dealt one by one: they even out what the whole units left (8 000 workgroups of a
// 512-scan A4 launch: 1 000 per XCD; whole and half units gave 992 / 1 008 and four XCDs idle for the last 0.4 ms).
// chunk_weight[c] = the work of a workgroup of chunk c (any unit of measure), chunk_size[c] = its candidates.
// Returns the table [8][per_xcd] of unit * 4 + quarter (-1 = none).
std::vector<int32_t> slane_deal_units(const std::vector<double> &chunk_weight, const std::vector<int> &chunk_size, int ncq, int *per_xcd)
{
    static_assert(SL_CHUNK == 4 * SL_SLOT, "a unit is four slots");
    struct Item {
        int unit, q0, nq;
        double w;
    };
    std::vector<Item> items;
    const int nchunks = (int)chunk_weight.size();
    for (int c = 0; c < nchunks; c++)
        for (int cq = 0; cq < ncq; cq++) {
            const int u = c * ncq + cq, n = chunk_size[(size_t)c];
            if (n == SL_CHUNK) items.push_back({u, 0, 4, chunk_weight[(size_t)c] * n});
            else
                for (int q = 0; q * SL_SLOT < n; q++) items.push_back({u, q, 1, chunk_weight[(size_t)c] * std::min(SL_SLOT, n - q * SL_SLOT)});
        }
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.w > b.w; });
    std::vector<std::vector<int32_t>> mine(8);
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Item &it : items) {
        int best = 0;
        for (int x = 1; x < 8; x++)
            if (load[x] < load[best]) best = x;
        for (int q = 0; q < it.nq; q++) mine[(size_t)best].push_back(it.unit * 4 + it.q0 + q);
        load[best] += it.w;
    }
    size_t most = 0;
    for (auto &m : mine) most = std::max(most, m.size());
    std::vector<int32_t> tab(8 * most, -1);
    for (int x = 0; x < 8; x++)
        for (size_t k = 0; k < mine[(size_t)x].size(); k++) tab[(size_t)x * most + k] = mine[(size_t)x][k];
    *per_xcd = (int)most;
    return tab;
}

}  // namespace omr
