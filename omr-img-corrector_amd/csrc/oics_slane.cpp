// oics_slane.cpp -- C ABI of the scan-lane sweep (slane.hpp; include/omrdeskew.h "scan-lane sweep").
#include <math.h>

#include "../../include/omrdeskew.h"
#include "engine.hpp"
#include "slane.hpp"

using namespace omr;

extern "C" {

// The program of one (matrix, strip) on the HOST (no GPU involved): what the waves of the scan-lane sweep
// execute, for inspection and for tests/ (the CPU interpreter in oracle/ runs it against the oracle's sweep).
int omr_slane_strip_program(int32_t rows, int32_t cols, const double *fwd_M, int32_t strip, uint32_t *seg_out,
                            uint32_t *fetch_out, int32_t *seg_dwords_per_row, int32_t *n_records, int32_t *pre_rows,
                            int32_t *most_segments, int32_t *guard_cols, int32_t *guard_rows)
{
    if (rows <= 0 || cols <= 0 || rows >= 32767 || cols >= 32767) return fail(OMR_ERR_ASSERT, "bad image size %dx%d", cols, rows);
    if (!fwd_M) return fail(OMR_ERR_BADARG, "null matrix");
    for (int k = 0; k < 6; k++)
        if (!isfinite(fwd_M[k])) return fail(OMR_ERR_BADARG, "matrix is not finite");
    SlaneGeom g;
    g.set(rows, cols);
    if (strip < 0 || strip >= g.NS) return fail(OMR_ERR_BADARG, "strip %d out of range (%d strips)", strip, g.NS);
    double Minv[6];
    invert_affine(fwd_M, Minv);
    std::vector<int32_t> ad, bd, x0, y0;
    slane_host_tables(Minv, rows, cols, ad, bd, x0, y0);
    int gx = 0, gy = 0;
    slane_guard_need(ad.data(), bd.data(), x0.data(), y0.data(), rows, cols, &gx, &gy);
    g.set(rows, cols, gx, gy);
    if (guard_cols) *guard_cols = g.gx;
    if (guard_rows) *guard_rows = g.gy;
    const int most = slane_strip_segments(g, ad.data(), bd.data(), x0.data(), y0.data(), strip);
    if (most_segments) *most_segments = most;
    if (most < 0) return fail(OMR_ERR_NOTIMPL, "strip %d does not fit the scan-lane scheme (segments / ring columns)", strip);
    const int cls = slane_class(most);
    if (seg_dwords_per_row) *seg_dwords_per_row = slane_seg_dwords(cls);
    if (n_records) *n_records = slane_exec_records(rows);
    if (pre_rows) *pre_rows = SL_PRE;
    if (!seg_out || !fetch_out) return OMR_OK;  // size query
    if (!slane_strip_program(g, ad.data(), bd.data(), x0.data(), y0.data(), strip, cls, seg_out, fetch_out))
        return fail(OMR_ERR_NOTIMPL, "strip %d: the ring schedule does not fit", strip);
    return OMR_OK;
}

}  // extern "C"
