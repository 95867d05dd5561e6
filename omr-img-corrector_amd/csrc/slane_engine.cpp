// slane_engine.cpp -- host side of the scan-lane sweep: the plan (every strip's program, generated on the device by
// slane_build.hip, or on the host's cores by slane_plan.cpp -- the reference implementation -- and uploaded), the per-launch scratch and the enqueue (slane.hpp).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "../../include/omrdeskew.h"
#include "engine.hpp"
#include "slane.hpp"

namespace omr {

static int host_threads()
{
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw ? hw : 4u, 16u));
}

static const char *kNoFit = "a candidate does not fit the scan-lane scheme (more than 8 segments per word, more than 4 word "
                            "columns per source row, or a ring schedule that does not fit 16 source rows)";

// the streams' places in the program buffer, from the strips' classes
void SlanePlan::layout()
{
    int64_t off = 0;
    for (auto &S : strips) {
        S.seg_offset = off;
        off += ((int64_t)nexec * slane_seg_dwords(S.cls) + 63) & ~63ll;
        S.fet_offset = off;
        off += ((int64_t)nexec * SL_FREC + 63) & ~63ll;
    }
    // the null program (class 0: nothing to fetch, empty words) for the places of a workgroup beyond the last strip
    null_seg = off;
    off += ((int64_t)nexec * slane_seg_dwords(0) + 63) & ~63ll;
    null_fet = off;
    off += ((int64_t)nexec * SL_FREC + 63) & ~63ll;
    prog_dwords = off + 2048;  // the kernel requests two records past a stream's last one (and may prefetch a turn or two beyond it)
}

// the device generator (slane_build.hip): scan -> classes and layout on the host -> emit
int SlanePlan::generate_on_device(const SweepTables &t)
{
    const size_t ntasks = (size_t)A * g.NS, nrow = ntasks * (size_t)g.rowsG;
    DevBuf cmin, cmax, first, last, most, bad, used, freg, d_seg, d_fet, d_cls;
    OMR_HIP(cmin.alloc(4 * nrow));
    OMR_HIP(cmax.alloc(4 * nrow));
    OMR_HIP(first.alloc(4 * nrow));
    OMR_HIP(last.alloc(4 * nrow));
    OMR_HIP(most.alloc(4 * ntasks));
    OMR_HIP(bad.alloc(4));
    OMR_HIP(hipMemset(most.p, 0, 4 * ntasks));
    OMR_HIP(hipMemset(bad.p, 0, 4));
    SlaneBuild b{};
    b.g = g;
    b.nrec = nexec;
    b.adelta = t.adelta.as<int32_t>();
    b.bdelta = t.bdelta.as<int32_t>();
    b.xy0 = t.xy0.as<int2_t>();
    b.cmin = cmin.as<int32_t>(), b.cmax = cmax.as<int32_t>(), b.first = first.as<int32_t>(), b.last = last.as<int32_t>();
    b.most = most.as<int32_t>(), b.bad = bad.as<int32_t>();
    OMR_HIP(launch_slane_build_scan(b, (int)ntasks, nullptr));
    std::vector<int32_t> hm(ntasks);
    OMR_HIP(hipMemcpy(hm.data(), most.p, 4 * ntasks, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ntasks; i++) {
        strips[i].nseg = hm[i] > 8 ? -1 : hm[i];
        strips[i].cls = hm[i] > 8 ? -1 : slane_class(hm[i]);
        if (strips[i].cls < 0) return fail(OMR_ERR_NOTIMPL, "%s", kNoFit);
    }
    layout();
    std::vector<int64_t> hs(ntasks), hf(ntasks);
    std::vector<int32_t> hc(ntasks);
    for (size_t i = 0; i < ntasks; i++) hs[i] = strips[i].seg_offset, hf[i] = strips[i].fet_offset, hc[i] = strips[i].cls;
    OMR_HIP(d_seg.alloc(8 * ntasks));
    OMR_HIP(d_fet.alloc(8 * ntasks));
    OMR_HIP(d_cls.alloc(4 * ntasks));
    OMR_HIP(hipMemcpy(d_seg.p, hs.data(), 8 * ntasks, hipMemcpyHostToDevice));
    OMR_HIP(hipMemcpy(d_fet.p, hf.data(), 8 * ntasks, hipMemcpyHostToDevice));
    OMR_HIP(hipMemcpy(d_cls.p, hc.data(), 4 * ntasks, hipMemcpyHostToDevice));
    OMR_HIP(used.alloc(ntasks * (size_t)nexec));
    OMR_HIP(freg.alloc(ntasks * (size_t)nexec * SL_FETCH));
    OMR_HIP(hipMemset(used.p, 0, ntasks * (size_t)nexec));
    OMR_HIP(hipMemset(freg.p, SL_DUMMY, ntasks * (size_t)nexec * SL_FETCH));
    OMR_HIP(prog.alloc(sizeof(uint32_t) * (size_t)prog_dwords));
    OMR_HIP(hipMemset(prog.p, 0, sizeof(uint32_t) * (size_t)prog_dwords));
    b.used = used.as<uint8_t>(), b.freg = freg.as<uint8_t>();
    b.prog = prog.as<uint32_t>();
    b.seg_off = d_seg.as<int64_t>(), b.fet_off = d_fet.as<int64_t>(), b.cls = d_cls.as<int32_t>();
    b.null_seg = null_seg, b.null_fet = null_fet;
    OMR_HIP(launch_slane_build_emit(b, (int)ntasks, nullptr));
    int32_t hb = 0;
    OMR_HIP(hipMemcpy(&hb, bad.p, 4, hipMemcpyDeviceToHost));  // (synchronises: the scratch is released on return)
    if (hb) return fail(OMR_ERR_NOTIMPL, "%s", kNoFit);
    return OMR_OK;
}

// the host generator (slane_plan.cpp, the reference implementation): pass 1, the class of every strip (most segments per
// word); pass 2, the streams; candidates dealt to the host's threads; one upload
int SlanePlan::generate_on_host(const SweepTables &t)
{
    const SweepDims &d = t.dims;
    const int NS = g.NS, T = host_threads();
    std::atomic<int> next{0};
    std::atomic<int> bad{0};
    auto pass = [&](bool emit, uint32_t *host_prog) {
        next = 0;
        std::vector<std::thread> pool;
        for (int th = 0; th < T; th++)
            pool.emplace_back([&, emit, host_prog]() {
                std::vector<int32_t> ad, bd, x0, y0;
                for (;;) {
                    const int a = next.fetch_add(1);
                    if (a >= A || bad.load()) return;
                    slane_host_tables(&t.host_minv[6 * (size_t)a], d.rows, d.cols, ad, bd, x0, y0);
                    for (int st = 0; st < NS; st++) {
                        SlaneStrip &S = strips[(size_t)a * NS + st];
                        if (!emit) {
                            S.nseg = slane_strip_segments(g, ad.data(), bd.data(), x0.data(), y0.data(), st);
                            S.cls = S.nseg < 0 ? -1 : slane_class(S.nseg);
                            if (S.cls < 0) bad = 1;
                        } else if (!slane_strip_program(g, ad.data(), bd.data(), x0.data(), y0.data(), st, S.cls,
                                                        host_prog + S.seg_offset, host_prog + S.fet_offset)) {
                            bad = 1;
                        }
                    }
                }
            });
        for (auto &th : pool) th.join();
    };
    pass(false, nullptr);
    if (bad.load()) return fail(OMR_ERR_NOTIMPL, "%s", kNoFit);
    layout();
    std::vector<uint32_t> host((size_t)prog_dwords, 0u);
    pass(true, host.data());
    slane_null_program(nexec, 0, host.data() + null_seg, host.data() + null_fet);
    if (bad.load()) return fail(OMR_ERR_NOTIMPL, "%s", kNoFit);
    OMR_HIP(prog.alloc(sizeof(uint32_t) * (size_t)prog_dwords));
    OMR_HIP(hipMemcpy(prog.p, host.data(), sizeof(uint32_t) * (size_t)prog_dwords, hipMemcpyHostToDevice));
    return OMR_OK;
}

int SlanePlan::build(const SweepTables &t, bool on_host)
{
    const SweepDims &d = t.dims;
    if (d.rows + SL_PRE + 1 >= SL_MAX_RECORDS) return fail(OMR_ERR_NOTIMPL, "scan-lane sweep: more than %d rows (%d counter planes)", SL_MAX_RECORDS - SL_PRE - 2, SL_DUMP);
    if (d.cols > 65535) return fail(OMR_ERR_NOTIMPL, "scan-lane sweep: more than 65535 columns (row counts travel as u16)");
    // the zero guard around the bit images: what the steepest candidate reaches outside the image
    int gx = 0, gy = 0;
    {
        std::vector<int32_t> ad, bd, x0, y0;
        for (int a = 0; a < d.A; a++) {
            int cx, cy;
            slane_host_tables(&t.host_minv[6 * (size_t)a], d.rows, d.cols, ad, bd, x0, y0);
            slane_guard_need(ad.data(), bd.data(), x0.data(), y0.data(), d.rows, d.cols, &cx, &cy);
            gx = std::max(gx, cx), gy = std::max(gy, cy);
        }
    }
    g.set(d.rows, d.cols, gx, gy);
    if (g.image_bytes() > 0xffffffffull)  // a pair's byte offset is a dword of the fetch stream, a scan group's image one buffer descriptor
        return fail(OMR_ERR_NOTIMPL, "scan-lane sweep: the bit image of a scan group (%zu bytes with its guard) exceeds 4 GB", g.image_bytes());
    A = d.A;
    nrec = slane_records(d.rows), nexec = slane_exec_records(d.rows), hrow0 = nrec - nexec + SL_PRE;
    const int NS = g.NS;
    strips.assign((size_t)A * NS, SlaneStrip{0, 0, -1, 0});
    if (int rc = on_host ? generate_on_host(t) : generate_on_device(t)) return rc;
    // ---- tasks: candidate-major, all strips of a candidate together (they share the candidate's row counts, see
    // slane_kernel); the candidates with the most segments per word (the steepest angles) go first, so that the
    // launch does not end on its longest tasks
    std::vector<int> order((size_t)A);
    for (int a = 0; a < A; a++) order[(size_t)a] = a;
    auto weight = [&](int a) {
        int w = 0;
        for (int st = 0; st < NS; st++) w += slane_exec_slots(strips[(size_t)a * NS + st].cls);
        return w;
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return weight(x) > weight(y); });
    {   // ... and the 32 candidates that share an XCD (a unit of slane_kernel's launch order) turn the same way: +theta and
        // -theta weigh the same but read source columns up to 2 rows tan(theta) apart, so a mixed chunk shares half as much
        // in that XCD's L2.  Blocks of 32 are taken in turn from the clockwise and the counter-clockwise candidates.
        std::vector<int> side[2], merged;
        for (int a : order) side[t.host_minv[6 * (size_t)a + 1] < 0.0 ? 1 : 0].push_back(a);
        size_t at[2] = {0, 0};
        while (at[0] < side[0].size() || at[1] < side[1].size()) {
            int pick = at[0] >= side[0].size() ? 1 : at[1] >= side[1].size() ? 0 : (weight(side[0][at[0]]) >= weight(side[1][at[1]]) ? 0 : 1);
            for (int k = 0; k < SL_CHUNK && at[pick] < side[pick].size(); k++) merged.push_back(side[pick][at[pick]++]);
        }
        order.swap(merged);
    }
    tasks.clear();
    for (int a : order)
        for (int st = 0; st < NS; st++) tasks.push_back(a * NS + st);
    chunk_weight.clear(), chunk_size.clear();
    for (size_t c0 = 0; c0 < order.size(); c0 += SL_CHUNK) {
        const size_t n = std::min<size_t>(SL_CHUNK, order.size() - c0);
        double w = 0;
        // (a workgroup's time ~ 5 + executed slots per word, measured: 0.70 ms at 2 slots, 1.21 ms at 7 -- tools/kstamps_lanes.py)
        for (size_t i = 0; i < n; i++) w += 5.0 * NS + weight(order[c0 + i]);
        chunk_weight.push_back(w / (double)n), chunk_size.push_back((int)n);
    }
    unit_tabs.clear();
    OMR_HIP(d_tasks.alloc(sizeof(int32_t) * tasks.size()));
    OMR_HIP(hipMemcpy(d_tasks.p, tasks.data(), sizeof(int32_t) * tasks.size(), hipMemcpyHostToDevice));
    built = true;
    return OMR_OK;
}

// the units of a launch of ncq (strip group, group of scan groups) combinations, dealt to the XCDs (made once per ncq)
int SlanePlan::units_for(int ncq, const int32_t **d_tab, int *per_xcd) const
{
    std::lock_guard<std::mutex> lk(unit_mu);
    UnitTab &u = unit_tabs[ncq];
    if (!u.tab.p) {
        const std::vector<int32_t> h = slane_deal_units(chunk_weight, chunk_size, ncq, &u.per_xcd);
        OMR_HIP(u.tab.alloc(sizeof(int32_t) * h.size()));
        OMR_HIP(hipMemcpy(u.tab.p, h.data(), sizeof(int32_t) * h.size(), hipMemcpyHostToDevice));
    }
    *d_tab = u.tab.as<int32_t>(), *per_xcd = u.per_xcd;
    return OMR_OK;
}

int SlaneScratch::create(const SlanePlan &p, int groups)
{
    if (groups < 1 || groups > 64) return fail(OMR_ERR_BADARG, "1..64 scan groups per launch");
    nsg = groups;
    const SlaneGeom &g = p.g;
    const size_t nscp = (size_t)nsg * SL_LANES, ntasks = p.tasks.size();
    // the scan groups' bit images, group_stride() (a power of two) apart from a base aligned to it: none straddles a 4 GB
    // boundary, so the kernel's 32-bit add to a descriptor's base never carries (slane.hpp)
    const size_t gstride = g.group_stride(), bits_b = ((size_t)nsg + 1) * gstride;
    OMR_HIP(bits.alloc(bits_b));
    OMR_HIP(hipMemset(bits.p, 0, bits_b));  // entry 0 and the guard columns stay zero for good
    bits_base = (uint32_t *)(((uintptr_t)bits.p + gstride - 1) & ~(uintptr_t)(gstride - 1));
    // row counts, two records per dword, and behind them the black-pixel totals [candidate][scan] (added up by the
    // column-count kernel): both are accumulated with atomics, so every launch leaves them zero again
    rows_bytes = sizeof(uint32_t) * ((size_t)p.A * (p.nrec / 2) * nscp + (size_t)p.A * nscp);
    OMR_HIP(hrows.alloc(rows_bytes));
    OMR_HIP(hipMemset(hrows.p, 0, rows_bytes));
    OMR_HIP(guard.alloc(sizeof(int32_t)));
    OMR_HIP(hipMemset(guard.p, 0, sizeof(int32_t)));
    OMR_HIP(vproj.alloc(sizeof(uint16_t) * (size_t)p.A * g.cols * nscp));  // column counts <= rows <= 65535
    OMR_HIP(vsd.alloc(sizeof(double) * nscp * p.A));
    OMR_HIP(hsd.alloc(sizeof(double) * nscp * p.A));
    OMR_HIP(best.alloc(sizeof(int32_t) * nscp));
    // ---- one descriptor per (candidate in launch order, place in its workgroups, scan group of the padded table),
    // for each composition of a workgroup: 4 strips x 4 scan groups (sgw_log 2), 8 x 2 (1), 16 x 1 (0).  (2 x 8 for
    // launches of 512 scans was built and measured: 41.2 ms against 40.8 -- eight-way sharing of a program through the
    // scalar cache buys nothing over four-way.)
    // Places beyond the last strip or the last scan group are null tasks: the empty program, buffer descriptors of
    // size 0 (their fetches read zeros, their row-count adds are dropped), a spare slot for the counter dump.
    OMR_HIP(planes.alloc(sizeof(uint32_t) * (ntasks * nsg + 1) * SL_K * SL_DUMP * SL_LANES));
    const uint64_t prog0 = (uint64_t)p.prog.p;
    for (int lg = 0; lg < 3; lg++) {
        const int sgw = 1 << lg, places = 16 >> lg;
        const int NSp = ((g.NS + places - 1) / places) * places, nsgp = ((nsg + sgw - 1) / sgw) * sgw;
        std::vector<SlaneTask> h((size_t)p.A * NSp * nsgp);
        for (int ai = 0; ai < p.A; ai++)
            for (int st = 0; st < NSp; st++) {
                const size_t ti = (size_t)ai * g.NS + (st < g.NS ? st : 0);  // index into p.tasks (launch order)
                const int task = p.tasks[ti], a = task / g.NS;
                const SlaneStrip &S = p.strips[(size_t)task];
                const int place = st % places;
                // the SL_BLOCK / 2 pair rows of a scan group's LDS buffer, dealt to the strips of the workgroup
                constexpr int PR = SL_BLOCK / 2;
                const int first = places <= PR ? (PR / places) * place : place % PR, count = places <= PR ? PR / places : place < PR ? 1 : 0;
                for (int sg = 0; sg < nsgp; sg++) {
                    const bool real = st < g.NS && sg < nsg;
                    const bool counts = sg < nsg;  // a null strip of a real scan group still flushes that group's row counts
                    SlaneTask &k = h[((size_t)ai * NSp + st) * nsgp + sg];
                    memset(&k, 0, sizeof k);
                    k.seg = prog0 + 4ull * (uint64_t)(real ? S.seg_offset : p.null_seg);
                    k.fet = prog0 + 4ull * (uint64_t)(real ? S.fet_offset : p.null_fet);
                    const uint64_t hb = (uint64_t)hrows.p + (counts ? 4ull * ((uint64_t)a * (p.nrec / 2) * nscp + (uint64_t)sg * SL_LANES) : 0ull);
                    k.hrsrc[0] = (uint32_t)hb;
                    k.hrsrc[1] = (uint32_t)(hb >> 32) & 0xffffu;
                    k.hrsrc[2] = counts ? (uint32_t)((size_t)(p.nrec / 2) * nscp * 4 - (size_t)sg * SL_LANES * 4) : 0u;
                    k.hrsrc[3] = 0x00020000u;
                    k.planes = (uint64_t)planes.p + 4ull * ((real ? ti * nsg + sg : ntasks * nsg) * SL_K * SL_DUMP * SL_LANES);
                    const uint64_t base = (uint64_t)bits_base + (counts ? (uint64_t)sg * gstride : 0ull);
                    k.rsrc[0] = (uint32_t)base;
                    k.rsrc[1] = (uint32_t)(base >> 32) & 0xffffu;
                    k.rsrc[2] = real ? (uint32_t)g.image_bytes() : 0u;
                    k.rsrc[3] = 0x00020000u;
                    k.nrec = (uint32_t)p.nrec;
                    k.nexec = (uint32_t)p.nexec;
                    k.hpitch = (uint32_t)(nscp * 4);
                    k.cls = real ? S.cls : 0;
                    static_assert(SL_BLOCK / 2 <= (1 << SL_WAVE_FIRST_BITS) && 8 < (1 << SL_WAVE_COUNT_BITS), "SlaneTask::wave fields");
                    k.wave = first | (count << SL_WAVE_COUNT_SHIFT);
                    k.lds_base = (uint32_t)((sg % sgw) * 2 * (SL_BLOCK / 2) * SL_LANES * 4);
                }
            }
        OMR_HIP(descs[lg].alloc(sizeof(SlaneTask) * h.size()));
        OMR_HIP(hipMemcpy(descs[lg].p, h.data(), sizeof(SlaneTask) * h.size(), hipMemcpyHostToDevice));
    }
    return OMR_OK;
}

// pack -> sweep -> column counts -> std-dev -> arg-max for `nscans` device-resident scans (at most nsg * 64)
// (black_max < 0: d_img holds scans ALREADY packed to 1 bit per pixel, [rows][NW] dwords each, scan_stride bytes apart)
int slane_enqueue(const SlanePlan &p, SlaneScratch &s, const uint8_t *d_img, int64_t scan_stride, int64_t step, int nscans,
                  int black_max, hipStream_t stream, hipStream_t post_stream, hipEvent_t ev_mid, double *d_v_sd, double *d_h_sd,
                  int32_t *d_best, hipEvent_t ev0, hipEvent_t ev1)
{
    const bool packed = black_max < 0;
    if (!d_img || nscans < 1 || nscans > s.nsg * SL_LANES) return fail(OMR_ERR_BADARG, "scan-lane launch: %d scans, scratch holds %d", nscans, s.nsg * SL_LANES);
    if (!packed && step < p.g.cols) return fail(OMR_ERR_BADARG, "step_bytes %lld < cols %d", (long long)step, p.g.cols);
    if (packed && ((scan_stride & 3) != 0 || scan_stride < (int64_t)p.g.rows * p.g.NW * 4 || ((uintptr_t)d_img & 3) != 0))
        return fail(OMR_ERR_BADARG, "packed scans: rows x %d dwords each, 4-byte aligned", p.g.NW);
    const int used = (nscans + SL_LANES - 1) / SL_LANES;  // scan groups that hold scans; the descriptors are laid out for s.nsg
    const size_t nscp = (size_t)s.nsg * SL_LANES;
    if (s.rows_dirty) {  // the previous launch was asked to keep its row counts (omr_batch_lanes_keep)
        OMR_HIP(hipMemsetAsync(s.hrows.p, 0, s.rows_bytes, stream));
        s.rows_dirty = false;
    }
    if (packed) OMR_HIP(launch_slane_pack_bits((const uint32_t *)d_img, scan_stride / 4, p.g, nscans, s.bits_base, stream));
    else OMR_HIP(launch_slane_pack(d_img, scan_stride, step, p.g, nscans, black_max, s.bits_base, stream));
    if (ev0) OMR_HIP(hipEventRecord(ev0, stream));
    {   // the workgroup's composition: 16 strips x 1 scan group, 8 x 2 or 4 x 4 (the scratch holds a table for each)
        const int lg = used <= 1 ? 0 : used == 2 ? 1 : 2, sgw = 1 << lg, places = 16 >> lg;
        const int nsgq = (used + sgw - 1) / sgw, NQ = (p.g.NS + places - 1) / places;
        const int32_t *d_tab = nullptr;
        int per_xcd = 0;
        if (int rc = p.units_for(NQ * nsgq, &d_tab, &per_xcd)) return rc;
        OMR_HIP(launch_slane(s.descs[lg].as<SlaneTask>(), nsgq, ((s.nsg + sgw - 1) / sgw) * sgw, p.A, NQ, lg, s.guard.as<int32_t>(),
                             d_tab, per_xcd, stream));
        s.guard_pending = true;
    }
    if (ev1) OMR_HIP(hipEventRecord(ev1, stream));
    if (post_stream && ev_mid) {
        OMR_HIP(hipEventRecord(ev_mid, stream));
        OMR_HIP(hipStreamWaitEvent(post_stream, ev_mid, 0));
        stream = post_stream;
    }
    OMR_HIP(launch_slane_vproj(s.planes.as<uint32_t>(), p.d_tasks.as<int32_t>(), (int)p.tasks.size(), used, s.nsg, p.g.NS, p.g.cols,
                               p.g.off, p.nrec, s.vproj.as<uint16_t>(), s.hrows.as<uint32_t>() + (size_t)p.A * (p.nrec / 2) * nscp, stream));
    double *vs = d_v_sd ? d_v_sd : s.vsd.as<double>(), *hs = d_h_sd ? d_h_sd : s.hsd.as<double>();
    OMR_HIP(launch_slane_stddev(s.vproj.as<uint16_t>(), s.hrows.as<uint32_t>(), s.hrows.as<uint32_t>() + (size_t)p.A * (p.nrec / 2) * nscp,
                                p.A, p.g.cols, p.g.rows, p.nrec / 2, p.hrow0, used, s.nsg,
                                nscans, vs, hs, stream));
    if (d_best) OMR_HIP(launch_argmax_path1(vs, hs, p.A, d_best, stream, nscans));
    // the row counts are accumulated with atomics: cleared here, behind their only reader and off the sweep's stream
    // (the caller orders the next launch on this scratch set behind this stream's work)
    if (s.keep_rows) s.rows_dirty = true;
    else OMR_HIP(hipMemsetAsync(s.hrows.p, 0, s.rows_bytes, stream));
    return OMR_OK;
}

}  // namespace omr
