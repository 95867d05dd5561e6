"""The scan-lane sweep on the GPU (omr_batch_set_lanes): 64 scans per wavefront, the geometry as
wave-uniform programs.  Bar: bit-exact -- integer projections, f64 std-dev bit patterns and the arg-max index equal
the CPU oracle's (projection.rs:47-65, calculate.rs:13-23, projection.rs:125-190) for every scan of a batch, in every
lane position, for partial groups, and equal to the run-merging path at the headline size."""
import os

import numpy as np
import pytest
import torch

import oics
from oics import projection, synth

pytestmark = pytest.mark.gpu


def make_scans(rows, cols, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n):
        kind = i % 5
        if kind == 0:
            img = np.where(rng.random((rows, cols)) < rng.uniform(0.02, 0.6), 0, 255).astype(np.uint8)
        elif kind == 1:
            img = synth.make_binary_card(rows, cols, seed * 100 + i, skew=float(rng.uniform(-4, 4)))[0]
        elif kind == 2:
            img = np.full((rows, cols), 255, np.uint8)
            img[rng.integers(0, rows, 40), :] = 0  # horizontal rules
            img[:, rng.integers(0, cols, 25)] = 0  # vertical rules
        elif kind == 3:
            img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)  # non-binary: threshold fused in the pack
        else:
            img = np.zeros((rows, cols), np.uint8) if i % 2 else np.full((rows, cols), 255, np.uint8)
        out.append(np.ascontiguousarray(img))
    return out


def run_lanes(scans, max_angle, step, lanes, want_proj=()):
    rows, cols = scans[0].shape
    n = len(scans)
    dev = torch.device("cuda:0")
    buf = torch.empty((n, rows, cols), dtype=torch.uint8, device=dev)
    for i, s in enumerate(scans):
        buf[i] = torch.from_numpy(s).to(dev)
    _, A = projection.candidate_count(max_angle, step)
    best = torch.zeros(n, dtype=torch.int32, device=dev)
    vs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    b = projection.Batch(rows, cols, max_angle, step, n_streams=1)
    b.set_lanes(lanes)
    if want_proj:
        b.lanes_keep(True)
    b.run_device(buf.data_ptr(), rows * cols, cols, n, 127, best.data_ptr(), vs.data_ptr(), hs.data_ptr())
    b.sync()
    proj = {}
    if n <= lanes:  # a single launch: scratch set 0 still holds it
        for scan, a in want_proj:
            proj[(scan, a)] = b.lanes_projections(scan, a, rows, cols)
    b.close()
    return best.cpu().numpy(), vs.cpu().numpy(), hs.cpu().numpy(), proj


@pytest.mark.parametrize("rows,cols,max_angle,step,n", [(200, 300, 5, 0.5, 70), (97, 131, 9, 1.5, 64), (333, 64, 10, 1.0, 5),
                                                        (120, 1000, 3, 0.25, 130), (5003, 150, 4, 1.0, 66),
                                                        # widths = 31 / 30 modulo 32 at +-10 degrees: the destination grid is
                                                        # moved so that no partial word needs a ninth slot (slane_grid_offset)
                                                        (260, 607, 10, 2.5, 66), (180, 958, 10, 5.0, 9)])
def test_lanes_match_the_oracle(oracle, rows, cols, max_angle, step, n):
    scans = make_scans(rows, cols, n, rows + cols)
    N, A = oracle.candidate_count(max_angle, step)
    probes = [(0, 0), (n - 1, A - 1), (n // 2, A // 2), (min(n - 1, 63), 1)]
    best, vs, hs, proj = run_lanes(scans, max_angle, step, 64 * ((n + 63) // 64), probes)
    for i, img in enumerate(scans):
        binimg = np.where(img <= 127, 0, 255).astype(np.uint8)
        evp, ehp, evs, ehs = oracle.sweep(binimg, max_angle, step)
        for (scan, a), (vp, hp) in proj.items():
            if scan == i:
                assert (vp == evp[a]).all(), "vproj scan %d candidate %d" % (scan, a)
                assert (hp == ehp[a]).all(), "hproj scan %d candidate %d" % (scan, a)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all(), "v_sd bits, scan %d" % i
        assert (hs[i].view(np.uint64) == ehs.view(np.uint64)).all(), "h_sd bits, scan %d" % i
        assert best[i] == oracle.argmax_path1(evs, ehs)[0]


def test_lanes_row_pitch_scan_stride_and_threshold(oracle):
    """Scans that sit in a larger buffer: rows 13 bytes apart from packed (an unaligned pitch: the pack's byte path),
    scans a guard band apart, grey values with another threshold than 127."""
    rows, cols, n, pitch, gap = 140, 211, 70, 224, 777
    rng = np.random.Generator(np.random.PCG64(5))
    scans = [rng.integers(0, 256, (rows, cols)).astype(np.uint8) for _ in range(n)]
    stride = rows * pitch + gap
    dev = torch.device("cuda:0")
    host = np.full(n * stride, 7, np.uint8)
    for i, s in enumerate(scans):
        host[i * stride:i * stride + rows * pitch].reshape(rows, pitch)[:, :cols] = s
    buf = torch.from_numpy(host).to(dev)
    N, A = oracle.candidate_count(6, 1.0)
    best = torch.zeros(n, dtype=torch.int32, device=dev)
    vs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    b = projection.Batch(rows, cols, 6, 1.0, n_streams=1)
    b.set_lanes(128)
    b.run_device(buf.data_ptr(), stride, pitch, n, 90, best.data_ptr(), vs.data_ptr(), hs.data_ptr())
    b.sync()
    b.close()
    best, vs, hs = best.cpu().numpy(), vs.cpu().numpy(), hs.cpu().numpy()
    for i in (0, 1, 63, 64, 69):
        binimg = np.where(scans[i] <= 90, 0, 255).astype(np.uint8)
        _, _, evs, ehs = oracle.sweep(binimg, 6, 1.0)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all() and (hs[i].view(np.uint64) == ehs.view(np.uint64)).all()
        assert best[i] == oracle.argmax_path1(evs, ehs)[0]


def test_lanes_in_several_launches(oracle):
    # 200 scans through launches of 64: the scratch sets alternate, the last launch is a partial group
    rows, cols = 150, 220
    scans = make_scans(rows, cols, 200, 7)
    best, vs, hs, _ = run_lanes(scans, 6, 0.5, 64)
    for i in (0, 63, 64, 127, 128, 191, 192, 199):
        binimg = np.where(scans[i] <= 127, 0, 255).astype(np.uint8)
        _, _, evs, ehs = oracle.sweep(binimg, 6, 0.5)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all() and (hs[i].view(np.uint64) == ehs.view(np.uint64)).all()
        assert best[i] == oracle.argmax_path1(evs, ehs)[0]


def test_lanes_headline_size(oracle):
    """C2's shape and sweep, 66 scans (two scan groups, the second nearly empty): two scans against the oracle on all
    host cores, every scan against the run-merging path."""
    rows, cols = 3508, 2480
    cards = [synth.make_card(rows, cols, 3 + i) for i in range(6)]
    scans = [cards[i % 6][0] for i in range(66)]
    best, vs, hs, proj = run_lanes(scans, 10, 0.05, 128, [(0, 0), (65, 399), (5, 137)])
    dev = torch.device("cuda:0")
    ref = {}
    b = projection.Batch(rows, cols, 10, 0.05, n_streams=1)
    buf = torch.empty((6, rows, cols), dtype=torch.uint8, device=dev)
    for i in range(6):
        buf[i] = torch.from_numpy(cards[i][0]).to(dev)
    rb = torch.zeros(6, dtype=torch.int32, device=dev)
    rv = torch.zeros((6, 400), dtype=torch.float64, device=dev)
    rh = torch.zeros((6, 400), dtype=torch.float64, device=dev)
    b.run_device(buf.data_ptr(), rows * cols, cols, 6, 127, rb.data_ptr(), rv.data_ptr(), rh.data_ptr())
    b.sync()
    b.close()
    rb, rv, rh = rb.cpu().numpy(), rv.cpu().numpy(), rh.cpu().numpy()
    for i in range(66):
        assert (vs[i].view(np.uint64) == rv[i % 6].view(np.uint64)).all(), "v_sd vs run-merging, scan %d" % i
        assert (hs[i].view(np.uint64) == rh[i % 6].view(np.uint64)).all(), "h_sd vs run-merging, scan %d" % i
        assert best[i] == rb[i % 6]
    for i in (0, 5):
        binimg = np.where(scans[i] <= 127, 0, 255).astype(np.uint8)
        evp, ehp, evs, ehs = oracle.sweep(binimg, 10, 0.05, threads=os.cpu_count() or 4, fast=True)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all() and (hs[i].view(np.uint64) == ehs.view(np.uint64)).all()
        for (scan, a), (vp, hp) in proj.items():
            if scan == i:
                assert (vp == evp[a]).all() and (hp == ehp[a]).all()
        assert abs((best[i] - 200) * 0.05 - cards[i][1]) < 0.5


def _a4_card(seed):
    return synth.make_card(3508, 2480, seed)


def test_lanes_the_launch_bench_py_times(oracle):
    """THE launch form bench.py times (round-4 verdict, item 1): C2's shape and sweep, 500 scans in ONE scan-lane launch
    -- set_lanes(512): 4 x 4 workgroups, TWO quads of scan groups (slane.hip: sgq = cq / NQ, scan groups 4 .. 7), the last
    group partial -- with different content in every lane of every group (no scan equals another: a second-quad
    workgroup that read the first quad's images, or wrote to its slots, cannot pass).  One scan of EVERY scan group
    against the oracle on all 400 candidates (f64 bits of both std-devs, arg-max; projection.rs:47-65, :125-190), integer
    projections of probed (scan, candidate) pairs in both quads, and the black-pixel total of every scan as a cheap
    whole-batch property (every scan's row counts and column counts of a candidate add up to the same number, which
    is the scan's own: a misplaced lane shows up as another scan's total)."""
    import multiprocessing as mp
    rows, cols, n = 3508, 2480, 500
    with mp.get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:  # spawned: HIP is initialised in this process
        base = [c[0] for c in pool.map(_a4_card, range(70, 78))]
    dev = torch.device("cuda:0")
    buf = torch.empty((n, rows, cols), dtype=torch.uint8, device=dev)
    tb = [torch.from_numpy(b).to(dev) for b in base]
    shifts = [((37 * i) % 211 - 105, (53 * i) % 127 - 63) for i in range(n)]  # 500 distinct (dy, dx) pairs
    assert len(set(shifts)) == n
    for i in range(n):
        buf[i] = torch.roll(tb[i % 8], shifts=shifts[i], dims=(0, 1))
    del tb
    black = (buf <= 127).sum(dim=(1, 2)).cpu().numpy()
    A = 400
    best = torch.zeros(n, dtype=torch.int32, device=dev)
    vs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    b = projection.Batch(rows, cols, 10, 0.05, n_streams=1)
    b.set_lanes(512)
    b.lanes_keep(True)
    b.run_device(buf.data_ptr(), rows * cols, cols, n, 127, best.data_ptr(), vs.data_ptr(), hs.data_ptr())
    b.sync()
    probes = {(3, 0): None, (70, 399): None, (200, 17): None, (255, 200): None, (256, 1): None, (333, 399): None, (400, 123): None,
              (447, 0): None, (448, 250): None, (499, 399): None}
    for scan, a in probes:
        probes[(scan, a)] = b.lanes_projections(scan, a, rows, cols)
    b.close()
    best, vs, hs = best.cpu().numpy(), vs.cpu().numpy(), hs.cpu().numpy()
    for (scan, a), (vp, hp) in probes.items():  # every scan's own black pixels, whatever the candidate lets fall outside
        assert int(vp.sum()) == int(hp.sum()) and int(vp.sum()) <= int(black[scan])
    checked = (0, 100, 191, 192, 256, 300, 319, 383, 384, 447, 448, 499)  # groups 0, 1, 2, 3 | 4, 4, 4, 5, 6, 6, 7, 7
    assert {i // 64 for i in checked} == set(range(8))
    for i in checked:
        binimg = np.where(np.roll(base[i % 8], shifts[i], axis=(0, 1)) <= 127, 0, 255).astype(np.uint8)
        assert int((binimg == 0).sum()) == int(black[i])
        evp, ehp, evs, ehs = oracle.sweep(binimg, 10, 0.05, threads=os.cpu_count() or 4, fast=True)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all(), "v_sd bits, scan %d (group %d)" % (i, i // 64)
        assert (hs[i].view(np.uint64) == ehs.view(np.uint64)).all(), "h_sd bits, scan %d (group %d)" % (i, i // 64)
        assert best[i] == oracle.argmax_path1(evs, ehs)[0]
        for (scan, a), (vp, hp) in probes.items():
            if scan == i:
                assert (vp == evp[a]).all() and (hp == ehp[a]).all(), "projections of scan %d, candidate %d" % (scan, a)
    # the probes of scans that were not swept by the oracle above: against the oracle's warp of that one candidate
    Ms = oracle.rotation_matrices(rows, cols, 10, 0.05)
    for (scan, a), (vp, hp) in probes.items():
        if scan in checked:
            continue
        binimg = np.where(np.roll(base[scan % 8], shifts[scan], axis=(0, 1)) <= 127, 0, 255).astype(np.uint8)
        evp, ehp, _, _ = oracle.sweep_matrices(binimg, Ms[a:a + 1], threads=1, fast=True)
        assert (vp == evp[0]).all() and (hp == ehp[0]).all(), "projections of scan %d, candidate %d" % (scan, a)
    assert len({vs[i].tobytes() for i in range(n)}) == n, "500 different scans must give 500 different score vectors"


def _a4_600_card(seed):
    return synth.make_card(7016, 4960, seed)


def test_lanes_a4_at_600_dpi(oracle):
    """A batch of 600-dpi A4 scans (4960 x 7016, 34.8 MP each; 13 counter planes per word: more than 4095 rows) through
    the scan-lane sweep at the full +-10 deg @ 0.05 deg sweep -- 18 GB of programs generated on the device --: 66 scans in
    launches of 64 (a partial second launch), two of them against the oracle on all 400 candidates, every repeat of a card
    bit-identical to its first copy, the reference's < 0.5 deg criterion (lib.rs:103-113) on all."""
    import multiprocessing as mp
    rows, cols, n = 7016, 4960, 66
    with mp.get_context("spawn").Pool(min(3, os.cpu_count() or 1)) as pool:
        base = pool.map(_a4_600_card, range(90, 93))
    dev = torch.device("cuda:0")
    buf = torch.empty((n, rows, cols), dtype=torch.uint8, device=dev)
    tb = [torch.from_numpy(b[0]).to(dev) for b in base]
    for i in range(n):
        buf[i] = tb[i % 3]
    del tb
    A = 400
    best = torch.zeros(n, dtype=torch.int32, device=dev)
    vs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    b = projection.Batch(rows, cols, 10, 0.05, n_streams=1)
    b.set_lanes(64)
    assert b.lanes_program_bytes() > 12e9  # (the point of the case: programs and offsets beyond 4 GB)
    b.run_device(buf.data_ptr(), rows * cols, cols, n, 127, best.data_ptr(), vs.data_ptr(), hs.data_ptr())
    b.sync()
    b.close()
    del buf
    torch.cuda.empty_cache()
    best, vs, hs = best.cpu().numpy(), vs.cpu().numpy(), hs.cpu().numpy()
    for i in range(n):
        assert (vs[i].view(np.uint64) == vs[i % 3].view(np.uint64)).all() and (hs[i].view(np.uint64) == hs[i % 3].view(np.uint64)).all(), i
        assert best[i] == best[i % 3] and abs((best[i] - 200) * 0.05 - base[i % 3][1]) < 0.5
    for i in (1, 65):  # launch 0 / the partial launch 1
        binimg = np.where(base[i % 3][0] <= 127, 0, 255).astype(np.uint8)
        _, _, evs, ehs = oracle.sweep(binimg, 10, 0.05, threads=os.cpu_count() or 4, want_proj=False, fast=True)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all() and (hs[i].view(np.uint64) == ehs.view(np.uint64)).all(), i
        assert best[i] == oracle.argmax_path1(evs, ehs)[0]


@pytest.mark.parametrize("n", [320, 467, 512])
def test_lanes_five_to_eight_scan_groups_every_scan(oracle, n):
    """A small shape through the same launch form -- set_lanes(512), 4 x 4 workgroups, two quads of scan groups, 5 / 8
    (partial) / 8 groups in use -- with EVERY scan of the batch against the oracle."""
    rows, cols = 150, 220
    scans = make_scans(rows, cols, n, 1000 + n)
    best, vs, hs, proj = run_lanes(scans, 6, 0.5, 512, [(0, 0), (n - 1, 23), (256, 5), (300, 11), (319, 0)])
    for i, img in enumerate(scans):
        binimg = np.where(img <= 127, 0, 255).astype(np.uint8)
        evp, ehp, evs, ehs = oracle.sweep(binimg, 6, 0.5)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all(), "v_sd bits, scan %d" % i
        assert (hs[i].view(np.uint64) == ehs.view(np.uint64)).all(), "h_sd bits, scan %d" % i
        assert best[i] == oracle.argmax_path1(evs, ehs)[0]
        for (scan, a), (vp, hp) in proj.items():
            if scan == i:
                assert (vp == evp[a]).all() and (hp == ehp[a]).all()


@pytest.mark.parametrize("rows,cols,max_angle,step", [(200, 300, 5, 0.5), (333, 64, 10, 1.0), (1754, 1240, 10, 0.25), (97, 131, 9, 1.5),
                                                      (640, 1000, 7, 0.5), (3508, 2480, 10, 0.05), (7016, 300, 6, 1.0),
                                                      (700, 1343, 10, 1.0), (400, 94, 10, 2.0)])
def test_device_built_programs_equal_the_host_generator(rows, cols, max_angle, step):
    """The plan's programs are generated on the device (slane_build.hip); the host generator (slane_plan.cpp, the one
    tests/test_slane_program.py runs through the CPU interpreter against the oracle) must produce the same buffer,
    dword for dword: classes, layout, segment words, fetch schedule, commit lists, the null program."""
    b = projection.Batch(rows, cols, max_angle, step, n_streams=1)
    b.set_lanes(64)
    n, diff = b.lanes_check_programs()
    b.close()
    assert n > 0 and diff == 0, (n, diff)


def test_steep_sweep_is_refused(oracle):
    # +-20 degrees: more than 8 segments per word -> -213, the context stays on the run-merging / gather path
    b = projection.Batch(300, 400, 20, 1.0, n_streams=1)
    with pytest.raises(oics.OmrError) as e:
        b.set_lanes(64)
    assert e.value.code == -213
    b.close()


def test_host_batch_pageable_pinned_and_packed(oracle):
    """omr_host_batch: 150 binarised scans from host memory (>= 64 per device: the scan-lane sweep) through the three
    transfer modes -- pageable (copier threads -> pinned ring), page-locked source, PACKED (the copier threads turn every
    scan into 1 bit per pixel, 1/8 of the bytes are uploaded, a device kernel interleaves the packed rows) -- on one
    context, then packed again with launches of 128; every mode gives the same bits and they are the oracle's."""
    rows, cols, n = 180, 261, 150  # (261 columns: the last word of a packed row is partial, and rows are not whole dwords)
    scans = [np.where(s <= 127, 0, 255).astype(np.uint8) for s in make_scans(rows, cols, n, 31)]
    hb = projection.HostBatch(rows, cols, 6, 0.5, n, n_devices=1)
    nd, spl, lane = hb.info()
    assert nd == 1 and lane and spl % 64 == 0
    best, ang, vs, hs = hb.run(scans, want_sd=True)
    pinned_t = torch.empty((n, rows, cols), dtype=torch.uint8).pin_memory()
    pn = pinned_t.numpy()
    for i in range(n):
        pn[i] = scans[i]
    best2, _, vs2, hs2 = hb.run([pn[i] for i in range(n)], pinned=True, want_sd=True)
    assert (best == best2).all() and (vs.view(np.uint64) == vs2.view(np.uint64)).all() and (hs.view(np.uint64) == hs2.view(np.uint64)).all()
    # packed: scans with a row pitch (views into a wider buffer) as well
    wide = np.full((n, rows, cols + 19), 7, np.uint8)
    wide[:, :, :cols] = np.stack(scans)
    for src in (scans, [wide[i, :, :cols] for i in range(n)]):
        best3, ang3, vs3, hs3 = hb.run(src, want_sd=True, packed=True)
        assert (best == best3).all() and (ang == ang3).all()
        assert (vs.view(np.uint64) == vs3.view(np.uint64)).all() and (hs.view(np.uint64) == hs3.view(np.uint64)).all()
    hb.set_launch(128)
    assert hb.info()[1] == 128
    best4, _, vs4, hs4 = hb.run(scans, want_sd=True, packed=True)
    best5, _, vs5, _ = hb.run(scans, want_sd=True)  # the u8 path on the re-sized context
    hb.close()
    assert (best == best4).all() and (vs.view(np.uint64) == vs4.view(np.uint64)).all() and (hs.view(np.uint64) == hs4.view(np.uint64)).all()
    assert (best == best5).all() and (vs.view(np.uint64) == vs5.view(np.uint64)).all()
    N, A = oracle.candidate_count(6, 0.5)
    for i in range(0, n, 7):
        _, _, evs, ehs = oracle.sweep(scans[i], 6, 0.5)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all() and (hs[i].view(np.uint64) == ehs.view(np.uint64)).all()
        assert best[i] == oracle.argmax_path1(evs, ehs)[0] and ang[i] == (best[i] - N) * 0.5
    # a context on the run-merging path (fewer than 64 scans per device) takes the packed mode as a plain transfer
    hs_ = projection.HostBatch(rows, cols, 6, 0.5, 20, n_devices=1)
    assert not hs_.info()[2]
    b6, _, v6, _ = hs_.run(scans[:20], want_sd=True, packed=True)
    hs_.close()
    assert (b6 == best[:20]).all() and (v6.view(np.uint64) == vs[:20].view(np.uint64)).all()


def test_sweep_batch_of_mixed_shapes(oracle):
    """omr_sweep_batch buckets a batch by shape (the reference corrects one file per call, any size, task.rs:19-38; its own
    dataset holds 1240x1150 and ~1237x1300 sheets): three shapes interleaved, one of them with >= 64 scans (scan-lane
    sweep), the others on the run-merging path; every result at its scan's own position, bit for bit the oracle's."""
    shapes = [(115, 124), (130, 124), (97, 131)]
    counts = [70, 9, 21]
    scans, kinds = [], []
    pools = [[np.where(s <= 127, 0, 255).astype(np.uint8) for s in make_scans(r, c, k, 77 + r)] for (r, c), k in zip(shapes, counts)]
    at = [0, 0, 0]
    order = np.random.Generator(np.random.PCG64(4)).permutation(np.repeat(np.arange(3), counts))
    for k in order:
        scans.append(pools[k][at[k]])
        at[k] += 1
    best, ang, vs, hs = projection.sweep_batch(scans, 6, 0.5, n_devices=1, want_sd=True)
    N, A = oracle.candidate_count(6, 0.5)
    for i, img in enumerate(scans):
        _, _, evs, ehs = oracle.sweep(img, 6, 0.5)
        assert (vs[i].view(np.uint64) == evs.view(np.uint64)).all() and (hs[i].view(np.uint64) == ehs.view(np.uint64)).all(), i
        assert best[i] == oracle.argmax_path1(evs, ehs)[0] and ang[i] == (best[i] - N) * 0.5


def test_host_batch_refuses_more_devices_than_visible():
    with pytest.raises(oics.OmrError) as e:
        projection.HostBatch(100, 100, 5, 1.0, 8, n_devices=64)
    assert e.value.code == -5
