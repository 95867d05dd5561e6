"""Parity of the Hough-line deskew path through the C ABI against the CPU
oracle: Canny edges byte for byte, HoughLinesP segments integer for integer and in the same order,
angles / status / candidates bit for bit; correct_default's decision and its rotated image."""
import os
import sys
import numpy as np
import pytest

import oics
from oics import hough, omr, synth
from oics.types import ResultStatus

pytestmark = pytest.mark.gpu


def card(rows, cols, seed, channels=1):
    g, th = synth.make_card(rows, cols, seed)
    if channels == 1:
        return g, th
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    bgr = np.stack([g] * channels, axis=2).astype(np.int16)
    bgr[:, :, :3] += rng.integers(-12, 13, size=(rows, cols, 3), dtype=np.int16)  # channels differ: exercises the max-channel rule
    return np.clip(bgr, 0, 255).astype(np.uint8), th


@pytest.mark.parametrize("rows,cols,seed,cn", [(64, 48, 3, 1), (230, 248, 4, 1), (120, 97, 5, 3), (512, 512, 1, 1),
                                              (333, 517, 6, 3), (200, 300, 7, 4), (1754, 1240, 8, 1)])
def test_canny_matches_oracle(oracle, rows, cols, seed, cn):
    img, _ = card(rows, cols, seed, cn)
    got = hough.canny(img, 50.0, 150.0)
    exp = oracle.canny(img, 50.0, 150.0)
    assert got.shape == exp.shape and (got == exp).all(), "Canny edges differ: %d px" % int((got != exp).sum())
    assert set(np.unique(got).tolist()) <= {0, 255}


def test_canny_threshold_order_and_flat_image(oracle):
    img, _ = card(96, 128, 9)
    assert (hough.canny(img, 150.0, 50.0) == oracle.canny(img, 50.0, 150.0)).all()  # swapped thresholds
    flat = np.full((40, 50), 200, np.uint8)
    assert not hough.canny(flat).any()
    stripes = np.zeros((64, 64), np.uint8)
    stripes[:, 20:40] = 255
    assert (hough.canny(stripes) == oracle.canny(stripes)).all()


@pytest.mark.parametrize("rows,cols,seed,mll,mlg", [(64, 48, 3, 10, 2), (230, 248, 4, 20, 5), (230, 248, 4, 150, 50),
                                                   (512, 512, 1, 150, 50), (512, 512, 2, 100, 15),
                                                   (700, 300, 5, 60, 200), (1754, 1240, 8, 150, 50),
                                                   (512, 512, 1, 100, 63), (512, 512, 1, 100, 64), (300, 300, 7, 30, 0),
                                                   (600, 900, 9, 120, 127), (600, 900, 9, 120, 128)])
def test_hough_lines_p_matches_oracle(oracle, rows, cols, seed, mll, mlg):
    img, _ = card(rows, cols, seed)
    edges = oracle.canny(img)
    exp = oracle.hough_lines_p(edges, mll, mlg)
    got = hough.hough_lines_p(edges, 1.0, np.pi / 180.0, 0, mll, mlg)
    assert got.shape == exp.shape, "segment count %d vs %d" % (len(got), len(exp))
    assert (got == exp).all()


def test_hough_lines_p_threshold_and_sparse(oracle):
    img = np.zeros((200, 260), np.uint8)
    img[50, 20:240] = 255          # one horizontal line
    img[20:180, 130] = 255         # one vertical line
    for k in range(150):           # one diagonal
        img[20 + k, 30 + k] = 255
    for thr in (0, 10, 60):
        exp = oracle.hough_lines_p(img, 30, 3, threshold=thr)
        got = hough.hough_lines_p(img, 1.0, np.pi / 180.0, thr, 30, 3)
        assert got.shape == exp.shape and (got == exp).all(), thr
    empty = np.zeros((64, 64), np.uint8)
    assert len(hough.hough_lines_p(empty, 1.0, np.pi / 180.0, 0, 10, 2)) == 0
    with pytest.raises(oics.OmrError) as e:   # 0.25 degree steps = 720 accumulator angles
        hough.hough_lines_p(img, 1.0, np.pi / 720.0, 0, 30, 3)
    assert e.value.code == -213


@pytest.mark.parametrize("rows,cols,seed,cn", [(230, 248, 4, 1), (512, 512, 1, 1), (400, 300, 11, 3)])
def test_drivers_match_oracle(oracle, rows, cols, seed, cn):
    img, _ = card(rows, cols, seed, cn)
    for mll, mlg in ((150.0, 50.0), (40.0, 8.0)):
        try:
            e_ang, e_n = oracle.get_angle_with_hough(img, mll, mlg)
        except RuntimeError:
            with pytest.raises(oics.OmrError) as e:
                hough.get_angle_with_hough(img, mll, mlg)
            assert e.value.code == -215
            continue
        got = hough.get_angle_with_hough(img, mll, mlg)
        assert np.float64(got).view(np.uint64) == np.float64(e_ang).view(np.uint64)
        ea, es, ec, _ = oracle.get_result_from_edges_detection(img, mll, mlg)
        r = omr.get_result_from_edges_detection(img, mll, mlg)
        assert np.float64(r.angle).view(np.uint64) == np.float64(ea).view(np.uint64)
        assert int(r.status) == es
        assert r.candidates.size == ec.size and (r.candidates.view(np.uint64) == ec.view(np.uint64)).all()


def test_edges_detection_batch_device(oracle):
    import torch
    rows, cols, n = 300, 420, 6
    imgs = [card(rows, cols, 20 + i)[0] for i in range(n)]
    d = torch.from_numpy(np.stack(imgs)).to("cuda:0")
    ang, st, nl = omr.edges_detection_batch_device(d.data_ptr(), n, rows * cols, rows, cols, 1, cols, 60.0, 10.0)
    for i in range(n):
        ea, es, ec, en = oracle.get_result_from_edges_detection(imgs[i], 60.0, 10.0)
        assert nl[i] == en
        assert np.float64(ang[i]).view(np.uint64) == np.float64(ea).view(np.uint64) and st[i] == es


def test_correct_default(oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    for seed, shape in ((31, (1150, 1240)), (32, (1150, 1240)), (33, (690, 744))):
        g, th = synth.make_card(shape[0], shape[1], seed)
        bgr = np.stack([g, g, g], axis=2)
        ang, chk, rot = omr.correct_default(bgr, 45, 0.2, 248, 230, 150.0, 50.0)
        # oracle: the same composition (omr.rs:351-399)
        pa, pst, pc = oracle.get_result_from_projection(bgr, 45, 0.2, 248, 230)
        if pst == 0:
            e_ang, e_chk = pa, False
        else:
            ea, es, ec, _ = oracle.get_result_from_edges_detection(bgr, 150.0, 50.0)
            e_ang, e_chk = oracle.correct_default_decision(pa, pst, pc, ea)
        assert np.float64(ang).view(np.uint64) == np.float64(e_ang).view(np.uint64) and chk == e_chk
        exp = oracle.rotate_mat(bgr, ang, 1.0, 0, (255, 255, 255, 0), 1)
        assert rot.shape == exp.shape and (rot == exp).all()


def test_correct_default_decision_rule(oracle):
    cases = [(1.0, 0, [], 5.0), (1.0, 1, [1.0, 1.2], 1.05), (1.0, 1, [1.0, 1.2], 1.3), (0.0, 2, [0.4, -0.2, 0.41], 0.43),
             (0.0, 2, [0.4, -0.2], 3.0), (0.0, 2, [], 3.0), (2.0, 2, [1.0, 3.0], 2.0)]
    for pa, st, cand, ea in cases:
        r = omr.OmrResult(pa, ResultStatus(st), np.array(cand, np.float64))
        assert omr.correct_default_decision(r, ea) == oracle.correct_default_decision(pa, st, cand, ea)


def test_golden_vectors(golden_dir):
    import glob
    import os
    files = sorted(glob.glob(os.path.join(golden_dir, "hough_*.npz")))
    assert len(files) >= 5
    for f in files:
        d = np.load(f)
        img = d["img"]
        edges = hough.canny(img)
        assert (np.packbits(edges != 0, axis=1, bitorder="little") == d["edges_bits"]).all(), f
        mll, mlg = float(d["min_line_length"]), float(d["max_line_gap"])
        lines = hough.hough_lines_p(edges, 1.0, np.pi / 180.0, 0, mll, mlg)
        assert lines.shape == d["lines"].shape and (lines == d["lines"]).all(), f
        if len(lines):
            a1 = hough.get_angle_with_hough(img, mll, mlg)
            assert np.float64(a1).view(np.uint64) == d["hough_rs_angle_bits"], f
            r = omr.get_result_from_edges_detection(img, mll, mlg)
            assert np.float64(r.angle).view(np.uint64) == d["omr_rs_angle_bits"] and int(r.status) == int(d["omr_rs_status"])
            assert (r.candidates.view(np.uint64) == d["omr_rs_candidate_bits"]).all(), f


def test_entry_points_are_reentrant(oracle):
    """The Tauri host runs correct_default on a pool of OS threads (thread_pool.rs:41-54): concurrent
    calls into the Hough and FFT entry points must not disturb each other."""
    import threading
    from oics import fft
    cards = [card(200 + 10 * k, 260 + 7 * k, 60 + k)[0] for k in range(4)]
    exp = [(oracle.get_result_from_edges_detection(c, 40.0, 8.0)[:2], oracle.canny(c)) for c in cards]
    pics = [fft.get_fft_image(c)[1] for c in cards]
    errs = []

    def work(k):
        try:
            for _ in range(3):
                r = omr.get_result_from_edges_detection(cards[k], 40.0, 8.0)
                assert (np.float64(r.angle).view(np.uint64) == np.float64(exp[k][0][0]).view(np.uint64)
                        and int(r.status) == exp[k][0][1])
                assert (hough.canny(cards[k]) == exp[k][1]).all()
                assert (fft.get_fft_image(cards[k])[1] == pics[k]).all()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_large_line_sets_use_the_device_vote(oracle):
    """More than 2048 segments: the O(n^2) +-0.1 degree vote runs on the GPU (angle_votes_kernel)."""
    img, _ = card(1754, 1240, 12)
    ea, es, ec, en = oracle.get_result_from_edges_detection(img, 60.0, 20.0)
    assert en > 2048
    r = omr.get_result_from_edges_detection(img, 60.0, 20.0)
    assert np.float64(r.angle).view(np.uint64) == np.float64(ea).view(np.uint64) and int(r.status) == es
    assert r.candidates.size == ec.size and (r.candidates.view(np.uint64) == ec.view(np.uint64)).all()
    e_ang, e_n = oracle.get_angle_with_hough(img, 60.0, 20.0)
    assert np.float64(hough.get_angle_with_hough(img, 60.0, 20.0)).view(np.uint64) == np.float64(e_ang).view(np.uint64)


def test_headline_size_matches_oracle(oracle):
    """BASELINE config 4's scan size (2480x3508): edges, segments and result against the oracle."""
    img, _ = card(3508, 2480, 2)
    edges = hough.canny(img)
    assert (edges == oracle.canny(img, fast=True)).all()
    exp = oracle.hough_lines_p(edges, 150.0, 50.0, fast=True)
    got = hough.hough_lines_p(edges, 1.0, np.pi / 180.0, 0, 150.0, 50.0)
    assert got.shape == exp.shape and (got == exp).all()
    ea, es, ec, en = oracle.get_result_from_edges_detection(img, 150.0, 50.0, fast=True)
    r = omr.get_result_from_edges_detection(img, 150.0, 50.0)
    assert en == len(exp)
    assert np.float64(r.angle).view(np.uint64) == np.float64(ea).view(np.uint64) and int(r.status) == es


@pytest.mark.parametrize("theta_div,rho", [(90, 1.0), (250, 1.0), (180, 2.0), (180, 0.5), (45, 3.0)])
def test_hough_lines_p_other_resolutions(oracle, theta_div, rho):
    """rho / theta other than the reference's (1, pi/180): accumulator shape, trig table and walks follow."""
    img, _ = card(300, 420, 9)
    edges = oracle.canny(img)
    for thr in (0, 20):
        exp = oracle.hough_lines_p(edges, 30, 5, rho=rho, theta=np.pi / theta_div, threshold=thr)
        got = hough.hough_lines_p(edges, rho, np.pi / theta_div, thr, 30, 5)
        assert got.shape == exp.shape and (got == exp).all()


def test_hough_lines_p_random_cases(oracle, monkeypatch):
    """Random shapes, densities and parameters (tests/fuzz/fuzz_hough.py: point counts around the 64-point draw rounds,
    dense rows / columns, gaps on both sides of the 64-step rule, other rho / theta): segments equal the oracle's."""
    import runpy
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_hough.py")
    monkeypatch.setattr(sys, "argv", [tool, "80", "11"])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(tool, run_name="__main__")
    assert e.value.code == 0
