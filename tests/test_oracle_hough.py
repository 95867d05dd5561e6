"""The Hough-line path of the CPU oracle (oracle/oracle_hough.c): hand-computed micro cases, the
independent numpy / scalar-Python restatement (oracle/oracle_np.py) and the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle_np as onp
from oics import synth


def hough_files(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "hough_*.npz")))


def test_sobel_and_canny_of_a_step_edge(oracle):
    img = np.zeros((9, 12), np.uint8)
    img[:, 6:] = 255
    dx, dy = oracle.sobel3_16s(img)
    assert (dy == 0).all()
    assert (dx[:, 5, 0] == 1020).all() and (dx[:, 6, 0] == 1020).all()      # (1 + 2 + 1) * 255 on both sides of the step
    assert (dx[:, :5, 0] == 0).all() and (dx[:, 7:, 0] == 0).all()          # replicate border: no edge at the frame
    e = oracle.canny(img)
    # two equal maxima side by side: "m > left && m >= right" keeps the LEFT one only
    exp = np.zeros_like(img)
    exp[:, 5] = 255
    assert (e == exp).all()


def test_canny_hysteresis_links_weak_to_strong(oracle):
    img = np.zeros((12, 40), np.uint8)
    img[:, 20:] = 30                      # weak step: |dx| = 120 -> between low 50 and high 150
    assert not oracle.canny(img).any()    # nothing strong: no edge survives
    img[0:3, 20:] = 60                    # a strong stretch (|dx| = 240) at the top of the same step
    e = oracle.canny(img)
    # rows 4.. are exactly the weak step that vanished above; now they hang (8-connected, through the
    # bend at rows 2-3) on the strong stretch and survive
    assert e[0:2, 19].all() and e[4:, 19].all()
    from scipy import ndimage
    lab, n = ndimage.label(e, structure=np.ones((3, 3), int))
    assert n == 1


def test_canny_picks_the_strongest_channel(oracle):
    g = np.zeros((10, 16), np.uint8)
    bgr = np.stack([g, g, g], axis=2)
    bgr[:, 8:, 1] = 200                    # only channel 1 has a step
    one = g.copy()
    one[:, 8:] = 200
    assert (oracle.canny(bgr) == oracle.canny(one)).all()


def test_rng_and_single_line_segment(oracle):
    # cv::RNG((uint64)-1): state' = (u32)state * 4164903690 + (state >> 32)
    s = (1 << 64) - 1
    s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & ((1 << 64) - 1)
    first = (s & 0xFFFFFFFF) % 100
    img = np.zeros((32, 160), np.uint8)
    img[10, 20:120] = 255                  # 100 points, raster order = left to right
    lines = oracle.hough_lines_p(img, 30, 3)
    # point `first` is drawn first: every bin ties at 1 vote -> angle 0 -> a vertical walk that finds
    # only the point itself -> rejected and erased.  The second point meets it at theta = 90 deg
    # (2 votes): the walk along the row bridges the 1-px hole and returns the whole line, left end first.
    left = 21 if first == 0 else 20
    right = 118 if first == 99 else 119
    assert lines.tolist() == [[left, 10, right, 10]]
    assert oracle.line_angles_f32(lines).tolist() == [0.0]


def test_angle_quirks(oracle):
    f = oracle.line_angles_f32
    assert f(np.array([[0, 0, 0, 100]], np.int32))[0] == 0.0                        # +90 % 45
    a = f(np.array([[0, 100, 0, 0]], np.int32))[0]
    assert a == 0.0 and np.signbit(a)                                             # -90 % 45 = -0.0 (sign of the dividend)
    assert abs(f(np.array([[0, 0, 10, 1146]], np.int32))[0] - 44.5) < 0.01       # quirk B9: +89.5 deg -> 44.5, not -0.5
    assert abs(f(np.array([[0, 0, 1000, -35]], np.int32))[0] + 2.0045) < 0.001


def test_votes(oracle):
    ang = np.array([0.0, 3.0, 3.05, 2.96, 10.0, 10.02], np.float32)
    assert oracle.vote_hough_rs(ang) == float(np.float32(3.0))                     # first of the 3-member cluster
    a, st, cand = oracle.vote_omr_rs(ang)
    assert a == float(np.float32(3.0)) and st == 1                                # 3.0, 3.05, 2.96 all count 3 -> NeedCheck
    assert cand.tolist() == [float(np.float32(3.0)), float(np.float32(3.05)), float(np.float32(2.96))]
    a, st, cand = oracle.vote_omr_rs(np.array([1.0, 5.0, 5.01], np.float32))
    assert st == 1 and len(cand) == 2
    a, st, cand = oracle.vote_omr_rs(np.array([1.0, 5.0, 5.01, 5.02, 9.0, 4.91], np.float32))
    assert a == 5.0 and st == 0 and cand.tolist() == [5.0]      # only 5.0 is within 0.1 of all of 5.01, 5.02 and 4.91
    with pytest.raises(RuntimeError):
        oracle.vote_hough_rs(np.zeros(0, np.float32))                             # angles[0] panics in the reference


@pytest.mark.parametrize("rows,cols,seed", [(48, 64, 3), (97, 120, 5), (150, 131, 8)])
def test_numpy_restatement_agrees(oracle, rows, cols, seed):
    g, _ = synth.make_card(rows, cols, seed)
    e = oracle.canny(g)
    assert (e == onp.canny_np(g)).all()
    rgb = np.stack([g, np.roll(g, 1, 0), np.roll(g, 2, 1)], axis=2)
    assert (oracle.canny(rgb) == onp.canny_np(rgb)).all()
    for mll, mlg in ((20, 5), (10, 2)):
        a = oracle.hough_lines_p(e, mll, mlg)
        b = onp.hough_lines_p_py(e, mll, mlg)
        assert a.shape == b.shape and (a == b).all()
        if len(a):  # numpy's float32 arctan2 need not be libm's: 1e-4 degrees, modulo the "% 45" wrap
            dlt = np.abs(oracle.line_angles_f32(a).astype(np.float64) - onp.line_angles_f32_np(a))
            assert np.minimum(dlt, 45.0 - dlt).max() < 1e-4


def test_golden_hough_present(golden_dir):
    assert len(hough_files(golden_dir)) >= 5


def test_oracle_reproduces_hough_golden(oracle, golden_dir):
    for f in hough_files(golden_dir):
        d = np.load(f)
        img = d["img"]
        edges = oracle.canny(img)
        assert (np.packbits(edges != 0, axis=1, bitorder="little") == d["edges_bits"]).all(), f
        lines = oracle.hough_lines_p(edges, float(d["min_line_length"]), float(d["max_line_gap"]))
        assert lines.shape == d["lines"].shape and (lines == d["lines"]).all(), f
        ang = oracle.line_angles_f32(lines)
        assert (ang.view(np.uint32) == d["angle_bits"]).all(), f
        if len(lines):
            assert np.float64(oracle.vote_hough_rs(ang)).view(np.uint64) == d["hough_rs_angle_bits"], f
            a2, st, cand = oracle.vote_omr_rs(ang)
            assert np.float64(a2).view(np.uint64) == d["omr_rs_angle_bits"] and st == int(d["omr_rs_status"]), f
            assert (cand.view(np.uint64) == d["omr_rs_candidate_bits"]).all(), f


def test_correct_default_decision(oracle):
    assert oracle.correct_default_decision(1.0, 0, [], 5.0) == (1.0, False)                 # Believed
    assert oracle.correct_default_decision(1.0, 1, [1.0, 1.2], 1.05) == (1.0, False)        # NeedCheck, agree
    assert oracle.correct_default_decision(1.0, 1, [1.0, 1.2], 1.3) == (1.3, True)          # NeedCheck, disagree
    assert oracle.correct_default_decision(0.0, 2, [0.4, -0.2, 0.41], 0.43) == (0.41, False)
    assert oracle.correct_default_decision(0.0, 2, [0.4, -0.2], 3.0) == (3.0, True)
    assert oracle.correct_default_decision(0.0, 2, [], 3.0) == (3.0, True)
    assert oracle.correct_default_decision(2.0, 2, [1.0, 3.0], 2.0) == (2.0, True)          # nearest is 1.0 away
