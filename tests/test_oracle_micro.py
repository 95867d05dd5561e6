"""Hand-computed micro cases that pin the oracle's fixed-point semantics.

Every expected value below is derived by hand from OpenCV 4.6.0's published warpAffine algorithm
(SURVEY.md Appendix A.2), not from running any code: pure translations and quarter turns make the
table arithmetic exact, so the floor-on-negative shift, the +512 rounding offset, round-half-even
and the white border can be checked one at a time.
"""
import numpy as np


def img_4x5():
    # rows x cols = 4 x 5, 0 = black, 255 = white
    return np.array([[0, 255, 255, 0, 255],
                     [255, 0, 255, 255, 255],
                     [255, 255, 0, 0, 255],
                     [0, 255, 255, 255, 0]], np.uint8)


def test_identity_is_exact(oracle):
    a = img_4x5()
    M = [1, 0, 0, 0, 1, 0]
    # adelta = 1024 x, X0 = 512  ->  X = (1024 x + 512) >> 10 = x
    ad, bd, X0, Y0 = oracle.warp_tables(oracle.invert_affine(M), 5, 4)
    assert ad.tolist() == [0, 1024, 2048, 3072, 4096] and bd.tolist() == [0] * 5
    assert X0.tolist() == [512] * 4 and Y0.tolist() == [512, 1536, 2560, 3584]
    assert (oracle.warp_affine(a, M) == a).all()


def test_translation_floor_on_negative_and_border(oracle):
    a = img_4x5()
    # forward shift right by 0.75 px: x_src = x - 0.75; X0 = rint(-768) + 512 = -256;
    # X = (1024 x - 256) >> 10 = x - 1 (arithmetic shift floors: -256 >> 10 = -1 -> border 255)
    out = oracle.warp_affine(a, [1, 0, 0.75, 0, 1, 0])
    exp = np.full_like(a, 255)
    exp[:, 1:] = a[:, :-1]
    assert (out == exp).all()
    # shift by exactly 0.5: X0 = -512 + 512 = 0 -> X = x (nearest picks the pixel itself)
    assert (oracle.warp_affine(a, [1, 0, 0.5, 0, 1, 0]) == a).all()
    # shift down by 1.25 rows: Y0[y] = rint((y - 1.25) * 1024) + 512 -> Y = floor(y - 0.75) = y - 1
    out = oracle.warp_affine(a, [1, 0, 0, 0, 1, 1.25])
    exp = np.full_like(a, 255)
    exp[1:, :] = a[:-1, :]
    assert (out == exp).all()


def test_round_half_to_even(oracle):
    a = img_4x5()
    # tx = 512.5 / 1024 exactly: (M1*y + M2) * 1024 = -512.5 -> cvRound = -512 (ties to even),
    # X0 = 0 -> X = x.  Round-half-away would give -513 -> X0 = -1 -> X = x - 1.
    assert (oracle.warp_affine(a, [1, 0, 512.5 / 1024, 0, 1, 0]) == a).all()
    # tx = 513.5 / 1024: -513.5 -> -514 (even) -> X0 = -2 -> X = (1024 x - 2) >> 10 = x - 1
    out = oracle.warp_affine(a, [1, 0, 513.5 / 1024, 0, 1, 0])
    exp = np.full_like(a, 255)
    exp[:, 1:] = a[:, :-1]
    assert (out == exp).all()


def test_quarter_turn_about_half_size_centre(oracle):
    # getRotationMatrix2D((2, 2), 90, 1): alpha ~ 6e-17, beta = 1 -> forward X' = y, Y' = 4 - x
    # inverse: x = 4 - Y', y = X'  =>  dst(x', y') = src[y = x'][x = 4 - y'];  y' = 0 -> x = 4: border
    a = np.arange(16, dtype=np.uint8).reshape(4, 4) * 10
    M = oracle.get_rotation_matrix_2d(2.0, 2.0, 90.0, 1.0)
    out = oracle.warp_affine(a, M)
    exp = np.full((4, 4), 255, np.uint8)
    for yp in range(1, 4):
        for xp in range(4):
            exp[yp, xp] = a[xp, 4 - yp]
    assert (out == exp).all()


def test_rotation_matrix_layout(oracle):
    # A.1: [alpha, beta, (1-alpha) cx - beta cy; -beta, alpha, beta cx + (1-alpha) cy], centre as f32
    M = oracle.get_rotation_matrix_2d(1240.0, 1754.0, 0.0, 1.0)
    assert M.tolist() == [1.0, 0.0, 0.0, -0.0, 1.0, 0.0]
    M = oracle.get_rotation_matrix_2d(10.0, 20.0, 180.0, 1.0)
    # alpha = -1, beta ~ 1.2e-16: (1 - -1) * 10 - beta * 20 = 20, beta * 10 + 2 * 20 = 40
    assert abs(M[0] + 1) < 1e-15 and abs(M[2] - 20) < 1e-12 and abs(M[5] - 40) < 1e-12
    M = oracle.get_rotation_matrix_2d(0.0, 0.0, 30.0, 2.0)
    assert abs(M[0] - 2 * np.cos(np.pi / 6)) < 1e-15 and abs(M[1] - 1.0) < 1e-15 and M[3] == -M[1]


def test_projections_count_zero_pixels_only(oracle):
    a = img_4x5()
    a[1, 2] = 7  # neither 0 nor 255: not counted (transfer.rs:322 `*item == 0`)
    assert oracle.vertical_projection(a).tolist() == [2, 1, 1, 2, 1]
    assert oracle.horizontal_projection(a).tolist() == [2, 1, 2, 2]
    h, v = oracle.mat_projection_data(a)  # omr.rs:8-39 returns (horizontal, vertical)
    assert h.tolist() == [2, 1, 2, 2] and v.tolist() == [2, 1, 1, 2, 1]


def test_population_std_dev(oracle):
    # calculate.rs:13-23: population (divide by n), not sample
    assert oracle.standard_deviation([2, 4, 4, 4, 5, 5, 7, 9]) == 2.0
    assert oracle.arithmetic_mean([1, 2, 3, 4]) == 2.5
    assert oracle.standard_deviation([3, 3, 3]) == 0.0
    v_sd, h_sd = oracle.projection_standard_deviations(img_4x5())  # order: (vertical, horizontal)
    assert v_sd == oracle.standard_deviation([2, 1, 1, 2, 1])
    assert h_sd == oracle.standard_deviation([2, 1, 2, 2])


def test_half_open_candidate_range(oracle):
    # projection.rs:36-38: N = (max/step) as u16, candidates -N..N (no +max endpoint)
    assert oracle.candidate_count(10, 0.05) == (200, 400)
    assert oracle.candidate_count(5, 0.5) == (10, 20)
    assert oracle.candidate_count(45, 0.2) == (225, 450)
    assert oracle.candidate_count(1, 0.3) == (3, 6)  # 3.33 truncates
    assert oracle.candidate_count(1, 2.0) == (0, 0)


def test_threshold_and_border_not_black(oracle):
    g = np.array([[0, 127, 128, 255]], np.uint8)
    assert oracle.threshold_binary(g).tolist() == [[0, 0, 255, 255]]  # A.3: src > 127 ? 255 : 0
    # an all-black image shifted by one column: the vacated column is white border -> not counted
    a = np.zeros((3, 4), np.uint8)
    out = oracle.warp_affine(a, [1, 0, 1, 0, 1, 0])
    assert oracle.vertical_projection(out).tolist() == [0, 3, 3, 3]


def test_argmax_tie_policy(oracle):
    # unique and equal maxima -> that index (projection.rs:153-157)
    idx, acc = oracle.argmax_path1([1, 5, 2], [1, 7, 2])
    assert idx == 1 and acc.tolist() == [1]
    # maxima disagree -> candidate with the larger v^2 + h^2 (:159-181)
    idx, acc = oracle.argmax_path1([1, 5, 2], [9, 1, 2])
    assert idx == 0 and acc.tolist() == [0]  # 1+81 = 82 > 25+1
    # exact tie of v^2+h^2: HashMap order decides in the reference -> both acceptable, lowest returned
    idx, acc = oracle.argmax_path1([3, 4, 0], [4, 3, 0])
    assert idx == 0 and acc.tolist() == [0, 1]
    # index 0 holding the maximum is listed twice by the reference (seed + revisit) -> goes through the
    # candidate branch but still wins
    idx, acc = oracle.argmax_path1([9, 1, 1], [9, 1, 1])
    assert idx == 0 and acc.tolist() == [0]
    # blank image: every score 0 -> no candidate beats 0.0 -> len / 2 (:183-186)
    idx, acc = oracle.argmax_path1([0, 0, 0, 0], [0, 0, 0, 0])
    assert idx == 2 and acc.tolist() == [2]


def test_path2_status(oracle):
    # omr.rs:184-221 with N = 1, step 0.5: indices 0,1 -> angles -0.5, 0.0
    ang, st, cand = oracle.select_path2([1, 2], [3, 5], 1, 0.5)
    assert (ang, st, cand.tolist()) == (0.0, 0, [0.0])  # unique h max -> Believed
    ang, st, cand = oracle.select_path2([1, 2], [5, 5], 1, 0.5)
    assert (ang, st, cand.tolist()) == (0.0, 1, [0.0])  # h tie, v decides -> NeedCheck
    ang, st, cand = oracle.select_path2([2, 2], [5, 5], 1, 0.5)
    assert (ang, st, cand.tolist()) == (0.0, 2, [-0.5, 0.0])  # full tie -> NotAResult, angle 0.0
    ang, st, cand = oracle.select_path2([0, 0], [0, 0], 1, 0.5)
    assert st == 2 and cand.tolist() == [-0.5, 0.0]  # blank: ties at 0.0 from the initial maxima


def test_erode_and_area_resize_by_hand(oracle):
    g = np.full((5, 5), 200, np.uint8)
    g[2, 2] = 10
    e1 = oracle.erode_cross3(g, 1)  # A.6: 5-point min, border = +inf
    exp = np.full((5, 5), 200, np.uint8)
    for (y, x) in ((2, 2), (1, 2), (3, 2), (2, 1), (2, 3)):
        exp[y, x] = 10
    assert (e1 == exp).all()
    e3 = oracle.erode_cross3(g, 3)  # L1 ball of radius 3 clipped to the image
    yy, xx = np.mgrid[0:5, 0:5]
    assert (e3 == np.where(abs(yy - 2) + abs(xx - 2) <= 3, 10, 200)).all()
    # A.7 integer factor 2: (a+b+c+d+2) >> 2;  factor 3: round-half-even of sum / 9 in float
    a = np.array([[1, 2, 10, 20], [3, 4, 30, 41]], np.uint8)
    assert oracle.resize_area(a, 1, 2).tolist() == [[(1 + 2 + 3 + 4 + 2) >> 2, (10 + 20 + 30 + 41 + 2) >> 2]]
    b = np.arange(9, dtype=np.uint8).reshape(3, 3) * 3  # sum 108 -> 12
    assert oracle.resize_area(b, 1, 1).tolist() == [[12]]
    b[0, 0] = 4  # sum 112 -> 12.44 -> 12;  b[0,0] = 5 -> 113/9 = 12.56 -> 13
    assert oracle.resize_area(b, 1, 1).tolist() == [[12]]
    b[0, 0] = 5
    assert oracle.resize_area(b, 1, 1).tolist() == [[13]]


def test_rgb2gray_weights(oracle):
    # A.5: (c0*9798 + c1*19235 + c2*3735 + 16384) >> 15, c0 = first byte in memory
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [90, 90, 90]]], np.uint8)
    assert oracle.rgb2gray(px).tolist() == [[(255 * 9798 + 16384) >> 15, (255 * 19235 + 16384) >> 15,
                                             (255 * 3735 + 16384) >> 15, 90]]


def test_resize_linear_hand_computed(oracle):
    """OpenCV's bilinear resize kernel (11-bit coefficients), the two ways the reference reaches it:
    INTER_LINEAR (scale_self with scale > 1, transfer.rs:82-86) and INTER_AREA when an axis enlarges
    (omr.rs:114-126 without a clamp, quirk B7).  Values below are worked by hand from resize.cpp's formulas."""
    row = np.array([[0, 100]], np.uint8)
    # INTER_LINEAR 2 -> 4: fx = (dx + 0.5) / 2 - 0.5 = -0.25 (clamped to the first pixel), 0.25, 0.75, 1.25 (clamped)
    assert oracle.resize_linear(row, 1, 4, False).tolist() == [[0, 25, 75, 100]]
    # INTER_AREA 2 -> 4 (integer enlargement): every area-mode fraction is 0 -> pixel replication
    assert oracle.resize_linear(row, 1, 4, True).tolist() == [[0, 0, 100, 100]]
    assert oracle.resize_area(row, 1, 4).tolist() == [[0, 0, 100, 100]]  # resize_area dispatches to it
    # INTER_AREA 2 -> 3: fx = (dx + 1) - (sx + 1) * 1.5 = -0.5 -> 0, 0.5, 0 -> [0, 50, 100]
    assert oracle.resize_area(row, 1, 3).tolist() == [[0, 50, 100]]
    # the vertical pass uses the same tables; rows are clipped into the image instead of the coefficients
    col = np.array([[0], [100]], np.uint8)
    assert oracle.resize_linear(col, 4, 1, False)[:, 0].tolist() == [0, 25, 75, 100]
    assert oracle.resize_area(col, 3, 1)[:, 0].tolist() == [0, 50, 100]
    # one axis shrinks, the other enlarges: still the bilinear emulation for both axes
    img = np.array([[0, 100, 200, 40], [80, 20, 60, 240]], np.uint8)
    got = oracle.resize_area(img, 3, 2)
    assert got.shape == (3, 2)
    # x: 4 -> 2 in area mode: sx = 0, 2 and fx = 1 - 1*0.5 = 0.5, 2 - 3*0.5 = 0.5 -> mean of (0,1) and (2,3)
    # y: 2 -> 3: fractions 0, 0.5, edge row -> rows [r0, (r0 + r1) / 2, r1]
    assert got.tolist() == [[50, 120], [50, 135], [50, 150]]
    # 3-channel interleaved data moves per channel
    rgb = np.stack([row, row[:, ::-1], row], axis=2)
    out = oracle.resize_linear(rgb, 1, 4, False)
    assert out[0, :, 0].tolist() == [0, 25, 75, 100] and out[0, :, 1].tolist() == [100, 75, 25, 0]
    # scale_self / shrink_to helpers (transfer.rs:66-126)
    big = oracle.scale_self(np.arange(12, dtype=np.uint8).reshape(3, 4) * 20, 1.5)
    assert big.shape == (4, 6)  # (3 * 1.5) as i32 = 4, (4 * 1.5) as i32 = 6
    assert oracle.shrink_to(img, 100, 100).tolist() == img.tolist()  # never enlarges
    assert oracle.shrink_to(img, 2, 0).shape == (1, 2)
