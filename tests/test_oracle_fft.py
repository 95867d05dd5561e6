"""The FFT path of the CPU oracle (oracle/oracle_fft.py): hand-computed micro cases and the golden picture."""
import os

import numpy as np

from oracle import oracle_fft as offt


def test_fft_shift_quadrants_and_odd_sizes():
    a = np.arange(16, dtype=np.float32).reshape(4, 4)
    s = offt.fft_shift(a)
    assert s.tolist() == [[10, 11, 8, 9], [14, 15, 12, 13], [2, 3, 0, 1], [6, 7, 4, 5]]
    b = np.arange(15, dtype=np.float32).reshape(3, 5)      # cx = 2, cy = 1: row 2 and column 4 stay (fft.rs:69-74)
    s = offt.fft_shift(b)
    assert s[2].tolist() == b[2].tolist() and s[:, 4].tolist() == b[:, 4].tolist()
    assert s[0, :4].tolist() == [7, 8, 5, 6] and s[1, :4].tolist() == [2, 3, 0, 1]


def test_spectrum_of_a_constant_and_of_a_cosine():
    g = np.full((8, 8), 255, np.uint8)
    re, im = offt.spectrum(g)
    assert re[4, 4] == 1.0 and np.abs(re).sum() == 1.0 and not im.any()   # DFT_SCALE: mean = 1 at the swapped DC
    x = np.arange(16)
    g = np.tile(np.round(127.5 + 127.5 * np.cos(2 * np.pi * 2 * x / 16)).astype(np.uint8), (16, 1))
    re, im = offt.spectrum(g)
    mag = np.hypot(re, im)
    assert mag[8, 8] > 0.49 and abs(mag[8, 10] - 0.25) < 0.01 and abs(mag[8, 6] - 0.25) < 0.01
    assert mag[7].max() < 1e-6


def test_pictures_saturation_quirk():
    rng = np.random.Generator(np.random.PCG64(3))
    g = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    m, lg = offt.get_fft_image(g)
    # magnitude_image is multiplied by 255 twice (fft.rs:110 and :134): everything but the faintest bins saturates
    assert (m == 255).mean() > 0.9 and m.min() == 0
    assert lg.max() == 255 and lg.min() == 0 and lg[16, 16] == 255


def test_vote_quirk():
    # fft.rs:231 re-reads line i: the first segment whose raw angle lies in [-45, 45] wins once there are two
    lines = [[0, 0, 0, 50], [0, 0, 100, 10], [0, 0, 100, -3]]          # 90 deg, 5.7 deg, -1.7 deg
    assert abs(offt.vote_fft_rs(lines) - np.degrees(np.arctan2(10, 100))) < 1e-12
    assert offt.vote_fft_rs(lines[:1]) == 0.0 and offt.vote_fft_rs([]) == 0.0
    assert offt.vote_fft_rs([[0, 0, 100, 10]]) == 0.0                   # a single line never gets a vote
    assert offt.vote_fft_rs([[0, 0, 0, 50], [0, 0, 1, 50]]) == 0.0      # only steep lines: nothing is chosen


def test_golden_picture(golden_dir):
    d = np.load(os.path.join(golden_dir, "fft_248x230_s4.npz"))
    m, lg = offt.get_fft_image(d["img"])
    # pocketfft in double precision is deterministic up to the last ulp; allow the picture tolerance
    assert np.abs(lg.astype(np.int16) - d["magnitude_log"].astype(np.int16)).max() <= 1
    assert (lg == d["magnitude_log"]).mean() > 0.999
    assert np.abs(m.astype(np.int16) - d["magnitude"].astype(np.int16)).max() <= 1
