"""Golden vectors of the Hough-line path, produced by the CPU oracle
(oracle/oracle_hough.c) -- "parity unpinned" like the sweep vectors (see make_golden.py).  Run from
the repo root:   python tests/golden/make_golden_hough.py
Inputs: seeded synthetic cards (gray and 3-channel) and the 248x230 gray reduction of one dataset
sheet that dataset_image001.npz already holds (a data file of the reference, not source)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
from oics import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case(name, img, mll, mlg):
    edges = orc.canny(img, 50.0, 150.0)
    lines = orc.hough_lines_p(edges, mll, mlg)
    ang = orc.line_angles_f32(lines)
    d = dict(img=img, min_line_length=mll, max_line_gap=mlg, edges_bits=np.packbits(edges != 0, axis=1, bitorder="little"),
             lines=lines, angle_bits=ang.view(np.uint32))
    if len(lines):
        a1 = orc.vote_hough_rs(ang)
        a2, st, cand = orc.vote_omr_rs(ang)
        d.update(hough_rs_angle_bits=np.float64(a1).view(np.uint64), omr_rs_angle_bits=np.float64(a2).view(np.uint64),
                 omr_rs_status=st, omr_rs_candidate_bits=cand.view(np.uint64))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, img.shape, "edges", int((edges != 0).sum()), "segments", len(lines))


def main():
    orc.build()
    g, _ = synth.make_card(48, 64, 3)
    case("hough_64x48_s3", g, 10.0, 2.0)
    g, _ = synth.make_card(230, 248, 4)
    case("hough_248x230_s4", g, 20.0, 5.0)
    case("hough_248x230_s4_long", g, 150.0, 50.0)
    g, _ = synth.make_card(97, 120, 5)
    rng = np.random.Generator(np.random.PCG64(105))
    bgr = np.clip(np.stack([g] * 3, axis=2).astype(np.int16) + rng.integers(-12, 13, (97, 120, 3), dtype=np.int16), 0, 255)
    case("hough_120x97_s5_bgr", bgr.astype(np.uint8), 20.0, 5.0)
    d = np.load(os.path.join(OUT, "dataset_image001.npz"))
    case("hough_dataset_image001", d["gray_small"], 40.0, 8.0)
    # FFT path (row f4): the two spectrum pictures of one card (oracle/oracle_fft.py, tolerance parity)
    from oracle import oracle_fft as offt
    g, _ = synth.make_card(230, 248, 4)
    m, lg = offt.get_fft_image(g)
    np.savez_compressed(os.path.join(OUT, "fft_248x230_s4.npz"), img=g, magnitude=m, magnitude_log=lg)


if __name__ == "__main__":
    main()
