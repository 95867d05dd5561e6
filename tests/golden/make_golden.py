"""Generates the golden vectors under tests/golden/ with the CPU oracle (oracle/liboracle.so).

The reference ships no golden vectors for this path and cannot run here (Rust + OpenCV 4.6.0,
SURVEY.md 8c), so these are outputs of the build's own oracle -- "parity unpinned" -- committed so
that (a) the oracle cannot drift silently and (b) the GPU box, where /root/reference does not
exist, can check the HIP path against fixed numbers.  Run from the repo root:
    python tests/golden/make_golden.py
Inputs: seeded synthetic cards (oics/synth.py) and four sheets of the reference's dataset
(/root/reference/dataset/dataset/*.jpg, decoded with PIL, reduced to 248x230 by the oracle's
INTER_AREA restatement -- data files, not source).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
from oics import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pack(bin_img):
    return np.packbits(bin_img == 0, axis=1, bitorder="little"), bin_img.shape


def sweep_case(name, bin_img, max_angle, step, scale=1.0, extra=None):
    vp, hp, vs, hs = orc.sweep(bin_img, max_angle, step, scale)
    idx, accept = orc.argmax_path1(vs, hs)
    N, A = orc.candidate_count(max_angle, step)
    ang2, st2, cand2 = orc.select_path2(vs, hs, N, step)
    bits, shape = pack(bin_img)
    d = dict(black_bits=bits, shape=np.array(shape), max_angle=max_angle, step=step, matrix_scale=scale,
             vproj=vp.astype(np.uint16), hproj=hp.astype(np.uint16), v_sd_bits=vs.view(np.uint64),
             h_sd_bits=hs.view(np.uint64), argmax=idx, accept=accept, path2_angle=ang2, path2_status=st2,
             path2_candidates=cand2)
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, bin_img.shape, "A=%d" % A, "argmax", idx, "angle", (idx - N) * step, "black %.3f" % (bin_img == 0).mean())


def main():
    orc.build()
    # (1) synthetic cards
    for (r, c, seed, ma, st) in ((48, 64, 11, 5, 0.5), (230, 248, 12, 45, 0.2), (512, 512, 1, 5, 0.5),
                                 (230, 248, 13, 10, 0.05)):
        b, theta = synth.make_binary_card(r, c, seed, skew=None if r != 512 else 2.4)
        sweep_case("synth_%dx%d_s%d" % (c, r, seed), b, ma, st, extra=dict(injected=theta))
    # quirk B4: rotation-matrix scale != 1 (omr.rs:159-163)
    b, theta = synth.make_binary_card(230, 248, 14)
    sweep_case("synth_248x230_s14_scale0p2", b, 10, 0.5, scale=0.2, extra=dict(injected=theta))
    # (2) dataset sheets through the path-2 front end (gray -> erode x3 -> INTER_AREA 5x -> threshold)
    ds = "/root/reference/dataset/dataset"
    if os.path.isdir(ds):
        from PIL import Image
        rng = np.random.Generator(np.random.PCG64(2019))
        for fn in ("image001.jpg", "image042.jpg", "image100.jpg", "SCN00025_2.jpg"):
            g = np.array(Image.open(os.path.join(ds, fn)).convert("L"))
            theta = float(rng.uniform(-9.5, 9.5))
            # lib.rs:156-166 style skew injection: LINEAR / DEFAULT rotate by -theta
            skewed = orc.rotate_mat(g, -theta, 1.0, interp=1, clip=0)
            er = orc.erode_cross3(skewed, 3)
            sc = min(248.0 / g.shape[1], 230.0 / g.shape[0])
            dc, dr = int(g.shape[1] * sc), int(g.shape[0] * sc)
            small = orc.resize_area(er, dr, dc)
            b = orc.threshold_binary(small)
            sweep_case("dataset_" + fn.split(".")[0], b, 45, 0.2, scale=sc,
                       extra=dict(injected=theta, gray_small=small, resize_scale=sc))
    # (3) front-end vectors: rgb2gray / erode / area resize (integer and fractional factor)
    rng = np.random.Generator(np.random.PCG64(7))
    rgb = rng.integers(0, 256, (60, 85, 3), dtype=np.uint8)
    gray = orc.rgb2gray(rgb)
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), rgb=rgb, gray=gray, eroded=orc.erode_cross3(gray, 3),
                        area_5x=orc.resize_area(gray, 12, 17), area_2x=orc.resize_area(gray, 30, 42),
                        area_frac=orc.resize_area(gray, 23, 31), area_rgb_5x=orc.resize_area(rgb, 12, 17),
                        area_rgb_frac=orc.resize_area(rgb, 25, 36),
                        warp_lin=orc.rotate_mat(rgb, 7.3, 1.0, interp=1, clip=1),
                        warp_nn=orc.rotate_mat(rgb, -12.6, 1.0, interp=0, clip=1))
    print("frontend ok")


if __name__ == "__main__":
    main()
