"""The CPU oracle against the committed golden vectors, against its independent numpy
restatement, and against the reference's only numeric acceptance criterion
(|detected - injected| < 0.5 deg, packages/lib/src/lib.rs:103-113)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle_np as onp


def load(path):
    d = np.load(path)
    rows, cols = [int(v) for v in d["shape"]]
    black = np.unpackbits(d["black_bits"], axis=1, bitorder="little")[:, :cols].astype(bool)
    return d, np.where(black, 0, 255).astype(np.uint8)


def sweep_files(golden_dir):
    return sorted(f for f in glob.glob(os.path.join(golden_dir, "*.npz")) if "frontend" not in f and "hough" not in f and "fft_" not in f)


def test_golden_present(golden_dir):
    assert len(sweep_files(golden_dir)) >= 9


@pytest.mark.parametrize("name", ["synth_64x48_s11", "synth_512x512_s1", "synth_248x230_s13",
                                  "synth_248x230_s14_scale0p2", "dataset_image001", "dataset_SCN00025_2"])
def test_oracle_reproduces_golden(oracle, golden_dir, name):
    d, b = load(os.path.join(golden_dir, name + ".npz"))
    vp, hp, vs, hs = oracle.sweep(b, int(d["max_angle"]), float(d["step"]), float(d["matrix_scale"]))
    assert (vp == d["vproj"]).all() and (hp == d["hproj"]).all()
    assert (vs.view(np.uint64) == d["v_sd_bits"]).all() and (hs.view(np.uint64) == d["h_sd_bits"]).all()
    idx, acc = oracle.argmax_path1(vs, hs)
    assert idx == int(d["argmax"]) and acc.tolist() == d["accept"].tolist()
    N, _ = oracle.candidate_count(int(d["max_angle"]), float(d["step"]))
    ang, st, cand = oracle.select_path2(vs, hs, N, float(d["step"]))
    assert ang == float(d["path2_angle"]) and st == int(d["path2_status"])
    assert cand.tolist() == d["path2_candidates"].tolist()


def test_threads_do_not_change_results(oracle, golden_dir):
    d, b = load(os.path.join(golden_dir, "synth_248x230_s13.npz"))
    r1 = oracle.sweep(b, 10, 0.05, threads=1)
    r4 = oracle.sweep(b, 10, 0.05, threads=4)
    r4f = oracle.sweep(b, 10, 0.05, threads=4, fast=True)
    for x, y, z in zip(r1, r4, r4f):
        assert (x == y).all() and (x == z).all()


@pytest.mark.parametrize("name", ["synth_64x48_s11", "synth_512x512_s1", "synth_248x230_s14_scale0p2"])
def test_c_oracle_equals_numpy_restatement(oracle, golden_dir, name):
    d, b = load(os.path.join(golden_dir, name + ".npz"))
    vp, hp, vs, hs = onp.sweep(b, int(d["max_angle"]), float(d["step"]), float(d["matrix_scale"]))
    assert (vp == d["vproj"]).all() and (hp == d["hproj"]).all()
    assert (vs.view(np.uint64) == d["v_sd_bits"]).all() and (hs.view(np.uint64) == d["h_sd_bits"]).all()
    assert onp.argmax_path1(vs, hs) == set(d["accept"].tolist())


def test_accuracy_criterion(golden_dir):
    # lib.rs:103-113: abs(injected - detected) < 0.5 whenever the projection result is believed
    n = 0
    for f in sweep_files(golden_dir):
        d = np.load(f)
        if "64x48" in f:
            continue  # injected skew lies outside that vector's +-5 deg range
        N = int(float(d["max_angle"]) / float(d["step"]))
        detected = (int(d["argmax"]) - N) * float(d["step"])
        assert abs(detected - float(d["injected"])) < 0.5, f
        if int(d["path2_status"]) == 0:
            assert abs(float(d["path2_angle"]) - float(d["injected"])) < 0.5, f
        n += 1
    assert n >= 8


def test_frontend_golden(oracle, golden_dir):
    d = np.load(os.path.join(golden_dir, "frontend.npz"))
    gray = oracle.rgb2gray(d["rgb"])
    assert (gray == d["gray"]).all()
    assert (oracle.erode_cross3(gray, 3) == d["eroded"]).all()
    assert (oracle.resize_area(gray, 12, 17) == d["area_5x"]).all()
    assert (oracle.resize_area(gray, 30, 42) == d["area_2x"]).all()
    assert (oracle.resize_area(gray, 23, 31) == d["area_frac"]).all()
    assert (oracle.resize_area(d["rgb"], 12, 17) == d["area_rgb_5x"]).all()
    assert (oracle.resize_area(d["rgb"], 25, 36) == d["area_rgb_frac"]).all()
    assert (oracle.rotate_mat(d["rgb"], 7.3, 1.0, interp=1, clip=1) == d["warp_lin"]).all()
    assert (oracle.rotate_mat(d["rgb"], -12.6, 1.0, interp=0, clip=1) == d["warp_nn"]).all()


def test_drivers_agree_with_sweep(oracle, golden_dir):
    d, b = load(os.path.join(golden_dir, "synth_512x512_s1.npz"))
    rgb = np.repeat(b[:, :, None], 3, axis=2)
    ang, idx = oracle.get_angle_with_projections(rgb, 5, 0.5, 1.0)
    assert idx == int(d["argmax"]) and ang == (idx - 10) * 0.5
    ang2, st2, cand2 = oracle.get_result_from_projection(rgb, 5, 0.5, 0, 0)
    assert st2 in (0, 1, 2)
