"""The reference-held pin: packages/lib/src/lib.rs:103-113 -- |injected - detected| < 0.5 deg whenever
!need_check -- over the reference's own 104-sheet dataset with the lib.rs:132-205 protocol and
correct_default(45, 0.2, 248, 230, 150.0, 50.0).  See tests/dataset_pin.py for the protocol and
tests/golden/make_dataset_pin.py for how the fixtures were made.

  * CPU tier : the oracle on one angle per sheet (104 cases) must reproduce the committed expectations bit
               for bit and meet the reference's criterion; the committed 936-case histogram must hold no ERROR.
  * Criterion as asserted here: the reference's own `assert < 0.5` (lib.rs:105-112) for every believed case, with
    ONE exception named explicitly -- image051 at injected 28.9 is detected at 28.4, exactly ON the bound
    (|28.9 - 28.4| evaluates to 0.5 in f64; class NOT_SO_RIGHT by lib.rs:222-223, not ERROR).  The reference's own
    comment records the same experience ("one case above 0.5 was seen", lib.rs:108).  Any other believed case at
    or above 0.5 fails the tests.
  * GPU tier : all 936 cases through omr_correct_default (C ABI -> HIP kernels): the reference's criterion,
               bit-equality with the oracle's committed results, and the class histogram next to the
               reference's claim "99.9 % < 0.4 deg" (lib.rs:108), written to gpurun_out/ for DESIGN.md.
This is accuracy-level: parity stays "unpinned at bit level" (the reference holds no golden vectors)."""
import json
import os
import struct
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dataset_pin as dp  # noqa: E402


def _bits(x):
    return struct.pack("<d", float(x)).hex()


def test_fixture_is_the_whole_reference_dataset():
    names = dp.sheets()
    assert len(names) == 104  # 100 x image0NN.jpg + 4 x SCN000NN_2.jpg
    assert sum(n.startswith("image") for n in names) == 100 and sum(n.startswith("SCN") for n in names) == 4
    exp = dp.load_expected()
    assert exp["params"] == list(dp.PARAMS)
    assert [(c["sheet"], c["idx"]) for c in exp["cases"]] == dp.cases()
    assert len(exp["cases"]) == 104 * dp.ANGLES_PER_SHEET >= 900
    assert all(-450 <= c["idx"] < 450 for c in exp["cases"])  # lib.rs:153


# the one believed case that sits exactly on the reference's bound (see the module docstring): (sheet, injected index)
BOUNDARY_CASE = ("image051.jpg", 289)


def _meets_criterion(sheet, idx, injected, detected):
    """lib.rs:105-112: assert < 0.5; the named boundary case may be equal to it."""
    d = abs(injected - detected)
    return d <= 0.5 if (sheet, idx) == BOUNDARY_CASE else d < 0.5


def test_committed_histogram_meets_the_reference_criterion():
    exp = dp.load_expected()
    hist, edge = {}, 0
    for c in exp["cases"]:
        inj = c["idx"] * 0.1
        cls = dp.classify(inj, c["angle"], c["need_check"])
        hist[cls] = hist.get(cls, 0) + 1
        assert cls == c["class"]
        if not c["need_check"]:
            assert _meets_criterion(c["sheet"], c["idx"], inj, c["angle"]), (c["sheet"], inj, c["angle"])  # lib.rs:105-112
            edge += abs(inj - c["angle"]) >= 0.5
    assert edge == 1  # exactly the named boundary case
    assert hist == exp["histogram"] and hist.get("ERROR", 0) == 0
    believed = sum(v for k, v in hist.items() if k != "NOT_BELIEVED")
    assert believed >= 0.99 * len(exp["cases"])  # NOT_BELIEVED is rare (3 of 936 when the fixtures were made)


def test_oracle_reproduces_the_committed_results(oracle):
    """One case per sheet on the CPU (the whole set takes minutes single-threaded; make_dataset_pin.py ran it)."""
    exp = dp.load_expected()["cases"]
    first = {}
    for c in exp:
        first.setdefault(c["sheet"], c)

    def one(c):
        bgr = dp.imread_color(c["sheet"])
        x = dp.inject(bgr, c["idx"] * 0.1, oracle)
        det, chk, pst = dp.oracle_correct_default(x, oracle)
        return c, det, chk, pst

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for c, det, chk, pst in ex.map(one, list(first.values())):
            assert _bits(det) == c["angle_bits"] and chk == c["need_check"] and pst == c["proj_status"], c["sheet"]
            if not chk:
                assert _meets_criterion(c["sheet"], c["idx"], c["idx"] * 0.1, det)


@pytest.mark.gpu
def test_correct_default_on_the_dataset_gpu(oracle):
    """All 936 cases through the product path; fixture inputs (skew injection, JPEG round trip) are made
    with the oracle / PIL exactly as for the committed expectations, so the GPU must match them bit for bit."""
    from oics import omr
    exp = dp.load_expected()["cases"]
    by_sheet = {}
    for c in exp:
        by_sheet.setdefault(c["sheet"], []).append(c)

    def prepare(item):
        name, cs = item
        bgr = dp.imread_color(name)
        return [(c, dp.inject(bgr, c["idx"] * 0.1, oracle)) for c in cs]

    hist, mism, worst, n, edge = {}, [], 0.0, 0, 0
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for batch in ex.map(prepare, sorted(by_sheet.items())):
            for c, x in batch:
                ang, chk, _ = omr.correct_default(x, *dp.PARAMS, want_image=False)
                inj = c["idx"] * 0.1
                cls = dp.classify(inj, ang, chk)
                hist[cls] = hist.get(cls, 0) + 1
                n += 1
                if not chk:
                    worst = max(worst, abs(inj - ang))
                    assert _meets_criterion(c["sheet"], c["idx"], inj, ang), (c["sheet"], inj, ang)  # lib.rs:105-112
                    edge += abs(inj - ang) >= 0.5
                if _bits(ang) != c["angle_bits"] or chk != c["need_check"]:
                    mism.append((c["sheet"], c["idx"], ang, c["angle"], chk, c["need_check"]))
    assert n == len(exp) >= 900
    assert not mism, "GPU differs from the oracle's committed results: %s" % mism[:5]
    assert hist.get("ERROR", 0) == 0 and edge <= 1
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        json.dump({"cases": n, "histogram": hist, "worst_believed_error_deg": worst, "believed_cases_at_0p5": edge,
                   "reference_claim": "99.9 % < 0.4 deg, ~100 % < 0.5 deg (lib.rs:108)",
                   "success_rate": hist.get("SUCCESS", 0) / n}, open(os.path.join(out, "dataset_pin_gpu.json"), "w"), indent=1)


@pytest.mark.gpu
def test_full_run_error_cases_are_the_oracles_too(oracle, golden_dir):
    """The reference's library test in full (104 sheets x 900 angles, tools/dataset_full.py on an MI355X,
    profiles/r03_dataset_full.md) left 34 believed cases with |injected - detected| > 0.5 deg -- the reference's own
    assertion (lib.rs:105-112) would fail on them.  They are the algorithm's misses, not the port's: on every one of them
    the CPU oracle detects the same angle, bit for bit, and it is the angle the full run recorded."""
    from oics import omr
    cases = json.load(open(os.path.join(golden_dir, "dataset_full_errors.json")))["cases"]
    assert len(cases) == 34

    def one(c):
        sheet, idx, recorded = c
        x = dp.inject(dp.imread_color(sheet), idx * 0.1, oracle)
        det, chk, _ = dp.oracle_correct_default(x, oracle)
        return c, x, det, chk

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for (sheet, idx, recorded), x, det, chk in ex.map(one, cases):
            ang, gchk, _ = omr.correct_default(x, *dp.PARAMS, want_image=False)
            assert _bits(ang) == _bits(det) and gchk == chk, (sheet, idx, ang, det)
            assert not gchk and abs(ang - recorded) < 1e-9 and abs(idx * 0.1 - ang) > 0.5, (sheet, idx, ang, recorded)


FULL = os.path.join(HERE, "golden", "full", "full_run_expected.npz")


def _full_expected():
    d = np.load(FULL)
    return [str(x) for x in d["sheets"]], d["idx"], d["angle_bits"], d["need_check"], d["input_crc32"]


def test_full_expectation_is_complete_and_reproduces_the_run():
    """tests/golden/full/full_run_expected.npz (make_dataset_full.py, CPU oracle): all 104 x 900 cases of lib.rs:130-245.
    Its class histogram is the one the GPU run recorded (profiles/r03_dataset_full.md) -- profiles/r04_dataset_diff.md is
    the case-by-case diff of that run against this file: 93 600 of 93 600 identical -- and it agrees with the 936-case
    fixture wherever the two overlap."""
    sheets, idx, bits, chk, crc = _full_expected()
    assert sheets == dp.sheets() and len(sheets) == 104 and list(idx) == list(range(-450, 450))
    assert bits.shape == chk.shape == crc.shape == (104, 900)
    ang = bits.view(np.float64)
    inj = np.asarray(idx, np.float64)[None, :] * 0.1
    d = np.abs(inj - ang)
    believed = chk == 0
    hist = {"NOT_BELIEVED": int((~believed).sum()), "ERROR": int((believed & (d > 0.5)).sum()),
            "NOT_SO_RIGHT": int((believed & (d > 0.4) & ~(d > 0.5)).sum()), "SUCCESS": int((believed & ~(d > 0.4)).sum())}
    assert hist == {"SUCCESS": 92160, "NOT_SO_RIGHT": 647, "ERROR": 34, "NOT_BELIEVED": 759}
    srow = {s: k for k, s in enumerate(sheets)}
    for c in dp.load_expected()["cases"]:
        a, b = srow[c["sheet"]], c["idx"] + 450
        assert _bits(ang[a, b]) == c["angle_bits"] and bool(chk[a, b]) == c["need_check"], (c["sheet"], c["idx"])


def stratified_cases():
    """every tenth angle with a per-sheet offset (9 360 cases), plus all of 35 <= |angle| < 40 deg -- where the
    projection and the edges result disagree most (740 of the 759 NOT_BELIEVED cases) -- for four sheets"""
    sheets = dp.sheets()
    out = []
    for k, s in enumerate(sheets):
        picks = set(range(-450 + (k % 10), 450, 10))
        if k in (0, 33, 66, 99):
            picks |= {i for i in range(-450, 450) if 350 <= abs(i) < 400}
        out += [(s, i) for i in sorted(picks)]
    return out


@pytest.mark.gpu
def test_stratified_tenth_of_the_full_run_gpu(oracle):
    """~9 560 cases of the full protocol through omr_correct_default, inputs made by the oracle / PIL as for the
    expectation (the CRC-32 of every input is checked against the recorded one): detected angle and need_check must
    equal the oracle's bit for bit.  Accuracy-level evidence on the reference's own data (the reference holds no
    golden outputs): it pins GPU == oracle, not oracle == OpenCV."""
    import zlib
    from oics import omr
    sheets, idx, bits, chk, crc = _full_expected()
    srow = {s: k for k, s in enumerate(sheets)}
    by_sheet = {}
    for s, i in stratified_cases():
        by_sheet.setdefault(s, []).append(i)

    def prepare(item):
        name, idxs = item
        bgr = dp.imread_color(name)
        return name, [(i, dp.inject(bgr, i * 0.1, oracle)) for i in idxs]

    n, mism = 0, []
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for name, batch in ex.map(prepare, sorted(by_sheet.items())):
            a = srow[name]
            for i, x in batch:
                b = i + 450
                assert (zlib.crc32(x.tobytes()) & 0xFFFFFFFF) == int(crc[a, b]), "input differs from the recorded one: %s %d" % (name, i)
                ang, need, _ = omr.correct_default(x, *dp.PARAMS, want_image=False)
                n += 1
                if np.float64(ang).view(np.uint64) != bits[a, b] or bool(need) != bool(chk[a, b]):
                    mism.append((name, i, ang, float(bits[a, b].view(np.float64)), need, bool(chk[a, b])))
    assert n >= 9360
    assert not mism, "GPU differs from the oracle on %d of %d cases: %s" % (len(mism), n, mism[:5])


@pytest.mark.gpu
def test_core_protocol_on_the_dataset_gpu():
    """packages/core/src/main.rs:17-252, the reference's comparative benchmark, through the drop-in API (tools/
    core_protocol.py): on the reference's 104 sheets, skewed by seeded angles in [-10, 10), the projection and the
    Hough-line method must find the angle (the reference holds no numbers for this protocol; measured: mean 0.13 /
    0.19 deg, max 0.58 / 0.46 deg), and the FFT method, which answers 0 when the spectrum's axis cross wins the
    vote, must be right where it answers something else."""
    import importlib.util
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "core_protocol.py")
    spec = importlib.util.spec_from_file_location("core_protocol", tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    j = mod.run(2024)
    assert j["sheets"] == 104
    for m, mean_max, frac in (("projection", 0.25, 0.95), ("hough", 0.30, 0.95)):
        r = j[m]
        assert r["answered"] == 104 and r["deviation_mean_deg"] < mean_max and r["within_0.5_deg"] >= frac * 104, (m, r)
        assert r["deviation_max_deg"] < 1.0, (m, r)
    assert j["fft"]["answered"] == 104 and j["fft"]["within_0.5_deg"] >= 40, j["fft"]
