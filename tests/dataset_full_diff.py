"""Offline diff of a full GPU dataset run against the oracle's expectation (round-3 verdict item 2).

  GPU side : tools/dataset_full.py on the GPU box -> part files with one row per case
             (sheet, test_iter_idx, injected, detected, need_check[, input_crc32])
  oracle   : tests/golden/full/full_run_expected.npz (tests/golden/make_dataset_full.py, CPU): detected-angle f64 bits,
             need_check, projection status and the CRC-32 of the image handed to correct_default, for all
             104 x 900 cases of lib.rs:130-245

Every GPU row is compared with the oracle's answer for the same (sheet, angle): detected angle bit for bit and
need_check.  Rows that carry input_crc32 also tell whether the skew-injected image itself was byte-identical (the GPU's
INTER_LINEAR warp is held to <= 1 grey level of the oracle's, north_star's tolerance, so an injected image may differ
by a level somewhere and the JPEG round trip may then differ too: such a case is "input differs", not a detection
mismatch).  Writes a markdown summary.  Usage:
  python tests/dataset_full_diff.py out.md part1.json [part2.json ...]"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EXPECTED = os.path.join(HERE, "golden", "full", "full_run_expected.npz")


def load_expected():
    d = np.load(EXPECTED)
    sheets = [str(s) for s in d["sheets"]]
    idx = {int(v): k for k, v in enumerate(d["idx"])}
    return sheets, idx, d["angle_bits"], d["need_check"], d["status"], d["input_crc32"]


def diff(parts):
    sheets, idx, bits, chk, status, crc = load_expected()
    srow = {s: k for k, s in enumerate(sheets)}
    n = same = same_input = diff_input = 0
    mism = []
    hist_gpu = {"SUCCESS": 0, "NOT_SO_RIGHT": 0, "ERROR": 0, "NOT_BELIEVED": 0}
    hist_orc = dict(hist_gpu)

    def cls(inj, det, need):
        d = abs(inj - det)
        return "NOT_BELIEVED" if need else "ERROR" if d > 0.5 else "NOT_SO_RIGHT" if d > 0.4 else "SUCCESS"

    for part in parts:
        for row in json.load(open(part))["rows"]:
            s, i, inj, det, need = row[:5]
            if s not in srow or int(i) not in idx:
                continue
            a, b = srow[s], idx[int(i)]
            n += 1
            edet = float(np.uint64(bits[a, b]).view(np.float64))
            eneed = bool(chk[a, b])
            hist_gpu[cls(inj, det, need)] += 1
            hist_orc[cls(inj, edet, eneed)] += 1
            input_same = None
            if len(row) > 5:
                input_same = int(row[5]) == int(crc[a, b])
                same_input += input_same
                diff_input += not input_same
            if np.float64(det).view(np.uint64) == bits[a, b] and bool(need) == eneed:
                same += 1
            else:
                mism.append((s, int(i), inj, det, bool(need), edet, eneed, input_same))
    return n, same, mism, hist_gpu, hist_orc, same_input, diff_input


def main():
    out, parts = sys.argv[1], sys.argv[2:]
    n, same, mism, hg, ho, si, di = diff(parts)
    with open(out, "w") as f:
        f.write("# Full dataset run: GPU (`omr_correct_default` through the C ABI) against the CPU oracle, case by case\n\n")
        f.write("%d cases compared (104 sheets x 900 injected angles, lib.rs:130-245); detected angle (f64 bits) and need_check "
                "identical in **%d** (%.4f %%); %d differ.\n\n" % (n, same, 100.0 * same / max(n, 1), len(mism)))
        if si + di:
            f.write("Skew-injected input image (after the JPEG round trip) byte-identical to the oracle's in %d cases, different in %d "
                    "(the GPU's INTER_LINEAR warp is held to <= 1 grey level of the oracle's).\n\n" % (si, di))
        f.write("| class (lib.rs:220-226) | GPU | oracle |\n|---|---|---|\n")
        for k in ("SUCCESS", "NOT_SO_RIGHT", "ERROR", "NOT_BELIEVED"):
            f.write("| %s | %d | %d |\n" % (k, hg[k], ho[k]))
        if mism:
            f.write("\n| sheet | injected | GPU detected / check | oracle detected / check | same input |\n|---|---|---|---|---|\n")
            for s, i, inj, det, need, edet, eneed, ins in mism[:60]:
                f.write("| %s | %.1f | %.2f / %s | %.2f / %s | %s |\n" % (s, inj, det, need, edet, eneed,
                                                                      "-" if ins is None else "yes" if ins else "no"))
    print(open(out).read())
    return 0 if not mism else 1


if __name__ == "__main__":
    sys.exit(main())
