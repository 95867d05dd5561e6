"""Build-container script: makes the dataset pin's fixtures.

  1. copies the reference's test data files dataset/dataset/*.jpg (104 grey JPEG sheets, 17 MB; data the
     reference's own tests read, lib.rs:27,143) byte for byte into tests/golden/dataset/;
  2. runs the CPU oracle through the lib.rs:132-205 protocol on 9 seeded angles per sheet (936 cases) and
     writes tests/golden/dataset_pin_expected.json: per case the oracle's (angle bits, need_check,
     projection status) -- the GPU test must reproduce them bit for bit -- plus the class histogram.

Usage (here only; /root/reference does not exist on the GPU box):  python tests/golden/make_dataset_pin.py"""
import json
import multiprocessing as mp
import os
import shutil
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SRC = "/root/reference/dataset/dataset"


def _sheet(job):
    name, idxs = job
    import dataset_pin as dp
    from oracle import oracle as orc
    bgr = dp.imread_color(name)
    out = []
    for idx in idxs:
        ang = idx * 0.1
        x = dp.inject(bgr, ang, orc)
        det, chk, pst = dp.oracle_correct_default(x, orc)
        out.append({"sheet": name, "idx": idx, "angle_bits": struct.pack("<d", det).hex(), "angle": det,
                    "need_check": bool(chk), "proj_status": int(pst), "class": dp.classify(ang, det, chk)})
    return out


def main():
    import dataset_pin as dp
    from oracle import oracle as orc
    orc.build()
    os.makedirs(dp.DATASET, exist_ok=True)
    for f in sorted(os.listdir(SRC)):
        if f.lower().endswith(".jpg"):
            shutil.copyfile(os.path.join(SRC, f), os.path.join(dp.DATASET, f))
    jobs = {}
    for s, idx in dp.cases():
        jobs.setdefault(s, []).append(idx)
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(_sheet, sorted(jobs.items()))
    flat = [c for r in res for c in r]
    hist = {}
    for c in flat:
        hist[c["class"]] = hist.get(c["class"], 0) + 1
    json.dump({"params": list(dp.PARAMS), "protocol": "packages/lib/src/lib.rs:132-205 (PIL codec, oracle warp)",
               "histogram": hist, "cases": flat}, open(dp.EXPECTED, "w"), indent=0)
    print(len(flat), "cases", hist)


if __name__ == "__main__":
    main()
