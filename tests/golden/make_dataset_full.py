"""Makes tests/golden/full/full_run_expected.npz: the CPU ORACLE's answer for every case of the reference's enabled
library test (packages/lib/src/lib.rs:130-245) -- 104 sheets x 900 injected angles (-45.0 .. +44.9 by 0.1,
lib.rs:153-154) = 93 600 runs of correct_default(45, 0.2, 248, 230, 150.0, 50.0) (lib.rs:192-205).

Per case (tests/dataset_pin.py): imread COLOR -> oracle rotate_mat(-angle, INTER_LINEAR, white, DEFAULT) -> RGB2GRAY ->
GRAY2RGB -> JPEG q100 round trip (PIL) -> the oracle's correct_default.  Stored per case: the detected angle's f64
bits, need_check, the projection status, and the CRC-32 of the image handed to correct_default (so a GPU-side run can
tell "same input, different answer" from "the injected image itself differed by a grey level").

CPU only, build container: about 0.13 s per case, 93 600 cases -> ~35 minutes on 6 processes.
Usage: python tests/golden/make_dataset_full.py [processes, default 6] [first sheet] [end sheet]
This is accuracy-level evidence (the reference holds no golden OUTPUTS): it pins GPU == oracle on
the reference's own data and protocol, not oracle == OpenCV."""
import os
import sys
import time
import zlib
from multiprocessing import get_context

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

OUT = os.path.join(HERE, "full", "full_run_expected.npz")
IDXS = list(range(-450, 450))


def one_sheet(name):
    import dataset_pin as dp
    from oracle import oracle as orc
    bgr = dp.imread_color(name)
    bits = np.zeros(len(IDXS), np.uint64)
    chk = np.zeros(len(IDXS), np.uint8)
    pst = np.zeros(len(IDXS), np.int8)
    crc = np.zeros(len(IDXS), np.uint32)
    for k, idx in enumerate(IDXS):
        inj = dp.inject(bgr, idx * 0.1, orc)
        ang, need, st = dp.oracle_correct_default(inj, orc)
        bits[k] = np.float64(ang).view(np.uint64)
        chk[k] = 1 if need else 0
        pst[k] = st
        crc[k] = zlib.crc32(inj.tobytes()) & 0xFFFFFFFF
    return name, bits, chk, pst, crc


def main():
    import dataset_pin as dp
    from oracle import oracle as orc
    orc.build()
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    names = dp.sheets()
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    end = int(sys.argv[3]) if len(sys.argv) > 3 else len(names)
    names = names[first:end]
    t0 = time.time()
    res = {}
    with get_context("spawn").Pool(procs) as pool:
        for n, (name, bits, chk, pst, crc) in enumerate(pool.imap_unordered(one_sheet, names)):
            res[name] = (bits, chk, pst, crc)
            print("%3d / %d sheets  %.0f s" % (n + 1, len(names), time.time() - t0), flush=True)
            # partial results survive an interrupted run
            np.savez_compressed(OUT + ".part.npz", sheets=np.array(sorted(res)), idx=np.array(IDXS, np.int32),
                                angle_bits=np.stack([res[s][0] for s in sorted(res)]),
                                need_check=np.stack([res[s][1] for s in sorted(res)]),
                                status=np.stack([res[s][2] for s in sorted(res)]),
                                input_crc32=np.stack([res[s][3] for s in sorted(res)]))
    order = sorted(res)
    np.savez_compressed(OUT, sheets=np.array(order), idx=np.array(IDXS, np.int32),
                        angle_bits=np.stack([res[s][0] for s in order]), need_check=np.stack([res[s][1] for s in order]),
                        status=np.stack([res[s][2] for s in order]), input_crc32=np.stack([res[s][3] for s in order]))
    os.remove(OUT + ".part.npz")
    print("wrote %s: %d sheets x %d angles, %.0f s" % (OUT, len(order), len(IDXS), time.time() - t0))


if __name__ == "__main__":
    main()
