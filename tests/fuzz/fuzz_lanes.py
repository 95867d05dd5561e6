"""Randomised parity run of the scan-lane sweep (development aid, not part of the test suite): random shapes, sweeps within
the scheme's +-10 degrees, batch sizes that leave partial scan groups and use every workgroup composition (1, 2, 4 scan
groups per workgroup; launches of 320 / 512 scans run two quads of scan groups), random content.  For every batch: the device-built programs against the host generator (dword for
dword), and for a sample of scans both std-dev vectors (f64 bits) and the arg-max against the CPU oracle.
Usage: python tests/fuzz/fuzz_lanes.py [batches] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

import oics
from oics import projection, synth
from oracle import oracle as orc

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
orc.build()
dev = torch.device("cuda:0")
bad, refused, checked = [], 0, 0
for c in range(batches):
    rows, cols = int(rng.integers(40, 1400)), int(rng.integers(33, 1400))
    max_angle = int(rng.integers(1, 11))
    step = float(rng.choice([1.0, 0.5, 0.25, 0.2]))
    n = int(rng.choice([1, 7, 64, 65, 100, 128, 130, 200, 257, 320, 400, 512]))
    lanes = int(rng.choice([64, 128, 256, 320, 512]))  # 320 / 512: two quads of scan groups in one launch (bench.py's form)
    scans = []
    for i in range(n):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            img = np.where(rng.random((rows, cols)) < rng.uniform(0.02, 0.6), 0, 255).astype(np.uint8)
        elif kind == 1:
            img = synth.make_binary_card(rows, cols, 1000 * c + i, skew=float(rng.uniform(-max_angle, max_angle)))[0]
        elif kind == 2:
            img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
        else:
            img = np.full((rows, cols), 255, np.uint8)
            img[rng.integers(0, rows, 20), :] = 0
            img[:, rng.integers(0, cols, 20)] = 0
        scans.append(np.ascontiguousarray(img))
    buf = torch.from_numpy(np.stack(scans)).to(dev)
    N, A = orc.candidate_count(max_angle, step)
    best = torch.zeros(n, dtype=torch.int32, device=dev)
    vs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((n, A), dtype=torch.float64, device=dev)
    b = projection.Batch(rows, cols, max_angle, step, n_streams=1)
    try:
        b.set_lanes(lanes)
    except oics.OmrError as e:
        assert e.code == -213
        refused += 1
        b.close()
        continue
    nd, diff = b.lanes_check_programs()
    if diff:
        bad.append(("programs", c, rows, cols, max_angle, step, diff))
    b.run_device(buf.data_ptr(), rows * cols, cols, n, 127, best.data_ptr(), vs.data_ptr(), hs.data_ptr())
    b.sync()
    b.close()
    bs, v, h = best.cpu().numpy(), vs.cpu().numpy(), hs.cpu().numpy()
    for i in sorted(set([0, n - 1, n // 2, min(n - 1, 63), min(n - 1, 64), min(n - 1, 256), min(n - 1, 300), min(n - 1, 319), min(n - 1, 448)])):
        binimg = np.where(scans[i] <= 127, 0, 255).astype(np.uint8)
        _, _, evs, ehs = orc.sweep(binimg, max_angle, step)
        checked += 1
        if not ((v[i].view(np.uint64) == evs.view(np.uint64)).all() and (h[i].view(np.uint64) == ehs.view(np.uint64)).all()
                and bs[i] == orc.argmax_path1(evs, ehs)[0]):
            bad.append(("scores", c, i, rows, cols, max_angle, step, n, lanes))
    # the same batch from HOST memory (omr_host_batch_run): binarised on the host, pageable u8 and PACKED transfers (the copier
    # threads pack to 1 bit per pixel, slane_pack_bits_kernel interleaves) must reproduce the resident run's bits
    if c % 3 == 0:
        hostscans = [np.where(sc <= 127, 0, 255).astype(np.uint8) for sc in scans]
        hb = projection.HostBatch(rows, cols, max_angle, step, n, n_devices=1)
        for packed in (False, True):
            hb_best, _, hb_v, hb_h = hb.run(hostscans, want_sd=True, packed=packed)
            if not ((hb_best == bs).all() and (hb_v.view(np.uint64) == v.view(np.uint64)).all() and (hb_h.view(np.uint64) == h.view(np.uint64)).all()):
                bad.append(("host batch", "packed" if packed else "pageable", c, rows, cols, max_angle, step, n))
        hb.close()
    print("batch %d: %dx%d +-%d @ %g, %d scans in launches of %d: %d mismatches so far" % (c, rows, cols, max_angle, step, n, lanes, len(bad)),
          flush=True)
print("batches", batches, "refused (-213)", refused, "scans checked", checked, "mismatches", len(bad), bad[:5])
