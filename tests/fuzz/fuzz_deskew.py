"""Randomised parity run of the batch deskew (development aid, not part of the test suite): random shapes (widths that
are and are not multiples of 4: staged and unstaged warp paths), candidate ranges up to 45 degrees, launch groups and
skews through omr_batch_deskew_device against the CPU oracle's rotate_mat -- NEAREST exact, LINEAR within one grey level.
Usage: python tests/fuzz/fuzz_deskew.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import projection, synth
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
orc.build()
dev = torch.device("cuda:0")
bad = []
for c in range(cases):
    rows, cols = int(rng.integers(40, 900)), int(rng.integers(40, 900))
    if c % 3 == 0:
        cols = (cols + 3) & ~3  # whole dwords: the LDS-staged warp
    max_angle = int(rng.choice([3, 10, 20, 45]))
    step = float(rng.choice([1.0, 0.5, 0.25]))
    n = int(rng.integers(1, 7))
    group = int(rng.choice([1, 2, 4, 8]))
    interp = int(rng.integers(0, 2))
    border = int(rng.choice([255, 0, 93]))
    skews = rng.uniform(-max_angle, max_angle, n)
    cards = np.stack([synth.make_card(rows, cols, 1000 * c + i, skew=float(s))[0] for i, s in enumerate(skews)])
    scans = torch.from_numpy(cards).to(dev)
    b = projection.Batch(rows, cols, max_angle, step, device=0, n_streams=1)
    b.set_group(group)
    dr, dc = b.deskew_canvas()
    out = torch.full((n, dr, dc), 7, dtype=torch.uint8, device=dev)
    size = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    best = torch.full((n,), -1, dtype=torch.int32, device=dev)
    b.deskew_device(scans.data_ptr(), rows * cols, cols, n, 127, interp, border, out.data_ptr(), dr * dc, dc, size.data_ptr(),
                    best.data_ptr())
    b.sync()
    N = b.N
    b.close()
    o, sz, bs = out.cpu().numpy(), size.cpu().numpy(), best.cpu().numpy()
    for i in range(n):
        angle = (int(bs[i]) - N) * step
        exp = orc.rotate_mat(cards[i], angle, 1.0, interp, (border, border, border, 0), 1)
        er, ec = exp.shape
        ok = tuple(sz[i]) == (er, ec)
        if ok:
            got = o[i, :er, :ec]
            ok = bool((got == exp).all()) if interp == 0 else int(np.abs(got.astype(np.int16) - exp.astype(np.int16)).max()) <= 1
            ok = ok and bool((o[i, er:, :] == 7).all()) and bool((o[i, :, ec:] == 7).all())
        if not ok:
            bad.append((c, i, rows, cols, max_angle, step, group, interp, border, angle))
print("cases", cases, "mismatches", len(bad), bad[:5])
