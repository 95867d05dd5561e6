"""Randomised parity run over the other entry points (development aid, not part of the test suite):
arbitrary matrices, the projection drivers, gray / threshold / rotate, Canny, HoughLinesP, the edges
driver, the FFT pictures -- against the CPU oracle.  Usage: python tools/fuzz_all.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch  # noqa: F401

import oics
from oics import fft, hough, omr, projection, synth, transfer
from oics.types import RotateClipStrategy
from oracle import oracle as orc
from oracle import oracle_fft as offt

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
orc.build()
bad = []


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def img_u8(rows, cols, cn=1):
    kind = int(rng.integers(0, 3))
    if kind == 0:
        g, _ = synth.make_card(max(rows, 16), max(cols, 16), int(rng.integers(1, 10**6)))
        g = g[:rows, :cols]
    elif kind == 1:
        g = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    else:
        g = np.clip(rng.normal(128, 60, (rows, cols)), 0, 255).astype(np.uint8)
    if cn == 1:
        return np.ascontiguousarray(g)
    out = np.stack([np.roll(g, k, axis=k % 2) for k in range(cn)], axis=2)
    return np.ascontiguousarray(out)


for c in range(cases):
    rows, cols = int(rng.integers(3, 500)), int(rng.integers(3, 500))
    what = c % 8
    try:
        if what == 0:  # arbitrary matrices
            A = int(rng.integers(1, 12))
            b = np.where(rng.random((rows, cols)) < 0.2, 0, 255).astype(np.uint8)
            Ms = np.zeros((A, 6))
            for i in range(A):
                th, s = rng.uniform(-np.pi, np.pi), rng.uniform(0.3, 2.5)
                Ms[i] = [s * np.cos(th) + rng.uniform(-0.2, 0.2), s * np.sin(th), rng.uniform(-cols, cols),
                         -s * np.sin(th), s * np.cos(th) + rng.uniform(-0.2, 0.2), rng.uniform(-rows, rows)]
            vp, hp, vs, hs = projection.projection_sweep(b, Ms)
            evp, ehp, evs, ehs = orc.sweep_matrices(b, Ms)
            ok = (vp == evp).all() and (hp == ehp).all() and (bits64(vs) == bits64(evs)).all() and (bits64(hs) == bits64(ehs)).all()
        elif what == 1:  # projection drivers on colour input
            rows, cols = max(rows, 40), max(cols, 40)
            img = img_u8(rows, cols, 3)
            ma, st = int(rng.choice([5, 10, 45])), float(rng.choice([0.5, 0.2, 1.0]))
            sc = float(rng.choice([1.0, 0.5, 0.2]))
            got = projection.get_angle_with_projections(img, ma, st, sc, 1)
            exp = orc.get_angle_with_projections(img, ma, st, sc)
            r = omr.get_result_from_projection(img, ma, st, 248, 230) if min(rows, cols) >= 248 else None
            ok = got == exp[0]
            if r is not None:
                ea, es, ec = orc.get_result_from_projection(img, ma, st, 248, 230)
                ok = ok and r.angle == ea and int(r.status) == es and r.candidates.tolist() == ec.tolist()
        elif what == 2:  # gray, threshold, rotate
            img = img_u8(rows, cols, int(rng.choice([3, 4])))
            g = transfer.transfer_rgb_image_to_gray_image(img).get_mat()
            ok = (g == orc.rgb2gray(img)).all()
            t = transfer.transfer_gray_image_to_thresh_binary(g).get_mat()
            ok = ok and (t == orc.threshold_binary(g)).all()
            ang = float(rng.uniform(-60, 60))
            clip = int(rng.integers(0, 2))
            src = img if rng.random() < 0.5 else g
            r0 = transfer.rotate_mat(src, ang, 1.0, 0, 0, (255, 255, 255, 0), RotateClipStrategy(clip)).get_mat()
            ok = ok and (r0 == orc.rotate_mat(src, ang, 1.0, 0, (255, 255, 255, 0), clip)).all()
            r1 = transfer.rotate_mat(src, ang, 1.0, 1, 0, (255, 255, 255, 0), RotateClipStrategy(clip)).get_mat()
            e1 = orc.rotate_mat(src, ang, 1.0, 1, (255, 255, 255, 0), clip)
            ok = ok and r1.shape == e1.shape and np.abs(r1.astype(np.int16) - e1.astype(np.int16)).max() <= 1
        elif what == 3:  # Canny
            cn = int(rng.choice([1, 1, 3, 4]))
            img = img_u8(rows, cols, cn)
            lo, hi = float(rng.uniform(5, 120)), float(rng.uniform(60, 400))
            ok = (hough.canny(img, lo, hi) == orc.canny(img, lo, hi)).all()
        elif what == 4:  # HoughLinesP
            img = img_u8(rows, cols, 1)
            e = orc.canny(img, 80.0, 200.0)
            mll, mlg, thr = float(rng.uniform(3, 80)), float(rng.uniform(0, 30)), int(rng.choice([0, 0, 5, 30]))
            got = hough.hough_lines_p(e, 1.0, np.pi / 180.0, thr, mll, mlg)
            exp = orc.hough_lines_p(e, mll, mlg, threshold=thr)
            ok = got.shape == exp.shape and (got == exp).all()
        elif what == 5:  # edges driver
            rows, cols = max(rows, 30), max(cols, 30)
            img = img_u8(rows, cols, int(rng.choice([1, 3])))
            mll, mlg = float(rng.uniform(5, 60)), float(rng.uniform(1, 20))
            try:
                ea, es, ec, _ = orc.get_result_from_edges_detection(img, mll, mlg)
            except RuntimeError:
                try:
                    omr.get_result_from_edges_detection(img, mll, mlg)
                    ok = False
                except oics.OmrError as ex:
                    ok = ex.code == -215
            else:
                r = omr.get_result_from_edges_detection(img, mll, mlg)
                ok = (bits64(r.angle) == bits64(ea)).all() and int(r.status) == es and (bits64(r.candidates) == bits64(ec)).all()
        elif what == 6:  # FFT pictures (tolerance: 1 grey level)
            rows, cols = max(rows, 8), max(cols, 8)
            if rng.random() < 0.3:
                rows, cols = 1 << int(rng.integers(3, 10)), 1 << int(rng.integers(3, 10))
            img = img_u8(rows, cols, 1)
            m, lg = fft.get_fft_image(img)
            em, elg = offt.get_fft_image(img)
            d = np.abs(lg.astype(np.int16) - elg.astype(np.int16))
            ok = d.max() <= 1 and (d == 0).mean() > 0.97
        else:  # batch with launch groups
            n, grp = int(rng.integers(1, 7)), int(rng.choice([1, 2, 3, 4]))
            rows, cols = max(rows, 24), max(cols, 24)
            cards = np.stack([np.where(rng.random((rows, cols)) < 0.15, 0, 255).astype(np.uint8) for _ in range(n)])
            d = torch.from_numpy(cards).to("cuda:0")
            N, A = projection.candidate_count(10, 0.5)
            best = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
            vs = torch.zeros((n, A), dtype=torch.float64, device="cuda:0")
            hs = torch.zeros((n, A), dtype=torch.float64, device="cuda:0")
            b = projection.Batch(rows, cols, 10, 0.5, device=0, n_streams=int(rng.choice([1, 2])))
            b.set_group(grp)
            b.run_device(d.data_ptr(), rows * cols, cols, n, 0, best.data_ptr(), vs.data_ptr(), hs.data_ptr())
            b.sync()
            b.close()
            ok = True
            for i in range(n):
                _, _, evs, ehs = orc.sweep(cards[i], 10, 0.5)
                ok = ok and (bits64(vs[i].cpu().numpy()) == bits64(evs)).all() and (bits64(hs[i].cpu().numpy()) == bits64(ehs)).all()
                ok = ok and int(best[i]) == orc.argmax_path1(evs, ehs)[0]
    except Exception as ex:  # noqa: BLE001
        ok = False
        print("EXCEPTION case", c, what, rows, cols, repr(ex)[:200])
    if not ok:
        bad.append((c, what, rows, cols))
        print("MISMATCH case", c, "kind", what, rows, cols)
print("cases %d mismatches %d %s" % (cases, len(bad), bad[:10]))
