"""The scan-lane sweep's programs on the CPU: the generator of the product library
(omr_slane_strip_program, host only -- no GPU involved) run through the oracle-side interpreter
(oracle/slane_interp.c: the register ring, late commits, (mask, pk) segments exactly as a wave executes them) must
reproduce the oracle's sweep (projection.rs:47-65) bit for bit: both integer projections of every candidate, for
several scans riding in different lanes.  This pins the program FORMAT and the generator before any GPU runs."""
import ctypes as C
import os

import numpy as np
import pytest

import oics
from oics import synth
from oracle import oracle as orc

u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)
def interleave(scans, gx, gy):
    """[lanes] binary u8 images -> entries[1 + (rows + 2 gy) * colsG][lanes] dwords (entry 0 and the guard zero)"""
    lanes = len(scans)
    rows, cols = scans[0].shape
    NW = (cols + 31) // 32
    colsG = NW + 2 * gx
    ent = np.zeros((1 + (rows + 2 * gy) * colsG, lanes), np.uint32)
    for ln, img in enumerate(scans):
        black = np.zeros((rows, NW * 32), np.uint8)
        black[:, :cols] = (img <= 127)
        words = np.packbits(black.reshape(rows, NW, 32), axis=2, bitorder="little").view(np.uint32).reshape(rows, NW)
        e = ent[1:, ln].reshape(rows + 2 * gy, colsG)
        e[gy:gy + rows, gx:gx + NW] = words
    return ent, NW, colsG


def grid(cols):
    """slane.hpp, slane_grid_offset / SlaneGeom::set: columns that do not exist in front of destination column 0, strips"""
    off = (32 - cols % 32) % 32
    if 1 <= off <= 3:
        off += 16
    return off, ((cols + off + 31) // 32 + 1) // 2


def strip_program(rows, cols, M, strip):
    L = oics.lib()
    rd, nrec, pre, most, gx, gy = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    Mc = np.ascontiguousarray(M, np.float64)
    rc = L.omr_slane_strip_program(rows, cols, Mc.ctypes.data_as(f64p), strip, None, None, C.byref(rd), C.byref(nrec),
                                   C.byref(pre), C.byref(most), C.byref(gx), C.byref(gy))
    if rc != 0:
        return None
    seg = np.zeros(rd.value * nrec.value, np.uint32)
    fet = np.zeros(4 * nrec.value, np.uint32)  # SL_FREC (slane.hpp, format v4)
    rc = L.omr_slane_strip_program(rows, cols, Mc.ctypes.data_as(f64p), strip, seg.ctypes.data_as(u32p),
                                   fet.ctypes.data_as(u32p), C.byref(rd), C.byref(nrec), C.byref(pre), C.byref(most), C.byref(gx),
                                   C.byref(gy))
    if rc != 0:
        return None
    return (seg, fet, gx.value, gy.value), rd.value, nrec.value, pre.value, most.value


def sweep_by_programs(scans, Ms):
    orc.build()
    OL = orc.lib()
    OL.orc_slane_run_strip.argtypes = [u32p, C.c_int, u32p, C.c_int, C.c_int, C.c_int, C.c_int, u32p, C.c_int64, C.c_int, u32p,
                                       u32p]
    OL.orc_slane_run_strip.restype = C.c_int
    lanes = len(scans)
    rows, cols = scans[0].shape
    off, NS = grid(cols)
    A = len(Ms)
    ents = {}  # the interleaved bit image per guard size (the single-matrix entry point sizes the guard per matrix)
    vp = np.zeros((lanes, A, cols), np.uint32)
    hp = np.zeros((lanes, A, rows), np.uint32)
    classes = {}
    for a in range(A):
        hrow = np.zeros((rows, lanes), np.uint32)
        for st in range(NS):
            got = strip_program(rows, cols, Ms[a], st)
            assert got is not None, "candidate %d strip %d does not fit: %s" % (a, st, oics.lib().omr_last_error())
            (seg, fet, gx, gy), rd, nrec, pre, most = got
            if (gx, gy) not in ents:
                ents[(gx, gy)] = interleave(scans, gx, gy)[0]
            ent = ents[(gx, gy)]
            classes[rd] = classes.get(rd, 0) + 1
            vcol = np.zeros((64, lanes), np.uint32)
            rc = OL.orc_slane_run_strip(seg.ctypes.data_as(u32p), rd, fet.ctypes.data_as(u32p), nrec, pre, rows, 2,
                                        ent.ctypes.data_as(u32p), ent.shape[0], lanes, hrow.ctypes.data_as(u32p),
                                        vcol.ctypes.data_as(u32p))
            assert rc == 0, "interpreter rejected the program of candidate %d strip %d" % (a, st)
            # (slane.hpp: destination word w covers columns 32 w - off ..)
            for b in range(64):
                col = st * 64 - off + b
                if 0 <= col < cols:
                    vp[:, a, col] = vcol[b]
                else:
                    assert (vcol[b] == 0).all()
        hp[:, a, :] = hrow.T
    return vp, hp, classes


@pytest.mark.parametrize("rows,cols,max_angle,step", [(120, 200, 10, 1.0), (97, 131, 5, 0.5), (64, 64, 9, 3.0),
                                                      (300, 70, 10, 2.5), (33, 450, 8, 2.0),
                                                      # widths = 31 / 30 / 29 modulo 32 at the sweep's edge: word 0 would
                                                      # hold 31 / 30 / 29 real bits and need a ninth slot for its white run
                                                      (210, 95, 10, 5.0), (150, 222, 10, 5.0), (90, 349, 10, 10.0)])
def test_programs_reproduce_the_oracle_sweep(rows, cols, max_angle, step):
    rng = np.random.Generator(np.random.PCG64(rows * 1000 + cols))
    scans = []
    for ln in range(5):
        dens = [0.5, 0.1, 0.9, 0.02, 0.3][ln]
        scans.append(np.where(rng.random((rows, cols)) < dens, 0, 255).astype(np.uint8))
    scans[3][:, :] = 0       # all black
    scans.append(np.full((rows, cols), 255, np.uint8))  # all white
    Ms = orc.rotation_matrices(rows, cols, max_angle, step)
    vp, hp, classes = sweep_by_programs(scans, Ms)
    for ln, img in enumerate(scans):
        evp, ehp, _, _ = orc.sweep_matrices(img, Ms)
        assert (vp[ln] == evp).all(), "vproj differs, lane %d" % ln
        assert (hp[ln] == ehp).all(), "hproj differs, lane %d" % ln
    assert sum(classes.values()) > 0


def test_record_classes_follow_the_angle():
    # small angles need few segments per word (16-dword records), the edge of the headline sweep more
    rows, cols = 400, 640
    for ang, want in ((0.0, 4), (0.4, 4), (3.0, 8), (10.0, 16)):
        M = orc.get_rotation_matrix_2d(cols / 2.0, rows / 2.0, ang, 1.0)
        _, rd, nrec, pre, most = strip_program(rows, cols, M, 3)
        assert rd == want, (ang, rd, most)
        assert nrec == (rows + pre + 15) // 16 * 16  # whole turns (slane_exec_records)


def test_steep_candidates_are_refused_not_mangled():
    # more than 8 segments per word (beyond about 10.5 degrees): the strip is reported as not fitting (-213) and the
    # engine leaves such candidates to the run-merging / gather kernels
    M = orc.get_rotation_matrix_2d(320.0, 200.0, 14.0, 1.0)
    assert strip_program(400, 640, M, 3) is None
    assert b"does not fit" in oics.lib().omr_last_error()


def test_headline_shape_sample_candidates():
    """A4 at the headline sweep's extremes and middle: a card in two lanes, three candidates, every strip."""
    rows, cols = 3508, 2480
    b0, _ = synth.make_binary_card(rows, cols, 2, skew=1.7)
    b1, _ = synth.make_binary_card(rows, cols, 9, skew=-4.2)
    Mall = orc.rotation_matrices(rows, cols, 10, 0.05)
    pick = [0, 137, 399]
    Ms = Mall[pick]
    vp, hp, classes = sweep_by_programs([b0, b1], Ms)
    for ln, img in enumerate((b0, b1)):
        evp, ehp, _, _ = orc.sweep_matrices(img, Ms, threads=os.cpu_count() or 4, fast=True)
        assert (vp[ln] == evp).all() and (hp[ln] == ehp).all()
