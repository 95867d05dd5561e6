"""Checks that need the DEBUG library (make debug, -DOMR_RUNS_DEBUG; lib/libomrdeskew_dbg.so) -- run as a process of its
own by tests/test_gpu_debuglib.py, because a process binds ONE libomrdeskew and the test session's is the release build.

  1. guard flags (round-4 verdict, item 6): omr_debug_poke_guard sets a scratch set's flag on the device as a kernel that
     could not sweep would; the next omr_batch_sync must return OMR_ERR_GPU (-217) and the one after that OMR_OK again --
     for a scan-lane context, a run-merging context, and through omr_host_batch_run's own synchronisation point;
  2. the multi-device worker loop of omr_host_batch_run (item 7): omr_debug_set_logical_devices(k) maps k logical devices
     onto device 0; n_devices = 2 and 3 (interleaved scans, one worker thread / context / copy stream / pinned ring per
     device, results gathered in order) must give the bits of n_devices = 1, and the oracle's on a sample.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch  # noqa: F401  (first: the library then shares torch's HIP runtime)

from oics import _lib as _l

_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "libomrdeskew_dbg.so")  # before the first lib() call
import oics
from oics import projection, synth
from oracle import oracle as orc

L = C.CDLL(_l.LIB_PATH)
L.omr_debug_poke_guard.argtypes = [C.c_void_p, C.c_int]
L.omr_debug_set_logical_devices.argtypes = [C.c_int]
orc.build()
assert oics.lib().omr_device_count() >= 1


def scans_of(rows, cols, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n):
        if i % 3 == 0:
            img = np.where(rng.random((rows, cols)) < rng.uniform(0.03, 0.5), 0, 255).astype(np.uint8)
        else:
            img = synth.make_binary_card(rows, cols, 50 * seed + i, skew=float(rng.uniform(-4, 4)))[0]
        out.append(np.ascontiguousarray(img))
    return out


# ---- 1. guard flags
rows, cols, n = 120, 200, 70
sc = scans_of(rows, cols, n, 3)
buf = torch.from_numpy(np.stack(sc)).to("cuda:0")
best = torch.zeros(n, dtype=torch.int32, device="cuda:0")
for mode in ("scan-lane", "run-merging"):
    b = projection.Batch(rows, cols, 5, 0.5, n_streams=1)
    if mode == "scan-lane":
        b.set_lanes(128)
    else:
        b.set_group(8)
    b.run_device(buf.data_ptr(), rows * cols, cols, n, 127, best.data_ptr())
    b.sync()  # a healthy launch: no flag
    ref = best.cpu().numpy().copy()
    assert L.omr_debug_poke_guard(b.handle, 0) == 0
    try:
        b.sync()
        raise SystemExit("FAIL: %s: a set guard flag did not fail omr_batch_sync" % mode)
    except oics.OmrError as e:
        assert e.code == -217, e.code
        assert ("slane_kernel" if mode == "scan-lane" else "runs_kernel") in e.message, e.message
    b.sync()  # the flag was cleared with the report
    b.run_device(buf.data_ptr(), rows * cols, cols, n, 127, best.data_ptr())
    b.sync()
    assert (best.cpu().numpy() == ref).all()
    b.close()
    print("OK guard flag -> -217 at omr_batch_sync, then clean again (%s context)" % mode)

# ---- 2. logical devices
rows, cols = 150, 220
n = 300
sc = scans_of(rows, cols, n, 9)
N, A = orc.candidate_count(6, 0.5)
hb1 = projection.HostBatch(rows, cols, 6, 0.5, n, n_devices=1)
b1, a1, v1, h1 = hb1.run(sc, want_sd=True)
hb1.close()
for i in range(0, n, 29):
    _, _, evs, ehs = orc.sweep(sc[i], 6, 0.5)
    assert (v1[i].view(np.uint64) == evs.view(np.uint64)).all() and (h1[i].view(np.uint64) == ehs.view(np.uint64)).all()
    assert b1[i] == orc.argmax_path1(evs, ehs)[0]
try:
    projection.HostBatch(rows, cols, 6, 0.5, n, n_devices=2)
    if oics.lib().omr_device_count() < 2:
        raise SystemExit("FAIL: two devices accepted on a one-GPU box without logical devices")
except oics.OmrError as e:
    assert e.code == -5
for k, m in ((2, n), (3, n), (4, 150), (2, 131)):
    assert L.omr_debug_set_logical_devices(k) == 0
    hb = projection.HostBatch(rows, cols, 6, 0.5, m, n_devices=k)
    nd, spl, lane = hb.info()
    assert nd == k, (nd, k)
    for rep in range(2):  # twice on one context: rings, stages and events are reused
        bk, ak, vk, hk = hb.run(sc[:m], want_sd=True, packed=(rep == 1))  # the second run: packed transfers
        assert (bk == b1[:m]).all() and (ak == a1[:m]).all(), "best / angle differ with %d logical devices" % k
        assert (vk.view(np.uint64) == v1[:m].view(np.uint64)).all() and (hk.view(np.uint64) == h1[:m].view(np.uint64)).all()
    hb.close()
    print("OK n_devices = %d (logical, on device 0): %d scans, %s path per device, results == n_devices = 1 bit for bit, in order"
          % (k, m, "scan-lane" if lane else "run-merging"))
assert L.omr_debug_set_logical_devices(0) == 0
# omr_sweep_batch (context made and released inside the call) with two logical devices
assert L.omr_debug_set_logical_devices(2) == 0
bs, as_, vs, hs = projection.sweep_batch(sc[:200], 6, 0.5, n_devices=2, want_sd=True)
assert (bs == b1[:200]).all() and (vs.view(np.uint64) == v1[:200].view(np.uint64)).all()
print("OK omr_sweep_batch with 2 logical devices == n_devices = 1")
assert L.omr_debug_set_logical_devices(0) == 0
print("ALL OK")
