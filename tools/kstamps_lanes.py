"""Where and when the workgroups of ONE scan-lane launch ran (development aid; needs a library built with
`tools/build_variant.sh stamp -DSLANE_STAMP`): per XCD the busy time of its CUs and the moment its last workgroup ended, so that
a static imbalance of the launch order (slane.hip: ids go round-robin to the 8 XCDs) shows as XCDs that finish early.
Usage: OMR_AB_LIB=omr-img-corrector_amd/lib/variants/libomrdeskew_stamp.so python3 tools/ab_lib.py tools/kstamps_lanes.py [scans]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
import numpy as np
import torch

from oics import _lib, projection, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ROWS, COLS = 3508, 2480
cards = [synth.make_card(ROWS, COLS, 3 + i) for i in range(8)]
dev = torch.device("cuda:0")
buf = torch.empty((n, ROWS, COLS), dtype=torch.uint8, device=dev)
for i in range(n):
    buf[i] = torch.from_numpy(cards[i % 8][0]).to(dev)
best = torch.zeros(n, dtype=torch.int32, device=dev)
b = projection.Batch(ROWS, COLS, 10, 0.05, n_streams=1)
b.set_lanes(n)
b.set_timing(True)
for it in range(2):
    b.run_device(buf.data_ptr(), ROWS * COLS, COLS, n, 127, best.data_ptr())
    b.sync()
ms, k = b.kernel_ms()
print("sweep kernel %.3f ms" % (ms / k))
L = C.CDLL(_lib.LIB_PATH)
N = 16384
st = np.zeros(3 * N, dtype=np.uint64)
rc = L.omr_debug_slane_stamps(st.ctypes.data_as(C.c_void_p), C.c_int(3 * N))
assert rc == 0, rc
st = st.reshape(N, 3)
ran = st[:, 1] > 0
ids = np.nonzero(ran)[0]
t0 = st[ran, 0].min()
start = (st[ran, 0] - t0).astype(np.float64) / 100.0  # us (100 MHz)
end = (st[ran, 1] - t0).astype(np.float64) / 100.0
xcc = (st[ran, 2] >> np.uint64(32)).astype(np.int64) & 15
hw = (st[ran, 2] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
print("%d workgroups ran; launch spans %.1f us" % (len(ids), end.max()))
print("id %% 8 == XCC id for %d of %d" % (int(((ids & 7) == xcc).sum()), len(ids)))
for x in range(8):
    m = xcc == x
    if not m.any():
        continue
    busy = (end[m] - start[m]).sum()
    ncu = len(set(zip(se[m].tolist(), cu[m].tolist())))
    print("XCC %d: %4d workgroups on %2d CUs, busy %.1f us per CU, first start %.1f, last start %.1f, last end %.1f us, longest %.1f shortest %.1f us"
          % (x, int(m.sum()), ncu, busy / max(ncu, 1), start[m].min(), start[m].max(), end[m].max(), (end[m] - start[m]).max(),
             (end[m] - start[m]).min()))
d = end - start
order = np.argsort(ids)
# durations along the launch order: mean of consecutive blocks of 256 ids
print("mean duration (us) per 256 consecutive ids:", " ".join("%.0f" % d[order][i:i + 256].mean() for i in range(0, len(ids), 256)))
b.close()
