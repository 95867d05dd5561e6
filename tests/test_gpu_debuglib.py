"""What can only be shown with the DEBUG library (lib/libomrdeskew_dbg.so, built by __graft_entry__.build() / `make
debug`): the kernels' guard flags surfacing as -217 at the production synchronisation points, and the multi-device
worker loop of omr_host_batch_run on logical devices mapped onto one GPU (round-4 verdict, items 6 and 7).  The checks
run in a process of their own (tests/debuglib_checks.py): a process binds one libomrdeskew, and this session's is the
release build -- which has neither hook (tests/test_abi.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_guard_flags_and_logical_devices_with_the_debug_library():
    dbg = os.path.join(ROOT, "omr-img-corrector_amd", "lib", "libomrdeskew_dbg.so")
    if not os.path.exists(dbg):  # a fresh checkout: build it in-tree, as __graft_entry__.build() does
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "omr-img-corrector_amd", "csrc"), "debug"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "debuglib_checks.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK guard flag") == 2 and r.stdout.count("OK n_devices") == 4
