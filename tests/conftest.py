import os
import sys

import pytest
import torch  # noqa: F401  (first: libomrdeskew.so then shares torch's libamdhip64.so.7 instead of loading a second HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "omr-img-corrector_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # a fresh checkout has no built library (it is git-ignored): compile it in-tree, exactly what
    # __graft_entry__.build() does.  This is a build step, not a fallback -- a failed build fails the run.
    import subprocess
    if not os.path.exists(os.path.join(PKG, "lib", "libomrdeskew.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc")])


def _has_gpu():
    try:
        import oics
        return oics.lib().omr_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not pass on a fallback: only skip gpu tests
    # when they were not asked for explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
