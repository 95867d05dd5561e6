"""The run-merging sweep kernel must keep 4 waves per SIMD (128 VGPRs; two 512-thread workgroups per CU):
an innocent edit can push the allocator into scratch spills.  This compiles runs.hip with resource remarks
and pins what the measurements rely on: no scratch, no VGPR spill, occupancy 4."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "omr-img-corrector_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_runs_kernel_has_no_spills(tmp_path):
    # the flags runs.hip is built with: the Makefile's common ones plus its RUNS_FLAGS (scheduler strategy)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    extra = re.search(r"^RUNS_FLAGS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
           "-fno-fast-math", *extra, "-I", CSRC, "-c", os.path.join(CSRC, "runs.hip"), "-Rpass-analysis=kernel-resource-usage",
           "-o", str(tmp_path / "runs.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = out.split("Function Name: ")
    mine = [b for b in blocks if b.startswith("_ZN3omr11runs_kernelE")]
    assert mine, "runs_kernel not found in the resource remarks"
    txt = mine[0]

    def field(name):
        m = re.search(re.escape(name) + r": (\d+)", txt)
        assert m, name
        return int(m.group(1))

    assert field("ScratchSize [bytes/lane]") == 0
    assert field("VGPRs Spill") == 0
    assert field("VGPRs") <= 128
    assert field("Occupancy [waves/SIMD]") == 4
