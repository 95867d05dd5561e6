"""rocprofv3 counter passes as child processes + CSV parsing (measurement plumbing for bench.py and
tools/; nothing here is on the data path).

Rules this follows (task statement / MI355X_MICROARCH.md):
  * counters are collected in their own runs, with --kernel-trace only (never with sys/hip/hsa traces);
  * FETCH_SIZE and WRITE_SIZE each get a pass of their own (TCC has 4 slots: 3 + 2 do not fit);
  * at most 8 SQ counters per pass;
  * the profiled program is `python3 <script> ...` itself, directly after `--`;
  * the child is started by a parent that has not touched the GPU yet.
"""
import csv
import glob
import os
import shutil
import statistics
import subprocess
import sys
from collections import defaultdict

# the units a sweep kernel can be bound by, one pass: vector issue, SCALAR issue (SALU + SMEM instructions against the CU's
# busy cycles: the scalar unit of a CU takes one instruction per cycle), the LDS array, and how long waves sit in s_waitcnt
SQ_PASS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
           "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_ANY"]
SQ_PASS2 = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"]


def rocprof():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


def _newest(root, pat):
    r = glob.glob(os.path.join(root, "**", pat), recursive=True)
    return max(r, key=os.path.getmtime) if r else None


def read_counters(root):
    """{kernel: {counter: [value per dispatch]}} -- rocprofv3 writes one row per (dispatch, counter)
    (several rows per dispatch when a counter has one instance per XCD/SE: those are summed)."""
    f = _newest(root, "*counter_collection.csv")
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    if f:
        for r in csv.DictReader(open(f)):
            key = (r.get("Dispatch_Id") or r.get("Correlation_Id"), r["Counter_Name"])
            per[key[0]][key[1]] += float(r["Counter_Value"])
            names[key[0]] = short(r["Kernel_Name"])
    out = defaultdict(lambda: defaultdict(list))
    for d, cs in per.items():
        for c, v in cs.items():
            out[names[d]][c].append(v)
    return out


def read_durations(root):
    """{kernel: [us per dispatch]} from the kernel trace of the same pass."""
    f = _newest(root, "*kernel_trace.csv")
    out = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def run_pass(counters, argv, out_dir, timeout_s=300, log=None):
    """One `rocprofv3 --pmc <counters> --kernel-trace -- python3 argv...` child.  Returns
    (counters dict, durations dict) or (None, reason)."""
    exe = rocprof()
    if exe is None:
        return None, "rocprofv3 not found"
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", out_dir, "--",
                                               sys.executable] + list(argv)
    try:
        p = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return None, "timeout after %d s" % timeout_s
    if log:
        with open(log, "wb") as f:
            f.write(p.stdout)
    if p.returncode != 0:
        return None, "rocprofv3 exit %d: %s" % (p.returncode, p.stdout[-300:].decode("utf-8", "replace"))
    c = read_counters(out_dir)
    if not c:
        return None, "no counter_collection.csv produced"
    return c, read_durations(out_dir)


def pick(kernels, needle):
    ks = [k for k in kernels if needle in k]
    return ks[0] if ks else None


def mean(xs):
    return statistics.fmean(xs) if xs else float("nan")


def hbm_bytes(fetch_kb, write_kb):
    """MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts
    128-B requests as 64 B for wide reads (x2); WRITE_SIZE is exact for 16-B-per-lane stores."""
    return (2.0 * fetch_kb + write_kb) * 1024.0
