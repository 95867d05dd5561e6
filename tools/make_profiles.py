"""Turn gpurun_out/<tag>/ (written on the GPU box by tools/profile_round.sh) into the committed summaries under
profiles/<tag>_*.  Every file names the commit it measured (gpurun_out/<tag>/COMMIT, written just before the
gpurun call).  Usage: python tools/make_profiles.py r03"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
from oics import pmc  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
commit = open(os.path.join(src, "COMMIT")).read().strip() if os.path.exists(os.path.join(src, "COMMIT")) else "unknown"
status = {}
import glob
_status_lines = []
for _f in sorted(glob.glob(os.path.join(src, "STATUS_*"))):
    _status_lines += open(_f).read().splitlines()
for ln in _status_lines:
    if " rc=" in ln:
        status[ln.split(" rc=")[0]] = int(ln.split(" rc=")[1])
failed = sorted(k for k, v in status.items() if v != 0)
STAMP = "Measured at commit `%s` on one MI355X (gpurun boxes), `bash tools/profile_round.sh %s a|b`.%s\n\n" % (
    commit, tag, ("  **Steps that did not end with exit code 0 (their sections below are missing or stale): %s.**" % ", ".join(failed))
    if failed else "  Every step of the recipe ended with exit code 0 (gpurun_out/%s/STATUS_a, STATUS_b)." % tag if status else "")


def read(name):
    p = os.path.join(src, name)
    return open(p).read() if os.path.exists(p) else ""


def last_json(text):
    for ln in reversed(text.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except Exception:  # noqa: BLE001
                pass
    return None


def stats_table(sub, f, top=12):
    ks = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not ks:
        f.write("(no kernel stats found)\n")
        return []
    rows = list(csv.DictReader(open(max(ks, key=os.path.getmtime))))
    f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:top]:
        f.write("| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %s |\n" % (
            pmc.short(r["Name"])[:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    return rows


def counter_table(sub, needle, f):
    c, d = pmc.read_counters(os.path.join(src, sub)), pmc.read_durations(os.path.join(src, sub))
    out = {}
    for k in sorted(c):
        if needle not in k:
            continue
        vals = {n: pmc.mean(v[1:] if len(v) > 2 else v) for n, v in c[k].items()}
        f.write("\n`%s` (%s; %d dispatches, mean duration %.1f us under the profiler)\n\n| counter | per launch |\n|---|---|\n" % (
            k, sub, len(next(iter(c[k].values()))), pmc.mean(d.get(k, []))))
        for n in sorted(vals):
            f.write("| %s | %.4g |\n" % (n, vals[n]))
        out[k] = (vals, pmc.mean(d.get(k, [])))
    return out


def hbm_table(fetch_sub, write_sub, f, needles=None, bytes_of=None):
    """Per kernel: FETCH_SIZE (its own pass) and WRITE_SIZE (its own pass), corrected as MI355X_MICROARCH.md's HBM
    section prescribes ((2 x FETCH_SIZE + WRITE_SIZE) x 1024 B), over the kernel's mean duration in those passes."""
    cf, df = pmc.read_counters(os.path.join(src, fetch_sub)), pmc.read_durations(os.path.join(src, fetch_sub))
    cw, dw = pmc.read_counters(os.path.join(src, write_sub)), pmc.read_durations(os.path.join(src, write_sub))
    f.write("| kernel | launches | mean us (profiled) | FETCH_SIZE KB | WRITE_SIZE KB | HBM MB per launch | HBM GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|---|\n")
    out = {}
    for k in sorted(cf):
        if needles and not any(n in k for n in needles):
            continue
        if k not in cw or "FETCH_SIZE" not in cf[k] or "WRITE_SIZE" not in cw[k]:
            continue
        fe, wr = pmc.mean(cf[k]["FETCH_SIZE"]), pmc.mean(cw[k]["WRITE_SIZE"])
        us = pmc.mean(df.get(k, []) + dw.get(k, []))
        b = pmc.hbm_bytes(fe, wr)
        f.write("| `%s` | %d | %.1f | %.0f | %.0f | %.2f | %.0f | %.3f |\n" % (k[:70], len(cf[k]["FETCH_SIZE"]), us, fe, wr, b / 1e6,
                                                                              b / us / 1e3, b / us / 1e3 / 8000))
        out[k] = (b, us)
    return out


# ---- bench line + kernel stats
line = last_json(read("bench_line.json"))
json.dump(line, open(os.path.join(dst, tag + "_bench_line.json"), "w"), indent=1)
with open(os.path.join(dst, tag + "_rocprofv3_kernel_stats_bench.md"), "w") as f:
    f.write("# %s -- `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-pmc`\n\n" % tag + STAMP)
    f.write("Default bench: %d scans per GPU per step (%d distinct cards), %d scans per kernel launch (sweep kernel: %s), 1 sweep "
            "stream + 1 post stream; the `value` leg (angle detection) is followed by the two deskew legs (`deskew_warp_kernel<true>` "
            "= LINEAR, `<false>` = NEAREST; they share the chip with the sweep of the next launch, so their durations here are "
            "longer than in `%s_deskew.md`); `slane_pack_kernel` / `slane_vproj_kernel` / `slane_stddev_kernel` are the scan-lane "
            "sweep's pack, column-count and std-dev stages; `runtab_kernel` / `runblk_kernel` / `rungeo_kernel` / `tables_kernel` and one "
            "`runs_kernel` launch are the creation of the run-merging plan every context still makes.  (The profiled command "
            "takes its cards from a file made beforehand and skips the host-memory leg: nothing under the profiler forks.)\n\n"
            % (line["config"]["scans_per_gpu_per_step"], line["config"]["distinct_cards_per_gpu"],
               line["config"]["scans_per_kernel_launch"], line["config"].get("sweep_kernel", "run-merging"), tag))
    rows = stats_table("stats", f)
    ks = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(max(ks, key=os.path.getmtime), os.path.join(dst, tag + "_rocprofv3_kernel_stats_bench.csv"))
    rl = line["roofline"]
    f.write("\nUn-profiled bench line of the same build (`profiles/%s_bench_line.json`): **%.0f images/s**, `roofline.kernel_ms` "
            "%.4f ms per launch of %d scans (HIP events on the launch stream, %d launches), bound `%s` frac %.3f; VALU issue %.3f, "
            "SCALAR issue %.3f, LDS busy %.3f (conflict share %.3f), HBM %.3f of 8 TB/s from the run's own FETCH_SIZE / WRITE_SIZE "
            "passes; waves in s_waitcnt %.3f of their cycles.\n" % (
                tag, line["value"], rl["kernel_ms"], rl["scans_per_launch"], rl["detail"]["launch_groups_timed"], rl.get("bound"),
                rl.get("frac") or float("nan"), rl.get("valu_frac") or float("nan"), rl.get("scalar_frac") or float("nan"),
                rl.get("lds_frac") or float("nan"), rl["detail"].get("lds_conflict_frac", float("nan")), rl.get("hbm_frac") or float("nan"),
                rl.get("wait_frac") or float("nan")))
    if line.get("deskew"):
        dk = line["deskew"]
        f.write("With the deskewed image produced inside the timed region (`omr_batch_deskew_device`): **%.0f images/s** LINEAR "
                "(%.3f of `value`), %.0f NEAREST (%.3f).\n" % (dk["linear_images_per_s"], dk["linear_over_value"],
                                                               dk["nearest_images_per_s"], dk["nearest_over_value"]))
    dom = "slane_kernel" if line["config"].get("sweep_kernel") == "scan-lane" else "runs_kernel"
    run = [r for r in rows if dom in r["Name"]]
    if run:
        f.write("The profiler's average for `omr::%s` (%.1f us) must agree with `roofline.kernel_ms` (%.1f us) up to the "
                "profiler's clock effect.\n" % (dom, float(run[0]["AverageNs"]) / 1e3, rl["kernel_ms"] * 1e3))
    if line.get("e2e_host"):
        f.write("Host-memory end to end (`e2e_host`): %s\n" % json.dumps(line["e2e_host"]))

# ---- sweep counters
with open(os.path.join(dst, tag + "_pmc_sweep.md"), "w") as f:
    f.write("# %s -- counters of the sweep kernel (C2: 2480x3508, A = 400)\n\n" % tag + STAMP)
    dom = "slane_kernel" if line and line["config"].get("sweep_kernel") == "scan-lane" else "runs_kernel"
    f.write("## HBM-side traffic and SQ counters measured INSIDE the bench run (bench.py's child passes, %d scans per launch, `%s`)\n"
            % (line["config"]["scans_per_kernel_launch"] if line else 0, dom))
    for sub in ("bench_pmc/fetch", "bench_pmc/write", "bench_pmc/sq"):
        counter_table(sub, dom, f)
    if line and line["roofline"].get("traffic"):
        rl = line["roofline"]
        f.write("\n(2 x FETCH_SIZE + WRITE_SIZE) x 1024 = **%.0f MB per launch of %d scans** = %.0f GB/s = %.3f of the 8 TB/s HBM peak, "
                "against %.0f MB of compulsory traffic (programs once per quad of scan groups + bit images + outputs) and %.0f MB of SURVEY-8(d) algorithmic bytes.\n" % (
                    rl["traffic"] / 1e6, rl["scans_per_launch"], rl["detail"]["hbm_measured_GBps"], rl["hbm_frac"],
                    rl["compulsory_bytes_per_launch"] / 1e6, rl["detail"]["algorithmic_bytes_per_launch"] / 1e6))
        c = rl["detail"].get("sq_counters_per_launch")
        if c:
            f.write("\n`roofline` of the bench line, reproduced from the SQ pass above: SCALAR = (SQ_INSTS_SALU + SQ_INSTS_SMEM) / kernel time / "
                    "(256 CUs x 2.4 GHz) = **%.3f** (against the CU's own busy cycles: %.3f); VALU = SQ_INSTS_VALU / kernel time / (1024 SIMDs x "
                    "2.4 GHz / 2) = **%.3f**; LDS = SQ_LDS_IDX_ACTIVE / kernel time / (256 x 2.4 GHz) = %.3f; waves waiting in s_waitcnt = "
                    "SQ_WAIT_ANY / SQ_WAVE_CYCLES = **%.3f**; `bound` = `%s` (the largest fraction), `frac` = %.3f.\n" % (
                        rl.get("scalar_frac") or float("nan"), rl["detail"].get("scalar_frac_of_busy_cu_cycles", float("nan")),
                        rl.get("valu_frac") or float("nan"), rl.get("lds_frac") or float("nan"), rl.get("wait_frac") or float("nan"),
                        rl.get("bound"), rl.get("frac") or float("nan")))
        json.dump({"kernel": rl["kernel"], "measured_at": "commit %s, %s" % (commit, tag),
                   "FETCH_SIZE_KB_per_launch": rl["detail"]["hbm_counters_KB"]["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": rl["detail"]["hbm_counters_KB"]["WRITE_SIZE"],
                   "sweep_kernel_hbm_bytes_per_launch": rl["traffic"], "scans_per_launch": rl["scans_per_launch"],
                   "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md HBM section), WRITE_SIZE as is, x1024 B"},
                  open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    if os.path.exists(os.path.join(src, "pmc_lanes", "summary.md")):
        f.write("\n## The scan-lane kernel, one launch of 512 scans (`bash tools/pmc_lanes.sh`: six passes, mean of the launches)\n\n")
        f.write(open(os.path.join(src, "pmc_lanes", "summary.md")).read())
        f.write("\n`python3 tools/klanes.py 512 512 3` (un-profiled, HIP events around the sweep kernel):\n\n```\n%s```\n"
                % read("klanes.log").replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", ""))
    f.write("\n## The run-merging kernel in batch mode (tools/kbatch.py: 8 scans per launch), two SQ passes\n")
    r1 = {}
    for sub in ("pmc_sq1", "pmc_sq2"):
        r1.update(counter_table(sub, "runs_kernel", f))
    c1 = pmc.read_counters(os.path.join(src, "pmc_sq1"))
    d1 = pmc.read_durations(os.path.join(src, "pmc_sq1"))
    k = pmc.pick(c1.keys(), "runs_kernel")
    if k and "SQ_BUSY_CU_CYCLES" in c1[k]:
        cyc = pmc.mean(c1[k]["SQ_BUSY_CU_CYCLES"][1:]) / 256.0
        us = pmc.mean(d1[k][1:])
        nv = pmc.mean(c1[k]["SQ_INSTS_VALU"][1:])
        f.write("\nClock held in the kernel: SQ_BUSY_CU_CYCLES / 256 CUs / duration = %.0f cycles / %.1f us = **%.2f GHz**.  VALU: %.3g "
                "wave-instructions per launch = %.0f per SIMD; at the 3.86 cycles per instruction of this kernel's mix "
                "(profiles/r02_valu_issue.md) that is %.2f of the held clock's cycles, at 2 cycles (plain VOP2) %.2f.\n"
                % (cyc, us, cyc / us / 1e3, nv, nv / 1024, nv / 1024 * 3.86 / cyc, nv / 1024 * 2 / cyc))
    f.write("\n## Un-profiled kernel timings\n\n`python3 tools/kbatch.py 16 8` (8 scans per launch, HIP events around the sweep stage):\n\n```\n%s```\n"
            % read("kbatch.log").replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", ""))
    if read("kbatch_groups.log"):
        f.write("\nLarger launch groups (`python3 tools/kbatch.py 8 16` / `8 32`; the bench's default is 32 scans per launch: the "
                "partial last wave of workgroups of a launch is amortised over more work):\n\n```\n%s```\n" % read("kbatch_groups.log"))
    f.write("\n`python3 tools/kbench.py 20` (one scan per launch, the three sweep kernels):\n\n```\n%s```\n"
            % read("kbench.log").replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", ""))
    f.write("\n## Phase clocks of wave 0 (debug library, tools/kstamps.py)\n\n```\n%s```\n" % read("kstamps.log").replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", ""))

# ---- the batch warp
with open(os.path.join(dst, tag + "_deskew.md"), "w") as f:
    f.write("# %s -- the batch's last stage: `deskew_warp_kernel` (CONTAIN warp of 8 A4 scans per launch by their detected angles)\n\n" % tag + STAMP)
    f.write("`python3 tools/bench_deskew.py 10` (the batch context is synchronised after every call, so the warp has the chip to itself):\n\n```\n%s```\n\n"
            % "\n".join(l for l in read("deskew.log").splitlines() if "amdgpu.ids" not in l))
    f.write("Kernel stats of the same script:\n\n")
    stats_table("deskew_stats", f, 8)
    f.write("\nHBM-side traffic (FETCH_SIZE and WRITE_SIZE each in a pass of its own) against the algorithmic bytes (scan in once + canvas out once):\n\n")
    hbm_table("deskew_fetch", "deskew_write", f, needles=("deskew_warp_kernel",))
    if read("deskew64.log"):
        f.write("\n## 64 scans per launch (`python3 tools/bench_deskew.py 6 64`: the per-tile records' kernel and the launch overhead amortised)\n\n```\n%s```\n\n"
                % "\n".join(l for l in read("deskew64.log").splitlines() if "amdgpu.ids" not in l))
        stats_table("deskew64_stats", f, 8)

# ---- where and when the workgroups of a scan-lane launch ran
if read("kstamps_lanes.log"):
    with open(os.path.join(dst, tag + "_lanes_stamps.md"), "w") as f:
        f.write("# %s -- one scan-lane launch of 512 A4 scans, workgroup by workgroup (tools/kstamps_lanes.py, library built with -DSLANE_STAMP)\n\n" % tag + STAMP)
        f.write("Start, end (constant 100 MHz clock) and XCC id of every workgroup: the units are dealt to the XCDs by `slane_deal_units` "
                "(DESIGN.md section 4.6), so every XCD runs 1 000 of the 8 000 workgroups and all end within 0.1 ms of each other; what is "
                "left between a CU's busy time and the launch's span is the tail of the last workgroups.\n\n```\n%s```\n"
                % "\n".join(l for l in read("kstamps_lanes.log").splitlines() if "amdgpu.ids" not in l))

# ---- micro-benchmark behind DESIGN.md's LDS-DMA statements
if read("lds_dma_window.log"):
    with open(os.path.join(dst, tag + "_lds_dma.md"), "w") as f:
        f.write("# %s -- LDS-DMA probe behind DESIGN.md section 4.1 (tools/lds_dma_window.hip)\n\n" % tag + STAMP)
        f.write("```\n%s```\n" % read("lds_dma_window.log"))
        if read("lds_u16_probe.log"):
            f.write("\n`tools/lds_u16_probe.hip` (DESIGN.md section 4.3: the bilinear taps of the batch warp): 16-bit LDS reads at byte "
                    "addresses -- do they work, what do they cost next to byte reads? -- and `v_dot4_u32_u8`:\n\n```\n%s```\n" % read("lds_u16_probe.log"))

if read("mov64_probe.log") or read("pf_probe.log"):
    with open(os.path.join(dst, tag + "_slane_microbench.md"), "w") as f:
        f.write("# %s -- probes behind two statements of DESIGN.md section 4.6\n\n" % tag + STAMP)
        f.write("`tools/mov64_probe.hip`: does `v_mov_b64` take a DST_REL index of the VGPR index mode?  (The ring commits of program format v3 "
                "move a pair of landing registers with one instruction.)\n\n```\n%s```\n\n" % read("mov64_probe.log"))
        f.write("`tools/pf_probe.hip`: `global_load_dword vdst, voffset, s[base:base+1]` under a partial EXEC and the index mode, step by step "
                "(the first prefetch attempt of round 5 used this form and faulted inside the sweep kernel; in isolation every step passes -- the "
                "kernel's prefetch now goes through a buffer descriptor whose range check drops every lane it is not meant to load).\n\n```\n%s```\n"
                % read("pf_probe.log"))

# ---- stages / fft / hough / calls
with open(os.path.join(dst, tag + "_stages.md"), "w") as f:
    f.write("# %s -- stage kernels either side of the sweep (SURVEY.md 8f rows 1-2)\n\n" % tag + STAMP)
    f.write("`python3 tools/bench_stages.py 30`: device-resident 2480x3508 scans, HIP events around single launches, median of 30; "
            "GB/s = compulsory bytes (input once + output once) / time.  The `[x9 tall]` rows run the same kernels on nine scans "
            "stacked into one image, i.e. with the launch ramp and tail amortised.\n\n```\n%s```\n" % "\n".join(
                l for l in read("stages.log").splitlines() if not l.startswith("{") and "amdgpu.ids" not in l))
    f.write("\nKernel stats of the same script (`rocprofv3 --kernel-trace --stats`):\n\n")
    stats_table("stages_stats", f, 14)
    f.write("\nMeasured HBM-side traffic per kernel (FETCH_SIZE and WRITE_SIZE each in a pass of its own; mean over ALL launches of a "
            "kernel in the script, single scans and the nine-scan stacks alike):\n\n")
    hbm_table("stages_fetch", "stages_write", f)
with open(os.path.join(dst, tag + "_fft.md"), "w") as f:
    f.write("# %s -- FFT path (BASELINE config 5)\n\n" % tag + STAMP)
    j = last_json(read("fft.log"))
    if j:
        f.write("`python3 tools/bench_fft.py 64 4` (mean over the repetitions; algorithmic = SURVEY.md Appendix C's count, 33 B/px: the u8 scan in + four passes over the complex f32 spectrum; "
                "kernel = what the kernels move: u8 in, half spectrum and |F| of it written and read once, two pictures out):\n\n")
        f.write("| case | scans/s | ms/scan (mean) | best | algorithmic GB/s | of 8 TB/s | kernel-bytes GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|---|\n")
        for k, v in j.items():
            f.write("| %s (batch %d) | %.0f | %.3f | %.3f | %.0f | %.3f | %.0f | %.3f |\n" % (
                k, v["batch"], v["scans_per_s"], v["ms_per_scan"], v["best_ms_per_scan"], v["algorithmic_GBps"],
                v["algorithmic_GBps"] / 8000, v["kernel_GBps"], v["kernel_GBps"] / 8000))
    f.write("\nKernel stats (`rocprofv3 --kernel-trace --stats -- python3 tools/bench_fft.py 32 3`, both sizes):\n\n")
    stats_table("fft_stats", f, 8)
    j3 = last_json(read("fft_a3.log"))
    if j3:
        f.write("\nAxes beyond 8192 points (`csrc/fft_big.hip`: chirp-z through global memory), `python3 tools/bench_fft.py 2 3 a3`:\n\n")
        for k, v in j3.items():
            f.write("* %s: **%.2f ms per scan** (%.1f scans/s)\n" % (k, v["ms_per_scan"], v["scans_per_s"]))
        f.write("\nKernel stats of the same script:\n\n")
        stats_table("fft_a3_stats", f, 10)
    for w, nm in (("c5", "4096 x 4096"), ("a4", "2480 x 3508")):
        f.write("\nMeasured HBM-side traffic, %s, 16 scans per call = two launches of 8 scans, warm-up included and of the same size (FETCH_SIZE and WRITE_SIZE each in a pass of its own):\n\n" % nm)
        hbm_table("fft_fetch_" + w, "fft_write_" + w, f, needles=("fft_pass_kernel", "fft_mixed_kernel", "fft_bluesub_kernel", "spec_pictures_kernel"))
        for needle in ("fft_pass_kernel", "fft_mixed_kernel", "fft_bluesub_kernel"):
            counter_table("fft_sq_" + w, needle, f)
with open(os.path.join(dst, tag + "_hough.md"), "w") as f:
    f.write("# %s -- Hough-line path (BASELINE config 4)\n\n" % tag + STAMP)
    j = last_json(read("hough.log"))
    if j:
        f.write("`python3 tools/bench_hough.py 256 8 2` (mean over the repetitions):\n\n```\n%s\n```\n\n" % json.dumps(j, indent=1))
    j1 = last_json(read("hough_single.log"))
    if j1:
        f.write("One resident A4 scan, Canny + HoughLinesP + vote, best of 5 (`python3 tools/hough_run.py 1 1 5`): **%.4f s**\n\n" % j1["seconds"])
    ml = "\n".join(l for l in read("mem_latency.log").splitlines() if "amdgpu.ids" not in l)
    if ml:
        f.write("What one dependent memory round trip costs a lone wave (`tools/mem_latency.hip`, shader cycles):\n\n```\n%s\n```\n\n" % ml)
    f.write("Phase clocks of one scan's sequential stage (debug library, tools/hstamps.py):\n\n```\n%s```\n\n" % (
        read("hstamps_a4.log").replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", "")))
    f.write("Kernel stats, 64 resident scans (`rocprofv3 --kernel-trace --stats -- python3 tools/hough_run.py 64 4 1`):\n\n")
    stats_table("hough_stats", f, 8)
    f.write("\nCounters of `ppht_kernel` (separate passes: FETCH_SIZE alone, WRITE_SIZE alone, eight SQ counters):\n")
    res = {}
    for sub in ("hough_fetch", "hough_write", "hough_sq"):
        res.update({sub: counter_table(sub, "ppht_kernel", f)})
    try:
        fe = list(res["hough_fetch"].values())[0]
        wr = list(res["hough_write"].values())[0]
        b = pmc.hbm_bytes(fe[0]["FETCH_SIZE"], wr[0]["WRITE_SIZE"])
        f.write("\nHBM-side traffic of the 64-scan `ppht_kernel` launch: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.2f GB in %.1f ms "
                "= %.0f GB/s = %.3f of 8 TB/s.\n" % (b / 1e9, fe[1] / 1e3, b / (fe[1] * 1e-6) / 1e9, b / (fe[1] * 1e-6) / 1e9 / 8000))
    except Exception as e:  # noqa: BLE001
        f.write("\n(HBM traffic not computed: %r)\n" % (e,))
    ab = "\n".join(l for l in read("hough_ab.log").splitlines() if "amdgpu.ids" not in l and not l.startswith("{"))
    if ab.strip():
        f.write("\n## Accumulator width and scans in flight (`bash tools/hough_ab.sh`, ONE box, both libraries built from this commit)\n\n"
                "`omr_hough_set_scans_in_flight(k)`: k workgroups, each takes the next scan of the batch when it has finished one "
                "(k = batch: one workgroup per scan, the default).  Rates are whole calls of `omr_edges_detection_batch_device` "
                "(Canny, hysteresis, point lists, HoughLinesP, votes), mean over the repetitions.\n\n```\n%s\n```\n" % ab)
with open(os.path.join(dst, tag + "_calls.md"), "w") as f:
    f.write("# %s -- per-call latency of the host-image drivers and the host-memory batch\n\n" % tag + STAMP)
    f.write("`python3 tools/bench_calls.py`:\n\n```\n%s```\n\n`python3 tools/bench_host.py`:\n\n```\n%s```\n\n`python3 tools/bench_threads.py`:\n\n```\n%s```\n" % (
        "\n".join(l for l in read("calls.log").splitlines() if "amdgpu.ids" not in l) + "\n",
        "\n".join(l for l in read("host.log").splitlines() if "amdgpu.ids" not in l) + "\n",
        "\n".join(l for l in read("threads.log").splitlines() if "amdgpu.ids" not in l) + "\n"))
    if read("ksmall.log"):
        f.write("\nThe app's default sweep as a batch (+-45 deg @ 0.2 deg = 450 candidates on 248 x 230 working images, getLibParams.ts:30-60), "
                "`python3 tools/ksmall.py 4096 64`:\n\n```\n%s\n```\n" % "\n".join(l for l in read("ksmall.log").splitlines() if "amdgpu.ids" not in l))
    if read("klanes600.log"):
        f.write("\nA batch of 600-dpi A4 scans (4960 x 7016) through the scan-lane sweep, `python3 tools/klanes.py 128 64 2 7016 4960`:\n\n```\n%s\n```\n"
                % "\n".join(l for l in read("klanes600.log").splitlines() if "amdgpu.ids" not in l))
cp = "\n".join(l for l in read("core_protocol.log").splitlines() if "amdgpu.ids" not in l)
if cp.strip():
    with open(os.path.join(dst, tag + "_core_protocol.md"), "w") as f:
        f.write("# %s -- the reference's comparative benchmark (packages/core/src/main.rs:17-252) through the drop-in API\n\n" % tag + STAMP)
        f.write("104 dataset sheets, each skewed by a seeded random angle in [-10, 10) (CONTAIN, INTER_LINEAR, white), then "
                "`get_angle_with_projections(45, 0.2, 0.2, 1)`, `get_angle_with_hough(125, 15)`, `get_angle_with_fft(125, 150, 150, 75)` "
                "and the rotation back; run time per sheet (host image in, host image out, no JPEG write) and |estimate - injected|. "
                "The reference publishes no numbers for this protocol; its `main()` runs the first two methods.\n\n"
                "`python3 tools/core_protocol.py`:\n\n```\n%s\n```\n" % cp)
print("profiles written for", tag, "commit", commit)
