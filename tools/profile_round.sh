#!/bin/bash
# Round profiling recipe (run on the GPU box through gpurun).  Everything lands under gpurun_out/<tag>/;
# tools/make_profiles.py (run in the build container afterwards) turns it into profiles/<tag>_* stamped with the
# commit it measured.  Rules kept: counters in their own runs with --kernel-trace only; FETCH_SIZE and WRITE_SIZE
# each alone (TCC has 4 slots: 3 + 2); at most 8 SQ counters per pass; the program itself (python3 <script>) directly
# after `--`; nothing under the profiler spawns a build.
#   usage: bash tools/profile_round.sh r02
set -eo pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c "from oracle import oracle as o; o.build()" > "$OUT/oracle_build.log" 2>&1   # before anything is profiled
# 1. the bench line exactly as the driver runs it (its own FETCH_SIZE / WRITE_SIZE / SQ child passes included)
timeout -k 10 900 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
rm -rf "$OUT/bench_pmc" && cp -r gpurun_out/bench_pmc "$OUT/bench_pmc" 2>/dev/null || true
echo "[profile] bench line done"
# 2. kernel stats of the same command (no counters in this run)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline --no-pmc > "$OUT/stats.log" 2>&1
echo "[profile] kernel stats done"
# 3. second SQ pass of the sweep kernel (wait states), single-scan launches
timeout -k 10 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- python3 tools/kbench.py 3 > "$OUT/pmc_sq2.log" 2>&1
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -- python3 tools/kbench.py 3 > "$OUT/pmc_sq1.log" 2>&1
echo "[profile] sweep SQ counters done"
timeout -k 10 300 python3 tools/kbench.py 20 > "$OUT/kbench.log" 2>&1
timeout -k 10 300 python3 tools/kstamps.py > "$OUT/kstamps.log" 2>&1 || true
for d in 0 1 2 4 6 7; do OMR_RUNS_DBG=$d timeout -k 10 120 python3 tools/kdbg.py 6 2>/dev/null | grep sweep >> "$OUT/kdbg.log" || true; done
echo "[profile] sweep kernel timings done"
# 4. stage kernels, FFT, Hough (un-profiled numbers + kernel stats + split counter passes for the Hough stage)
timeout -k 10 300 python3 tools/bench_stages.py 30 > "$OUT/stages.log" 2>&1 || true
timeout -k 10 300 python3 tools/bench_fft.py 64 4 > "$OUT/fft.log" 2>&1 || true
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fft_stats" -- python3 tools/bench_fft.py > "$OUT/fft_stats.log" 2>&1 || true
timeout -k 10 600 python3 tools/bench_hough.py 256 8 2 > "$OUT/hough.log" 2>&1 || true
timeout -k 10 300 python3 tools/hough_run.py 1 1 5 > "$OUT/hough_single.log" 2>&1 || true
timeout -k 10 300 python3 tools/hstamps.py > "$OUT/hstamps_a4.log" 2>&1 || true
timeout -k 10 300 python3 tools/hstamps.py 1754 1240 > "$OUT/hstamps_half.log" 2>&1 || true
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hough_stats" -- python3 tools/hough_run.py 64 4 1 > "$OUT/hough_stats.log" 2>&1 || true
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/hough_fetch" -- python3 tools/hough_run.py 64 4 1 > "$OUT/hough_fetch.log" 2>&1 || true
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/hough_write" -- python3 tools/hough_run.py 64 4 1 > "$OUT/hough_write.log" 2>&1 || true
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/hough_sq" -- python3 tools/hough_run.py 64 4 1 > "$OUT/hough_sq.log" 2>&1 || true
echo "[profile] stages / fft / hough done"
timeout -k 10 300 python3 tools/bench_calls.py > "$OUT/calls.log" 2>&1 || true
timeout -k 10 600 python3 tools/core_protocol.py > "$OUT/core_protocol.log" 2>&1 || true
timeout -k 10 300 python3 tools/bench_host.py > "$OUT/host.log" 2>&1 || true
# 5. micro-benchmarks behind DESIGN.md's issue-cost / LDS statements
for t in valu_issue valu_ops lds_unaligned lds_bytes mem_latency; do
  hipcc -O2 --offload-arch=gfx950 tools/$t.hip -o /tmp/$t > "$OUT/$t.build.log" 2>&1 && timeout -k 10 200 /tmp/$t > "$OUT/$t.log" 2>&1 || true
done
echo "[profile] all done"
