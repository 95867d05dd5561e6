#!/bin/bash
# Round profiling recipe (run on the GPU box through gpurun, in two calls: a gpurun call lasts 20 minutes at most).
# Everything lands under gpurun_out/<tag>/; tools/make_profiles.py (run in the build container afterwards) turns it
# into profiles/<tag>_* stamped with the commit it measured.  Rules kept: counters in their own runs with
# --kernel-trace only; FETCH_SIZE and WRITE_SIZE each alone (TCC has 4 slots: 3 + 2); at most 8 SQ counters per pass;
# the program itself (python3 <script>) directly after `--`; nothing under the profiler spawns a build or a child.
#   usage: bash tools/profile_round.sh r03 a     (bench line, sweep kernel, batch warp)
#          bash tools/profile_round.sh r03 b     (stage kernels, FFT, Hough, per-call latencies, micro-benchmarks)
#          bash tools/profile_round.sh r03 c     (Hough batch: int32 against 16-bit accumulator, scans in flight)
set -o pipefail
TAG=${1:-r03}
PART=${2:-a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# Every step removes what an earlier run left under its name and records its exit code in $OUT/STATUS_<part> (one file per gpurun call: each call starts on a fresh box and its files are merged back over the earlier ones; round-3 advice:
# a step that fails or times out must not leave the previous run's files to be stamped with the new commit;
# tools/make_profiles.py marks steps whose code is not 0).
step() { # step <name> <command...>: un-profiled measurement, output to $OUT/<name>.log
  local n=$1; shift
  rm -f "$OUT/$n.log"
  "$@" > "$OUT/$n.log" 2>&1; echo "$n rc=$?" >> "$OUT/STATUS_$PART"
}
prof() { # prof <outdir> <counters or --stats> -- program...
  local d=$1; shift
  rm -rf "$OUT/$d" "$OUT/$d.log"
  if [ "$1" = "--stats" ]; then shift; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$d" "$@" > "$OUT/$d.log" 2>&1; echo "$d rc=$?" >> "$OUT/STATUS_$PART"
  else local c=$1; shift; timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$d" "$@" > "$OUT/$d.log" 2>&1; echo "$d rc=$?" >> "$OUT/STATUS_$PART"; fi
}
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"
SQ2="SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
if [ "$PART" = "c" ]; then
  # 7. Hough batch: accumulator width and scans in flight, side by side on this box (builds the 16-bit variant here)
  bash tools/hough_ab.sh "$TAG"
elif [ "$PART" = "a" ]; then
  rm -f "$OUT/STATUS_$PART"
  # 1. the bench line exactly as the driver runs it (its own FETCH_SIZE / WRITE_SIZE / SQ child passes included)
  rm -f "$OUT/bench_line.json" "$OUT/bench_line.err"; rm -rf "$OUT/bench_pmc" gpurun_out/bench_pmc
  timeout -k 10 900 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"; echo "bench_line rc=$?" >> "$OUT/STATUS_$PART"
  cp -r gpurun_out/bench_pmc "$OUT/bench_pmc" 2>/dev/null || true
  echo "[profile] bench line done"
  # 2. kernel stats of the same command (no counters in this run).  The cards are made BEFORE the profiled run and
  # handed over by --cards: the program under `--` creates no child process (bench.py's card pool forks workers)
  rm -f "$OUT/cards.npy"
  python3 -c "import sys; sys.argv=['x']; import numpy as np, bench; c,_=bench.make_cards(64, 2); np.save('$OUT/cards.npy', c)" > "$OUT/cards.log" 2>&1; echo "cards rc=$?" >> "$OUT/STATUS_$PART"
  prof stats --stats -- python3 bench.py --no-cpu-baseline --no-pmc --no-e2e --cards "$OUT/cards.npy"
  rm -f "$OUT/cards.npy"
  echo "[profile] kernel stats done"
  # 3a. the scan-lane sweep kernel (one launch of 512 scans): timings, counters (tools/pmc_lanes.sh: six passes)
  step klanes timeout -k 10 300 python3 tools/klanes.py 512 512 3
  rm -rf "$OUT/pmc_lanes"; bash tools/pmc_lanes.sh "$OUT/pmc_lanes" all > "$OUT/pmc_lanes.log" 2>&1; echo "pmc_lanes rc=$?" >> "$OUT/STATUS_$PART"
  # 3b. the run-merging kernel (single scans and batches that do not fit the scan-lane scheme): as in round 3
  prof pmc_sq1 "$SQ1" -- python3 tools/kbatch.py 4 8
  prof pmc_sq2 "$SQ2" -- python3 tools/kbatch.py 4 8
  step kbatch timeout -k 10 300 python3 tools/kbatch.py 16 8
  step kbench timeout -k 10 300 python3 tools/kbench.py 20
  step kstamps timeout -k 10 300 python3 tools/kstamps.py
  echo "[profile] sweep kernels done"
  # 4. the batch warp on its own (sweep and warp never share the chip: the context is synchronised per call)
  step deskew timeout -k 10 300 python3 tools/bench_deskew.py 10
  prof deskew_stats --stats -- python3 tools/bench_deskew.py 10
  prof deskew_fetch FETCH_SIZE -- python3 tools/bench_deskew.py 4
  prof deskew_write WRITE_SIZE -- python3 tools/bench_deskew.py 4
  echo "[profile] batch warp done"
  step calls timeout -k 10 300 python3 tools/bench_calls.py
  step host timeout -k 10 300 python3 tools/bench_host.py 512 3
  step threads timeout -k 10 300 python3 tools/bench_threads.py
  step ksmall timeout -k 10 300 python3 tools/ksmall.py 4096 64
  step klanes600 timeout -k 10 300 python3 tools/klanes.py 128 64 2 7016 4960
  # where and when the workgroups of one scan-lane launch ran (a library of its own with -DSLANE_STAMP, built here)
  rm -f "$OUT/kstamps_lanes.log"
  bash tools/build_variant.sh stamp -DSLANE_STAMP > "$OUT/kstamps_lanes.build.log" 2>&1 && \
    OMR_AB_LIB=omr-img-corrector_amd/lib/variants/libomrdeskew_stamp.so timeout -k 10 300 python3 tools/ab_lib.py tools/kstamps_lanes.py > "$OUT/kstamps_lanes.log" 2>&1
  echo "kstamps_lanes rc=$?" >> "$OUT/STATUS_$PART"
  step deskew64 timeout -k 10 300 python3 tools/bench_deskew.py 6 64
  prof deskew64_stats --stats -- python3 tools/bench_deskew.py 6 64
else
  # 5. stage kernels, FFT, Hough: un-profiled numbers, kernel stats, FETCH_SIZE / WRITE_SIZE each alone
  step stages timeout -k 10 300 python3 tools/bench_stages.py 30
  prof stages_stats --stats -- python3 tools/bench_stages.py 10
  prof stages_fetch FETCH_SIZE -- python3 tools/bench_stages.py 5
  prof stages_write WRITE_SIZE -- python3 tools/bench_stages.py 5
  echo "[profile] stages done"
  step fft timeout -k 10 300 python3 tools/bench_fft.py 64 4
  step fft_a3 timeout -k 10 300 python3 tools/bench_fft.py 2 3 a3
  prof fft_a3_stats --stats -- python3 tools/bench_fft.py 2 2 a3
  prof fft_stats --stats -- python3 tools/bench_fft.py 32 3
  for w in c5 a4; do
    prof fft_fetch_$w FETCH_SIZE -- python3 tools/bench_fft.py 16 2 $w
    prof fft_write_$w WRITE_SIZE -- python3 tools/bench_fft.py 16 2 $w
    prof fft_sq_$w "$SQ1" -- python3 tools/bench_fft.py 16 2 $w
  done
  echo "[profile] fft done"
  step hough timeout -k 10 600 python3 tools/bench_hough.py 256 8 2
  step hough_single timeout -k 10 300 python3 tools/hough_run.py 1 1 5
  step hstamps_a4 timeout -k 10 300 python3 tools/hstamps.py
  prof hough_stats --stats -- python3 tools/hough_run.py 64 4 1
  prof hough_fetch FETCH_SIZE -- python3 tools/hough_run.py 64 4 1
  prof hough_write WRITE_SIZE -- python3 tools/hough_run.py 64 4 1
  prof hough_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/hough_run.py 64 4 1
  echo "[profile] hough done"
  step core_protocol timeout -k 10 600 python3 tools/core_protocol.py
  # 6. micro-benchmarks behind DESIGN.md's statements
  for t in lds_dma_window mov64_probe pf_probe lds_u16_probe; do
    rm -f "$OUT/$t.log"; hipcc -O2 --offload-arch=gfx950 -Wno-inline-asm -Wno-unused-value tools/$t.hip -o /tmp/$t > "$OUT/$t.build.log" 2>&1 && timeout -k 10 200 /tmp/$t > "$OUT/$t.log" 2>&1 || true
  done
fi
echo "[profile] part $PART done"
