#!/bin/bash
# Round profiling recipe (run on the GPU box through gpurun): un-profiled bench line, rocprofv3 kernel
# stats of the same command, HBM-side PMC passes (FETCH_SIZE and WRITE_SIZE in their own runs, as
# MI355X_MICROARCH.md prescribes), SQ counter passes of the sweep kernel.  Outputs under
# gpurun_out/<tag>/; tools/make_profiles.py turns them into profiles/<tag>_*.
#   usage: bash tools/profile_round.sh r01
set -eo pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
echo "[profile] bench line done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-cpu-baseline > "$OUT/stats.log" 2>&1
echo "[profile] kernel stats done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --no-cpu-baseline --steps 3 > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python bench.py --no-cpu-baseline --steps 3 > "$OUT/pmc_write.log" 2>&1
echo "[profile] HBM counters done"
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -- python tools/kbench.py 3 > "$OUT/pmc_sq1.log" 2>&1
timeout -k 10 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- python tools/kbench.py 3 > "$OUT/pmc_sq2.log" 2>&1
echo "[profile] SQ counters done"
timeout -k 10 300 python tools/kbench.py 10 > "$OUT/kbench.log" 2>&1
timeout -k 10 300 python tools/bench_stages.py > "$OUT/stages.log" 2>&1 || true
echo "[profile] all done"
