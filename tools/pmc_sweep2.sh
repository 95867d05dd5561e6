# instruction-cache / occupancy counters of the sweep kernel in batch mode; usage on the GPU box: bash tools/pmc_sweep2.sh [outdir]
set -e
OUT=${1:-gpurun_out/pmc_sweep2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d "$OUT/p1" -- python3 tools/kbatch.py 4 8 > "$OUT/p1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/p2" -- python3 tools/kbatch.py 4 8 > "$OUT/p2.log" 2>&1
python3 tools/pmc_summary.py "$OUT/p1" runs_kernel
python3 tools/pmc_summary.py "$OUT/p2" runs_kernel
