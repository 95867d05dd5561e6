# SQ counters of the sweep kernel in batch mode (two passes of 8 counters); usage on the GPU box: bash tools/pmc_sweep.sh [outdir]
set -e
OUT=${1:-gpurun_out/pmc_sweep}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/p1" -- python3 tools/kbatch.py 4 8 > "$OUT/p1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d "$OUT/p2" -- python3 tools/kbatch.py 4 8 > "$OUT/p2.log" 2>&1
python3 tools/pmc_summary.py "$OUT/p1" runs_kernel
python3 tools/pmc_summary.py "$OUT/p2" runs_kernel
